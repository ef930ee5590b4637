#!/bin/bash
# tools/ab_merge.sh — merge kernel variants (OTTER_MERGE_VARIANT) over the batch-size sweep
cd ${GRAFT_REPO_ROOT:-.}
for V in "" a b c; do
  echo "== OTTER_MERGE_VARIANT='$V'"
  OTTER_MERGE_VARIANT=$V timeout -k 10 200 python3 tools/bench_search_sweep.py --nq 1,32,64,256,2214 --steps 20 2>/dev/null | cut -c1-100 || exit 1
done
