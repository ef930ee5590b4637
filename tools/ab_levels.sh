#!/bin/bash
# A/B of the sample-level settings of the bf16 scans in one GPU call (same box, same clocks).
cd ${GRAFT_REPO_ROOT:-.}
for cfg in "OTTER_SAMPLE_LEVELS=1" "OTTER_SAMPLE_LEVELS=4" "OTTER_SAMPLE_LEVELS=4 OTTER_SAMPLE_DIV=8" "OTTER_SAMPLE_LEVELS=5 OTTER_SAMPLE_DIV=5" "OTTER_SAMPLE_LEVELS=2"; do
  echo "== $cfg"
  env $cfg timeout -k 10 200 python -u tools/bench_search_sweep.py --nq 256,1024,2214,4096 2>/dev/null | grep "^{" | cut -c1-110 || exit 1
  env $cfg timeout -k 10 200 python -u tools/bench_search_sweep.py --nq 2214 --dtype bf16 2>/dev/null | grep "^{" | cut -c1-110 || exit 1
done
