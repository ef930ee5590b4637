#!/bin/bash
# tools/ab_small.sh — stream-kernel A/B on the GPU box: stage shapes (OTTER_STREAM_VARIANT) and variant libraries
cd ${GRAFT_REPO_ROOT:-.}
timeout -k 10 60 variants/stream_probe | grep -E "nt_8192|ring3" || exit 1
for V in "" a b; do
  echo "== OTTER_STREAM_VARIANT='$V'"
  OTTER_STREAM_VARIANT=$V timeout -k 10 120 python3 tools/bench_search_sweep.py --nq 1,8 --steps 30 2>/dev/null | cut -c1-120 || exit 1
done
for L in "$@"; do
  for V in "" a; do
  echo "== lib $L OTTER_STREAM_VARIANT='$V'"
  OTTER_STREAM_VARIANT=$V OTTER_LIB=$PWD/variants/libotter_$L.so timeout -k 10 120 python3 tools/bench_search_sweep.py --nq 1,8 --steps 30 2>/dev/null | cut -c1-120 || exit 1
  done
done
