#!/bin/bash
# tools/ab_enc.sh <lib tags...> — encoder forward A/B on the GPU box (variant libraries under variants/), twice each
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
  echo "== default"; timeout -k 10 120 python3 tools/bench_encoder.py --steps 20 --warmup 3 2>/dev/null | cut -c1-110 || exit 1
  for L in "$@"; do
    echo "== lib $L"; OTTER_LIB=$PWD/variants/libotter_$L.so timeout -k 10 120 python3 tools/bench_encoder.py --steps 20 --warmup 3 2>/dev/null | cut -c1-110 || exit 1
  done
done
