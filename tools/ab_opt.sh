#!/bin/bash
# tools/ab_opt.sh — optimistic-threshold A/B on the GPU box: bench sweep at several batch sizes, sample levels (OTTER_OPT_THR=0) vs the optimistic start
cd ${GRAFT_REPO_ROOT:-.}
for NQ in 512 1024 2214 4096; do
  for SET in "OTTER_OPT_THR=0" "OTTER_OPT_THR=12"; do
    echo "== nq $NQ $SET"
    env $SET timeout -k 10 150 python3 tools/bench_search_sweep.py --nq $NQ --steps 20 2>/dev/null | cut -c1-100 || exit 1
  done
done
