#!/bin/bash
# tools/ab_env.sh "<VAR=val ...>" ... — same-box A/B of the 2214-query search under environment switches (each twice)
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
  for E in "$@"; do
    echo "== $E"; env $E timeout -k 10 150 python3 tools/bench_search_sweep.py --nq ${NQ:-2214} --steps 20 2>/dev/null | cut -c1-95 || exit 1
  done
done
