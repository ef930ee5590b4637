#!/bin/bash
# tools/ab_lib.sh <variant tag> [nq list] — same-box A/B of the search sweep: default library vs variants/libotter_<tag>.so, twice each
cd ${GRAFT_REPO_ROOT:-.}
TAG=$1; NQ=${2:-2214}
for rep in 1 2; do
  echo "== default"; timeout -k 10 150 python3 tools/bench_search_sweep.py --nq $NQ --steps 20 2>/dev/null | cut -c1-90 || exit 1
  echo "== $TAG"; OTTER_LIB=$PWD/variants/libotter_$TAG.so timeout -k 10 150 python3 tools/bench_search_sweep.py --nq $NQ --steps 20 2>/dev/null | cut -c1-90 || exit 1
done
