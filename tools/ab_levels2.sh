#!/bin/bash
# tools/ab_levels2.sh — sample-level plan of the coarse scan (tuning switches), 2214 and 4096 queries
cd ${GRAFT_REPO_ROOT:-.}
run() { echo "== $*"; env "$@" timeout -k 10 120 python3 tools/bench_search_sweep.py --nq 2214,4096 --steps 15 2>/dev/null | cut -c1-88 || exit 1; }
run OTTER_X=0
run OTTER_SAMPLE_LEVELS=4
run OTTER_SAMPLE_LEVELS=4 OTTER_SAMPLE_DIV=3
run OTTER_SAMPLE_LEVELS=3 OTTER_SAMPLE_DIV=6
run OTTER_SAMPLE_LEVELS=3 OTTER_SAMPLE_DIV=3
run OTTER_SAMPLE_LEVELS=2 OTTER_SAMPLE_DIV=8
run OTTER_SAMPLE_LEVELS=5 OTTER_SAMPLE_DIV=4
run OTTER_X=0
