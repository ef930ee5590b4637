#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
timeout -k 10 200 python -u tools/bench_search_sweep.py --nq 96,256,1024,2214,4096 2>/dev/null | grep "^{" | cut -c1-100 || exit 1
timeout -k 10 200 python -u tools/bench_search_sweep.py --nq 1,64,256,2214 --dtype bf16 2>/dev/null | grep "^{" | cut -c1-100 || exit 1
