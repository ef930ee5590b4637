#!/bin/bash
# tools/pmc_fetch.sh <tag> <program args...> — one rocprofv3 PMC pass (FETCH_SIZE only) of a python program.
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pass1 -- python3 $ROOT/"$@" > $OUT/pass1.log 2>&1 || { echo "pass failed"; tail -5 $OUT/pass1.log; exit 1; }
python3 $ROOT/tools/pmc_summary.py $OUT | grep -A2 "scan16\|scan_kernel" | head -12
