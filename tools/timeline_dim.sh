#!/bin/bash
# tools/timeline_dim.sh <dim> [nq] — per-dispatch timeline of one large-batch search at another dimension
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
DIM=${1:-2304}; NQ=${2:-2214}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format rocpd -d $ROOT/gpurun_out/tld_db -- python3 $ROOT/tools/bench_search_sweep.py --dim $DIM --nq $NQ --steps 3 --warmup 2 > $ROOT/gpurun_out/tld.log 2>&1 || { tail -5 $ROOT/gpurun_out/tld.log; exit 1; }
grep "^{" $ROOT/gpurun_out/tld.log | cut -c1-160
python3 $ROOT/tools/prof_timeline.py $(ls $ROOT/gpurun_out/tld_db/*/*.db | head -1) 24 scan16_kernel | tail -25
rm -rf $ROOT/gpurun_out/tld_db
