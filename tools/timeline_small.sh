#!/bin/bash
# tools/timeline_small.sh [nq] — per-dispatch timeline of small-batch searches (rocprofv3 --kernel-trace)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
NQ=${1:-1}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format rocpd -d $ROOT/gpurun_out/tls_db -- python3 $ROOT/tools/bench_search_sweep.py --nq $NQ --steps 4 --warmup 2 > $ROOT/gpurun_out/tls.log 2>&1 || { tail -5 $ROOT/gpurun_out/tls.log; exit 1; }
python3 $ROOT/tools/prof_timeline.py $(ls $ROOT/gpurun_out/tls_db/*/*.db | head -1) 27 stream16_kernel | tail -28
rm -rf $ROOT/gpurun_out/tls_db
