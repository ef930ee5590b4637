#!/bin/bash
# tools/timeline_mid.sh [nq] — per-dispatch timeline of mid-size batches (33 ... 512 queries; rocprofv3 --kernel-trace)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
NQ=${1:-64}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format rocpd -d $ROOT/gpurun_out/tlm_db -- python3 $ROOT/tools/bench_search_sweep.py --nq $NQ --steps 3 --warmup 2 > $ROOT/gpurun_out/tlm.log 2>&1 || { tail -5 $ROOT/gpurun_out/tlm.log; exit 1; }
python3 $ROOT/tools/prof_timeline.py $(ls $ROOT/gpurun_out/tlm_db/*/*.db | head -1) 30 scan16_kernel | tail -31
rm -rf $ROOT/gpurun_out/tlm_db
