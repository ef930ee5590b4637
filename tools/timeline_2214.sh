#!/bin/bash
# tools/timeline_2214.sh <tag> — per-dispatch timeline of the 2214-query search (rocprofv3 --kernel-trace, rocpd database)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-tl}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format rocpd -d $ROOT/gpurun_out/${TAG}_db -- python3 $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-encode --no-extra > $ROOT/gpurun_out/${TAG}.log 2>&1 || { tail -5 $ROOT/gpurun_out/${TAG}.log; exit 1; }
DB=$(ls $ROOT/gpurun_out/${TAG}_db/*/*.db | head -1)
python3 $ROOT/tools/prof_timeline.py $DB 45 scan16_kernel > $ROOT/gpurun_out/${TAG}_timeline.txt
rm -rf $ROOT/gpurun_out/${TAG}_db
tail -47 $ROOT/gpurun_out/${TAG}_timeline.txt
