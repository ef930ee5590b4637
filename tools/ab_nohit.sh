#!/bin/bash
# tools/ab_nohit.sh — duration of the one-scan dispatch with its real thresholds and with an unreachable one (SCAN_DIAG build,
# OTTER_DIAG_NOHIT=1: no row passes, results are garbage): what the appends of a scan cost
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
export OTTER_LIB=$ROOT/variants/libotter_diag.so
for v in 0 1; do
  if [ $v == 1 ]; then export OTTER_DIAG_NOHIT=1; fi
  rm -rf $ROOT/gpurun_out/nh_db
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format rocpd -d $ROOT/gpurun_out/nh_db -- python3 $ROOT/tools/bench_search_sweep.py --nq 2214 --steps 3 --warmup 2 > $ROOT/gpurun_out/nh.log 2>&1 || { tail -5 $ROOT/gpurun_out/nh.log; exit 1; }
  echo "== NOHIT=$v"; python3 $ROOT/tools/prof_timeline.py $(ls $ROOT/gpurun_out/nh_db/*/*.db | head -1) 60 | grep "scan16_kernel<0, 12" | tail -4
done
rm -rf $ROOT/gpurun_out/nh_db
