#!/bin/bash
# tools/ab_enc_prof.sh <kernel substring> <lib tags...> — per-kernel time (rocprofv3 kernel stats) of the encoder bench under the default
# library and variants/libotter_<tag>.so; prints the average duration of kernels whose name contains the substring.
K=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for L in default "$@"; do
  OUT=$ROOT/gpurun_out/abprof_$L; rm -rf $OUT; mkdir -p $OUT
  if [ "$L" != default ]; then export OTTER_LIB=$ROOT/variants/libotter_$L.so; fi
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/tools/bench_encoder.py --steps 5 --warmup 2 > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
  F=$(find $OUT -name '*kernel_stats.csv' | head -1)
  echo "== $L: $(grep -h blocks_per_s $OUT/run.log | cut -c1-60)"
  grep "$K" $F | awk -F, '{printf "   %s calls %s avg_us %.1f\n", substr($1,1,60), $2, $4/1000}'
done
