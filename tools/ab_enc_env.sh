#!/bin/bash
# tools/ab_enc_env.sh "<VAR=val>" ... — per-kernel times (rocprofv3 kernel stats) of the encoder bench under environment settings
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for E in "$@"; do
  i=$((i+1))
  OUT=$ROOT/gpurun_out/abenv_$i; rm -rf $OUT; mkdir -p $OUT
  env $E timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/tools/bench_encoder.py --steps 5 --warmup 2 > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
  F=$(find $OUT -name '*kernel_stats.csv' | head -1)
  echo "== $E: $(grep -h blocks_per_s $OUT/run.log | cut -c1-60)"
  head -9 $F | tail -8 | awk -F, '{printf "   %-62s calls %s avg_us %.1f\n", substr($1,1,62), $2, $4/1000}'
done
