#!/bin/bash
# tools/prof_enc_ragged.sh — per-kernel times of the encoder on the ragged batch (lengths ~U[64, 512]) and the dense one
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for M in "--ragged" ""; do
  OUT=$ROOT/gpurun_out/encrag; rm -rf $OUT; mkdir -p $OUT
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/tools/bench_encoder.py $M --steps 5 --warmup 2 > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
  F=$(find $OUT -name '*kernel_stats.csv' | head -1)
  echo "== ${M:-dense}: $(grep -h blocks_per_s $OUT/run.log | cut -c1-70)"
  python3 - "$F" <<P
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:7]:
    print("   %-50s calls %4s avg %8.1f us" % (r["Name"].replace("(anonymous namespace)::","")[:50], r["Calls"], float(r["AverageNs"])/1e3))
P
done
rm -rf $ROOT/gpurun_out/encrag
