#!/bin/bash
# tools/prof_enc.sh <tag> — rocprofv3 kernel stats of one encoder bench run (B = 128, L = 512, bf16); run on the GPU box.
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/tools/bench_encoder.py --steps 5 --warmup 2 > $OUT/run.log 2>&1 || { tail -5 $OUT/run.log; exit 1; }
F=$(find $OUT -name '*kernel_stats.csv' | head -1)
cp $F $ROOT/gpurun_out/${TAG}_encode_b128_kernel_stats.csv
python3 - "$F" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]:
    print("%-70s calls %5s avg_us %9.1f pct %5s" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
P
