#!/bin/bash
# tools/collect_r03.sh <tag> — the round's evidence on the GPU box, written under gpurun_out/<tag>_*:
# bench JSON; rocprofv3 kernel stats and PMC passes of `bench.py --no-extra --no-encode` (the d = 768 headline only,
# so scan16_kernel's average is one workload's); a per-dispatch timeline of one 2214-query search; encoder kernel
# stats and FETCH/WRITE passes.  Every GPU step under its own time limit; the script stops at the first failure.
set -u
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-encode --no-extra"
echo "[1] bench"; timeout -k 10 420 python3 $ROOT/bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || { tail -5 $OUT/${TAG}_bench.err; exit 1; }
echo "[2] kernel stats"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 $B > $OUT/${TAG}_stats.log 2>&1 || { tail -5 $OUT/${TAG}_stats.log; exit 1; }
cp $(ls $OUT/${TAG}_stats/*/*kernel_stats.csv | head -1) $OUT/${TAG}_bench_kernel_stats.csv
echo "[3] pmc"; bash $ROOT/tools/pmc.sh ${TAG}_bench bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-encode --no-extra > $OUT/${TAG}_bench_pmc.txt 2>&1 || { tail -5 $OUT/${TAG}_bench_pmc.txt; exit 1; }
python3 $ROOT/tools/traffic_from_pmc.py $OUT/pmc_${TAG}_bench $OUT/${TAG}_traffic_split.json > /dev/null || exit 1
echo "[4] timeline"; cd $ROOT && bash tools/timeline_2214.sh > $OUT/${TAG}_timeline_2214.txt 2>&1 || { tail -5 $OUT/${TAG}_timeline_2214.txt; exit 1; }
cd /tmp
echo "[5] encoder stats"; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_enc_stats -- python3 $ROOT/tools/bench_encoder.py --steps 10 --warmup 3 > $OUT/${TAG}_enc_stats.log 2>&1 || { tail -5 $OUT/${TAG}_enc_stats.log; exit 1; }
cp $(ls $OUT/${TAG}_enc_stats/*/*kernel_stats.csv | head -1) $OUT/${TAG}_encode_b128_kernel_stats.csv
echo "[6] encoder pmc"; bash $ROOT/tools/pmc.sh ${TAG}_enc tools/bench_encoder.py --steps 3 --warmup 1 > $OUT/${TAG}_encode_pmc.txt 2>&1 || { tail -5 $OUT/${TAG}_encode_pmc.txt; exit 1; }
python3 $ROOT/tools/encode_traffic_from_pmc.py $OUT/pmc_${TAG}_enc $OUT/${TAG}_encode_traffic.json 4 > /dev/null || exit 1
rm -rf $OUT/${TAG}_stats $OUT/${TAG}_enc_stats
echo done
