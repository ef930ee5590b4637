#!/bin/bash
# tools/collect_r02.sh <tag> — the round's evidence on the GPU box, written under gpurun_out/<tag>_*:
# bench JSON, rocprofv3 kernel stats of the same command, PMC passes (one run per pass), encoder kernel stats,
# batch-size sweeps.  Every GPU step under its own time limit; the script stops at the first failure.
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-encode"
echo "[1] bench"; timeout -k 10 420 python3 $ROOT/bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || { tail -5 $OUT/${TAG}_bench.err; exit 1; }
echo "[2] kernel stats"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 $B > $OUT/${TAG}_stats.log 2>&1 || { tail -5 $OUT/${TAG}_stats.log; exit 1; }
cp $(ls $OUT/${TAG}_stats/*/*kernel_stats.csv | head -1) $OUT/${TAG}_bench_kernel_stats.csv
echo "[3] pmc"; bash $ROOT/tools/pmc.sh ${TAG}_bench bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-encode > $OUT/${TAG}_bench_pmc.txt 2>&1 || { tail -5 $OUT/${TAG}_bench_pmc.txt; exit 1; }
cd /tmp
echo "[4] encoder stats"; timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_enc_stats -- python3 $ROOT/tools/bench_encoder.py --steps 10 --warmup 3 > $OUT/${TAG}_enc_stats.log 2>&1 || { tail -5 $OUT/${TAG}_enc_stats.log; exit 1; }
cp $(ls $OUT/${TAG}_enc_stats/*/*kernel_stats.csv | head -1) $OUT/${TAG}_encode_b128_kernel_stats.csv
echo "[5] sweeps"
timeout -k 10 200 python3 $ROOT/tools/bench_search_sweep.py --nq 1,2,4,8,16,17,24,32,33,64,128,256,512,1024,2214,4096 > $OUT/${TAG}_search_nq_sweep.jsonl 2>$OUT/${TAG}_sweep.err || { tail -5 $OUT/${TAG}_sweep.err; exit 1; }
timeout -k 10 200 python3 $ROOT/tools/bench_search_sweep.py --dtype bf16 --nq 1,8,16,17,32,64,256,2214 > $OUT/${TAG}_bf16_index_nq_sweep.jsonl 2>$OUT/${TAG}_sweep16.err || { tail -5 $OUT/${TAG}_sweep16.err; exit 1; }
timeout -k 10 120 $ROOT/variants/stream_probe > $OUT/${TAG}_stream_probe.txt 2>&1 || true
rm -rf $OUT/${TAG}_stats $OUT/${TAG}_enc_stats
echo done
