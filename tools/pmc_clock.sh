#!/bin/bash
# tools/pmc_clock.sh — one PMC pass each over the headline search and the encoder; prints tools/pmc_derived.py's
# figures (effective clock, MFMA busy, share of the clock-limited matrix peak).  Run on the GPU box from the repo root.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_clock
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA GRBM_GUI_ACTIVE"
timeout -s KILL 240 rocprofv3 --pmc $P --output-format csv -d $OUT/search -- python3 $ROOT/tools/bench_search_sweep.py --nq 2214 --steps 40 --warmup 5 > $OUT/search.log 2>&1 || { tail -5 $OUT/search.log; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc $P --output-format csv -d $OUT/enc -- python3 $ROOT/tools/bench_encoder.py --steps 20 --warmup 3 > $OUT/enc.log 2>&1 || { tail -5 $OUT/enc.log; exit 1; }
echo "== search, 2214 queries x 1M x 768 (bench_search_sweep.py --nq 2214)"
python3 $ROOT/tools/pmc_derived.py $OUT/search --min-us 300
echo "== encoder, 128 blocks x 512 tokens (bench_encoder.py)"
python3 $ROOT/tools/pmc_derived.py $OUT/enc --min-us 80
rm -rf $OUT/search $OUT/enc
