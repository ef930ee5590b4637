#!/bin/bash
# tools/pmc.sh <tag> <program args...> — rocprofv3 PMC passes (one run per pass) for the given python program.
# Run on the GPU box from the repo root: bash tools/pmc.sh r01_search bench.py --steps 3 --warmup 1 --no-cpu-baseline
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PASS_A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
PASS_B="FETCH_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU"
PASS_C="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
i=0
for P in "$PASS_A" "$PASS_B" "$PASS_C"; do
  i=$((i+1))
  timeout -s KILL 240 rocprofv3 --pmc $P --output-format csv -d $OUT/pass$i -- python3 $ROOT/"$@" > $OUT/pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/pass$i.log; exit 1; }
done
python3 $ROOT/tools/pmc_summary.py $OUT
