#!/bin/bash
# tools/ab.sh <lib1.so> <lib2.so> ... — bench each library variant (OTTER_LIB) with the same workload
for L in "$@"; do
  for W in 8 4; do
    if [ $W = 4 ]; then export OTTER_SCAN_W4=1; else unset OTTER_SCAN_W4; fi
    OTTER_LIB=$PWD/$L timeout -k 10 200 python -u bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('$L', 'W$W', round(j['value']), round(j['roofline']['kernel_ms'],3), round(j['roofline']['frac'],4), j['self_check'])"
  done
done
