#!/bin/bash
# tools/ab2.sh v1 v2 ... — bench scan-kernel tile variants (OTTER_SCAN_VARIANT letter, "d" = default)
for V in "$@"; do
  if [ $V = d ]; then unset OTTER_SCAN_VARIANT; else export OTTER_SCAN_VARIANT=$V; fi
  timeout -k 10 200 python -u bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('variant $V', round(j['value']), round(j['roofline']['kernel_ms'],3), round(j['roofline']['frac'],4), j['self_check'])"
done
