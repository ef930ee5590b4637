#!/bin/bash
# tools/stamp_variants.sh <lib tags...> — run the encoder bench under -DATTN_STAMP variant libraries, print the stamped cycle splits of one workgroup
cd ${GRAFT_REPO_ROOT:-.}
for L in "$@"; do
  echo "== $L"; OTTER_LIB=$PWD/variants/libotter_$L.so timeout -k 10 120 python3 tools/bench_encoder.py --steps 3 --warmup 1 2>&1 | grep "wg 100 item [23]" || exit 1
done
