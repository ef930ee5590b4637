#!/bin/bash
# tools/ab_enc_lib.sh <variant tag> — same-box A/B of the encoder bench (dense and ragged): default library vs variants/libotter_<tag>.so, twice each
cd ${GRAFT_REPO_ROOT:-.}
TAG=$1
for rep in 1 2; do
  for M in "" "--ragged"; do
    echo "== default ${M:-dense}: $(timeout -k 10 150 python3 tools/bench_encoder.py $M --steps 8 --warmup 3 2>/dev/null | cut -c1-72)" || exit 1
    echo "== $TAG ${M:-dense}: $(OTTER_LIB=$PWD/variants/libotter_$TAG.so timeout -k 10 150 python3 tools/bench_encoder.py $M --steps 8 --warmup 3 2>/dev/null | cut -c1-72)" || exit 1
  done
done
