#!/bin/bash
# tools/ab_grow.sh — the optimistic start's larger sample with its Poisson rank (default) against the one-round sample and
# rank 12 (OTTER_OPT_GROW=0), alternating on one box
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
  for V in 0 1; do
    OTTER_OPT_GROW=$V timeout -k 10 250 python3 tools/bench_search_sweep.py --nq ${1:-128,256,512,1024,2214,4096} --steps 30 > gpurun_out/ab_grow.jsonl 2>/dev/null || exit 1
    python3 - <<P
import json
print("grow=$V", " ".join("%d: %.3f (fb %s)" % (r["nq"], r["wall_ms"], r.get("fallback_queries")) for r in (json.loads(l) for l in open("gpurun_out/ab_grow.jsonl") if l.startswith("{"))))
P
  done
done
