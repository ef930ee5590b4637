#!/bin/bash
# tools/ab_rank.sh — rank of the optimistic threshold among the sample's block maxima (OTTER_OPT_THR), alternating
cd ${GRAFT_REPO_ROOT:-.}
for rep in 1 2; do
  for R in 12 10 8 16; do
    OTTER_OPT_THR=$R timeout -k 10 200 python3 tools/bench_search_sweep.py --nq 256,2214 --steps 30 > gpurun_out/ab_rank.jsonl 2>/dev/null || exit 1
    python3 - <<P
import json
print("rank $R", " ".join("%d: %.3f ms (fallback %s)" % (r["nq"], r["wall_ms"], r.get("fallback_queries")) for r in (json.loads(l) for l in open("gpurun_out/ab_rank.jsonl") if l.startswith("{"))))
P
  done
done
