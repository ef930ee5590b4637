#!/bin/bash
# tools/sweep_mid.sh [nq list] — scan/wall times of the mid-size batches on 1M x 768 (one line per nq)
NQ=${1:-16,32,33,48,64,65,128,256}
timeout -k 10 300 python tools/bench_search_sweep.py --nq $NQ > gpurun_out/sweep_mid.jsonl 2>&1
python3 - <<P
import json
for l in open("gpurun_out/sweep_mid.jsonl"):
    try: r=json.loads(l)
    except Exception: print(l.strip()[:200]); continue
    print(r["nq"], "scan_ms %.3f wall_ms %.3f fb %s" % (r["scan_ms"], r["wall_ms"], r.get("fallback_queries")))
P
