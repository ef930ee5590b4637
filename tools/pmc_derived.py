#!/usr/bin/env python3
"""Derived figures from ONE rocprofv3 --pmc pass (counter_collection.csv) that holds
GRBM_GUI_ACTIVE, SQ_BUSY_CYCLES, SQ_WAVE_CYCLES, SQ_VALU_MFMA_BUSY_CYCLES, SQ_INSTS_MFMA, SQ_WAIT_ANY,
SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY -- per kernel, over its dispatches of at least --min-us:

  clock      effective shader clock = GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration (MI355X_MICROARCH.md, DVFS
             give-back: reads high on dispatches shorter than ~0.3 ms)
  mfma_busy  SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8): share of SIMD cycles with the matrix
             pipe busy (the counter is in cycles: 16 per v_mfma_f32_16x16x32_bf16, 32 per 32x32x16)
  flops      16,384 flop per 16x16x32 bf16 MFMA x SQ_INSTS_MFMA / duration (--flop-per-mfma for other shapes)
  peak@clock 1024 SIMDs x 1024 flop/cycle x clock: what the matrix pipes could do at the clock the chip held
  wave cycles split: parked (s_waitcnt / barrier), issue-stalled, issuing

    python3 tools/pmc_derived.py <dir with *counter_collection.csv> [--kernel substr] [--min-us 300]
"""
import argparse
import csv
import glob
import os
from collections import defaultdict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("root")
    ap.add_argument("--kernel", default="")
    ap.add_argument("--min-us", type=float, default=300.0)
    ap.add_argument("--flop-per-mfma", type=float, default=16384.0)
    a = ap.parse_args()
    disp = defaultdict(dict)
    for path in glob.glob(os.path.join(a.root, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                key = (path, row["Dispatch_Id"])
                d = disp[key]
                d["name"] = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
                d[row["Counter_Name"]] = d.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                if "Start_Timestamp" in row and row["Start_Timestamp"]:
                    d["dur_us"] = (float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) / 1e3
    per = defaultdict(list)
    for d in disp.values():
        if a.kernel in d["name"] and d.get("dur_us", 0.0) >= a.min_us and d.get("GRBM_GUI_ACTIVE", 0) > 0:
            per[d["name"]].append(d)
    for name, ds in sorted(per.items(), key=lambda kv: -sum(x["dur_us"] for x in kv[1])):
        def tot(c):
            return sum(x.get(c, 0.0) for x in ds)
        dur = tot("dur_us") * 1e-6
        cyc = tot("GRBM_GUI_ACTIVE") / 8.0
        clock = cyc / dur
        busy = tot("SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * cyc) if cyc else 0.0
        flops = tot("SQ_INSTS_MFMA") * a.flop_per_mfma / dur
        peak = 1024.0 * 1024.0 * clock
        wc = tot("SQ_WAVE_CYCLES")
        print("%s: %d dispatches >= %.0f us, %.3f ms each" % (name, len(ds), a.min_us, dur * 1e3 / len(ds)))
        print("    clock %.2f GHz | mfma_busy %.1f %% | %.0f TFLOP/s issued = %.1f %% of the %.0f TFLOP/s the pipes do at that clock"
              " (%.1f %% of the nominal 2500)" % (clock / 1e9, 100 * busy, flops / 1e12, 100 * flops / peak if peak else 0,
                                                   peak / 1e12, 100 * flops / 2.5e15))
        if wc:
            print("    wave cycles: parked %.1f %% | issue-stalled %.1f %% | issuing %.1f %%"
                  % (100 * tot("SQ_WAIT_ANY") / wc, 100 * tot("SQ_WAIT_INST_ANY") / wc, 100 * tot("SQ_ACTIVE_INST_ANY") / wc))


if __name__ == "__main__":
    main()
