#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc csv output: per kernel, the mean / min / max of every counter over its
dispatches (min and max separate launches of one kernel that do different amounts of work)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(root):
    acc = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                name = row.get("Kernel_Name", "?")
                short = name.replace("(anonymous namespace)::", "").replace("void ", "")
                short = short.split("(")[0][:90]
                acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
    lines = []
    for kern, ctrs in sorted(acc.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES", [0]))):
        n = max(len(v) for v in ctrs.values())
        lines.append("%s  (dispatches=%d)" % (kern, n))
        for c, v in sorted(ctrs.items()):
            lines.append("    %-28s mean %.6g  min %.6g  max %.6g" % (c, sum(v) / len(v), min(v), max(v)))
    text = "\n".join(lines)
    print(text)
    with open(os.path.join(root, "summary.txt"), "w") as f:
        f.write(text + "\n")


if __name__ == "__main__":
    main(sys.argv[1])
