#!/usr/bin/env python3
"""Folds a rocprofv3 --pmc SQ_* counter collection into per-kernel fractions of wave-cycles.
    python profiles/sq_table.py gpurun_out/<dir> [kernel substrings...]"""
import collections
import csv
import glob
import os
import sys


def main():
    d = sys.argv[1]
    want = sys.argv[2:]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    # (gpurun merges every call's files into gpurun_out/: only the newest collection counts)
    for path in sorted(glob.glob(f"{d}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)[-1:]:
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("o3dr::", "")
            if want and not any(w in name for w in want):
                continue
            acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
            acc[name]["_n_" + r["Counter_Name"]] += 1
    for name, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
        wc = c.get("SQ_WAVE_CYCLES", 0) or 1.0
        parts = [f"{k[3:]}={c[k] / wc:.3f}" for k in sorted(c) if k.startswith("SQ_") and k != "SQ_WAVE_CYCLES" and "INSTS" not in k]
        insts = [f"{k[3:]}={c[k]:.3g}" for k in sorted(c) if "INSTS" in k and not k.startswith("_n_")]
        print(f"{name}: wave_cycles={wc:.3g} " + " ".join(parts) + " | " + " ".join(insts))


if __name__ == "__main__":
    main()
