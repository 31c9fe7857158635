#!/usr/bin/env python3
"""Turns two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; collected separately, kernel-trace only)
into profiles/<tag>_pmc_traffic.json: HBM bytes per launch for every kernel, with the gfx950
corrections of MI355X_MICROARCH.md "HBM": FETCH_SIZE reports half of the bytes of coalesced
streaming reads (verified here: k_voxel_keys reads exactly 16 B/point and FETCH_SIZE*1024 is 0.500
of that; k_radix_hist reads 4 B/record, ratio 0.500), WRITE_SIZE is exact for streaming stores
(k_reproject_emit: 16 B/point, ratio 1.00).  Counter unit is KiB.

    python profiles/make_pmc_traffic.py gpurun_out/r01_fetch gpurun_out/r01_write profiles/r01_pmc_traffic.json
"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load(d, counter):
    out = collections.defaultdict(list)
    # (gpurun merges every call's files into gpurun_out/: only the newest collection counts)
    for path in sorted(glob.glob(f"{d}/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)[-1:]:
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("o3dr::k_", "").split("<")[0]
                out[name].append(float(r["Counter_Value"]))
    return out


def main():
    fetch_dir, write_dir, out_path = sys.argv[1:4]
    dominant = sys.argv[4] if len(sys.argv) > 4 else "radix_scatter_lane"  # kernel symbol (without o3dr::k_)
    bench_class = sys.argv[5] if len(sys.argv) > 5 else "radix_scatter"   # bench.py's kernel class name
    f, w = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(f) | set(w)):
        n = max(len(f.get(k, [])), len(w.get(k, [])), 1)
        fetch = 2.0 * sum(f.get(k, [])) * 1024.0  # x2: gfx950 FETCH_SIZE counts 64 B per 128-B request
        write = sum(w.get(k, [])) * 1024.0
        kernels[k] = {"launches": n, "fetch_bytes_per_launch": round(fetch / n), "write_bytes_per_launch": round(write / n),
                      "hbm_bytes_per_launch": round((fetch + write) / n)}
    from bench import kernel_sources_sha1
    doc = {"kernel": bench_class, "kernel_symbol": "o3dr::k_" + dominant,
           "kernel_sources_sha1": kernel_sources_sha1(),  # bench.py only reports this file's traffic for the same device sources
           "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pcie-step --no-sor-leg",
           "hbm_bytes_per_launch": kernels.get(dominant, {}).get("hbm_bytes_per_launch"),
           "correction": "FETCH_SIZE x2 (gfx950), WRITE_SIZE x1, KiB -> bytes", "kernels": kernels}
    json.dump(doc, open(out_path, "w"), indent=1)
    print(json.dumps({k: v["hbm_bytes_per_launch"] for k, v in kernels.items()}))


if __name__ == "__main__":
    main()
