#!/bin/bash
# A/B of library builds on ONE box: bash profiles/ab_lib.sh name1 name2 ... (files online_3d_reconstruction_amd/lib/libo3dr_<name>.so,
# built beforehand with `make` + cp); each is copied over libo3dr.so and benched, the whole list twice (box noise shows).
lib=online_3d_reconstruction_amd/lib
mkdir -p gpurun_out
for round in 1 2; do
  for v in "$@"; do
    cp $lib/libo3dr_$v.so $lib/libo3dr.so
    timeout -k 10 240 python bench.py --steps 5 --warmup 2 --no-pcie-step ${AB_BENCH_FLAGS:---no-cpu-baseline} > gpurun_out/abl_${v}_$round.json 2> gpurun_out/abl_${v}_$round.err || { tail -5 gpurun_out/abl_${v}_$round.err; exit 1; }
    python - "$v" "$round" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/abl_{sys.argv[1]}_{sys.argv[2]}.json"))
print(sys.argv[1], sys.argv[2], d["ms_per_step"], "verified", d.get("verified"), d["kernel_ms_per_step"], flush=True)
PY
  done
done
