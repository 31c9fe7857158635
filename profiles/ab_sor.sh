#!/bin/bash
# A/B of library builds on ONE box for the outlier-removal path: bash profiles/ab_sor.sh name1 name2 ...
# (files online_3d_reconstruction_amd/lib/libo3dr_<name>.so, built beforehand); dense 720p, SOR on, 50 frames per step.
lib=online_3d_reconstruction_amd/lib
mkdir -p gpurun_out
for round in 1 2; do
  for v in "$@"; do
    cp $lib/libo3dr_$v.so $lib/libo3dr.so
    timeout -k 10 200 python bench.py --sor --frames 50 --steps 3 --warmup 1 --no-pcie-step --no-cpu-baseline > gpurun_out/abs_${v}_$round.json 2> gpurun_out/abs_${v}_$round.err || { tail -5 gpurun_out/abs_${v}_$round.err; exit 1; }
    python - "$v" "$round" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/abs_{sys.argv[1]}_{sys.argv[2]}.json"))
print(sys.argv[1], sys.argv[2], "frames/s", d["value"], "ms/step", d["ms_per_step"], flush=True)
PY
  done
done
