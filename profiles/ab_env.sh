#!/bin/bash
# A/B of one environment switch on the GPU box: bash profiles/ab_env.sh VAR value [value ...] [-- extra bench flags]
# Runs the default bench (verification on, no PCIe step) once per value and prints ms/step + the per-kernel split.
var=$1; shift
vals=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do vals+=("$1"); shift; done
[ "$1" == "--" ] && shift
mkdir -p gpurun_out
for m in "${vals[@]}"; do
  env "$var=$m" timeout -k 10 240 python bench.py --steps 5 --warmup 2 --no-pcie-step "$@" > gpurun_out/ab_${var}_$m.json 2> gpurun_out/ab_${var}_$m.err || { tail -5 gpurun_out/ab_${var}_$m.err; exit 1; }
  python - "$var" "$m" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/ab_{sys.argv[1]}_{sys.argv[2]}.json"))
print(sys.argv[1], sys.argv[2], d["ms_per_step"], "verified", d.get("verified"), d["kernel_ms_per_step"], flush=True)
PY
done
