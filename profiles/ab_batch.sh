#!/bin/bash
# frames per launch group (O3DR_BATCH_FRAMES) against the step time: does a working set that fits the 256 MB Infinity Cache pay?
for b in 256 64 32 16 8 4; do
  O3DR_BATCH_FRAMES=$b timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-pcie-step --no-cpu-baseline --no-sor-leg > gpurun_out/abb_$b.json 2> gpurun_out/abb_$b.err || { tail -3 gpurun_out/abb_$b.err; exit 1; }
  python - "$b" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/abb_{sys.argv[1]}.json"))
print("batch", sys.argv[1], "ms/step", d["ms_per_step"], d["kernel_ms_per_step"], flush=True)
PY
done
