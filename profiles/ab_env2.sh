#!/bin/bash
# like ab_env.sh for combinations: bash profiles/ab_env2.sh "A=1 B=2" "A=0" ...
mkdir -p gpurun_out
i=0
for combo in "$@"; do
  i=$((i+1))
  env $combo timeout -k 10 240 python bench.py --steps 5 --warmup 2 --no-pcie-step > gpurun_out/ab2_$i.json 2> gpurun_out/ab2_$i.err || { tail -5 gpurun_out/ab2_$i.err; exit 1; }
  python - "$combo" "$i" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/ab2_{sys.argv[2]}.json"))
print(sys.argv[1], "|", d["ms_per_step"], "verified", d.get("verified"), d["kernel_ms_per_step"], flush=True)
PY
done
