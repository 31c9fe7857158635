#!/bin/bash
# BASELINE.json configs[2], [3], [4] at their stated sizes on ONE GPU, verified against the oracle
# (gpurun --timeout 1200 -- 'bash profiles/r04_configs.sh [cfg2|cfg3|cfg4 ...]').
# One JSON line per config under gpurun_out/r04_config<i>_1gpu.json; a run that is killed at its limit stops the script.
set -o pipefail
out=gpurun_out
mkdir -p $out
( while sleep 45; do echo "[heartbeat] $(date +%T) $(free -g | awk '/Mem/{print "host used " $3 " GiB"}')"; done ) &
hb=$!
trap 'kill $hb 2>/dev/null' EXIT
B="python3 bench.py --gpus 1 --steps 3 --warmup 1 --no-pcie-step --no-sor-leg --no-reference-order --cpu-frames 100"
run() {
  name=$1; limit=$2; shift 2
  echo "== $name: $*"
  timeout -k 10 $limit "$@" > $out/r04_${name}_1gpu.json 2> $out/r04_${name}_1gpu.err
  rc=$?
  echo "== $name rc=$rc"
  tail -3 $out/r04_${name}_1gpu.err
  python3 - <<EOF
import json
try:
    d = json.load(open('$out/r04_${name}_1gpu.json'))
    print('$name', d['value'], 'frames/s', d['ms_per_step'], 'ms/step verified', d.get('verified'), d.get('points'))
except Exception as e:
    print('$name: no JSON line', e)
EOF
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed at its limit: stopping"; exit $rc; fi
}
want=${*:-cfg2 cfg3 cfg4}
for w in $want; do
  case $w in
    cfg2) run config2 1000 $B --total-frames 2000 ;;
    cfg3) run config3 700 $B --rows 1080 --cols 1920 --voxel-size 0.02 --min-points 3 --frames 200 ;;
    cfg4) run config4 1000 $B --rows 2160 --cols 4096 --jump-pixels 4 --total-frames 2000 ;;
  esac
done
