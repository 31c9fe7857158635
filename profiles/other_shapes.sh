#!/bin/bash
# The other shapes quoted in profiles/README.md, through the same bench (gpurun --timeout 1100 -- 'bash profiles/other_shapes.sh').
set -o pipefail
B="python3 bench.py --no-pcie-step --no-sor-leg --no-reference-order"
run() { name=$1; shift; timeout -k 10 400 "$@" > gpurun_out/shape_$name.json 2> gpurun_out/shape_$name.err; python3 -c "
import json
d=json.load(open('gpurun_out/shape_$name.json'))
print('$name', d.get('metric'), d['value'], d['ms_per_step'], d.get('verified'))
"; }
run cfg4 $B --rows 1080 --cols 1920 --voxel-size 0.02 --min-points 3 --frames 100 --cpu-frames 20
run cfg5 $B --rows 2160 --cols 4096 --jump-pixels 4 --frames 100 --cpu-frames 20
run cfg1 $B --jump-pixels 15 --frames 50
run sor $B --sor --frames 50 --no-cpu-baseline
run sor_cfg1 $B --sor --jump-pixels 15 --frames 50 --no-cpu-baseline
run blur30 $B --blur-kernel 30 --frames 50
run host python3 bench.py --host-inputs --no-cpu-baseline --no-sor-leg
run pinned python3 bench.py --host-inputs --pinned --no-cpu-baseline --no-sor-leg
timeout -k 10 300 python3 profiles/real_frames_bench.py > gpurun_out/shape_real.txt 2>&1; tail -2 gpurun_out/shape_real.txt
