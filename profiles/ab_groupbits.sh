#!/bin/bash
# A/B of the merge's group size (2^4 / 2^5 / 2^6 voxels) on three shapes.  The builds: make kGroupBits in csrc/o3dr_device.h
# overridable (`#ifndef O3DR_GROUP_BITS ...`), compile with -DO3DR_GROUP_BITS=4 / 6 into lib/libo3dr_g4.so / _g6.so, copy the
# default build to _g5.so.  Round 4: all three within 1 % on all three shapes; the header was left as it is.
lib=online_3d_reconstruction_amd/lib
F="--steps 4 --warmup 1 --no-cpu-baseline --no-pcie-step --no-sor-leg"
for round in 1 2; do
 for v in g5 g4 g6; do
  cp $lib/libo3dr_$v.so $lib/libo3dr.so
  for shape in headline cfg3 cfg4; do
    case $shape in
      headline) A="" ;;
      cfg3) A="--rows 1080 --cols 1920 --voxel-size 0.02 --min-points 3 --frames 100" ;;
      cfg4) A="--rows 2160 --cols 4096 --jump-pixels 4 --frames 200" ;;
    esac
    timeout -k 10 240 python3 bench.py $F $A > gpurun_out/abg_${v}_${shape}_$round.json 2>/dev/null || { echo "$v $shape failed"; continue; }
    python3 - "$v" "$shape" "$round" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/abg_{sys.argv[1]}_{sys.argv[2]}_{sys.argv[3]}.json"))
k = d["kernel_ms_per_step"]
print(sys.argv[1], sys.argv[2], sys.argv[3], d["ms_per_step"], "groups", k["centroid_runs"], "segments", k["run_segments"], "scatter", k["radix_scatter"], flush=True)
PY
  done
 done
done
cp $lib/libo3dr_g5.so $lib/libo3dr.so
