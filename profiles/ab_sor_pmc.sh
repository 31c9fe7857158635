#!/bin/bash
# LDS / VALU counters of the outlier-removal kernels for library builds on ONE box: bash profiles/ab_sor_pmc.sh name1 name2 ...
# (files online_3d_reconstruction_amd/lib/libo3dr_<name>.so); dense 720p, SOR on, 10 frames, one step.
lib=online_3d_reconstruction_amd/lib
mkdir -p gpurun_out
for v in "$@"; do
  cp $lib/libo3dr_$v.so $lib/libo3dr.so
  rm -rf gpurun_out/sorpmc_$v
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS \
    --kernel-trace --output-format csv -d gpurun_out/sorpmc_$v -- python3 bench.py --sor --frames 10 --steps 1 --warmup 0 --no-cpu-baseline --no-pcie-step > /dev/null 2> gpurun_out/sorpmc_$v.err || { tail -5 gpurun_out/sorpmc_$v.err; exit 1; }
  echo "== $v"
  python3 profiles/sq_table.py gpurun_out/sorpmc_$v k_sor
done
