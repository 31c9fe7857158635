#!/bin/bash
# Collects the round's profiles on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1100 -- 'bash profiles/collect.sh r03'
# Outputs under gpurun_out/<tag>_*; profiles/fold.sh <tag> then copies the summaries into profiles/.
set -e -o pipefail
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
BENCH="bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pcie-step --no-sor-leg"
# 1. per-kernel time summary (+ the bench line of that profiled run)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_stats -- python3 $BENCH > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_stats.err
echo "stats done"
# 2. HBM traffic: two separate PMC passes, kernel-trace only
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/${tag}_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pcie-step --no-sor-leg > /dev/null 2> $out/${tag}_fetch.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/${tag}_write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pcie-step --no-sor-leg > /dev/null 2> $out/${tag}_write.err
echo "write done"
# 3. SQ view of the same command
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d $out/${tag}_sq -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-pcie-step --no-sor-leg > /dev/null 2> $out/${tag}_sq.err
echo "sq done"
# 3b. the PMC passes folded into profiles/<tag>_pmc_traffic.json HERE, stamped with the SHA-1 of the device sources, so that
#     the bench line of step 4 carries `roofline.traffic` (bench.py only reports a traffic file measured on its own sources)
python3 profiles/make_pmc_traffic.py $out/${tag}_fetch $out/${tag}_write profiles/${tag}_pmc_traffic.json > /dev/null
cp profiles/${tag}_pmc_traffic.json $out/${tag}_pmc_traffic.json
echo "traffic folded"
# 3c. the outlier-removal path (dense 720p, SOR on, 50 frames per step): per-kernel time summary
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_sor_stats -- python3 bench.py --sor --frames 50 --steps 3 --warmup 1 --no-cpu-baseline --no-pcie-step > $out/${tag}_sor_bench_under_rocprof.json 2> $out/${tag}_sor_stats.err
echo "sor stats done"
# 4. the un-profiled bench line (default command: verification, reference-order comparison, SOR leg, cpu_baseline)
python3 bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
echo "bench done"
find $out/${tag}_stats -name "*kernel_stats.csv" | head -1
