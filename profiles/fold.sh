#!/bin/bash
# Copies the summaries of `profiles/collect.sh <tag>` from gpurun_out/ into profiles/ (run in the build container).
set -e
tag=${1:-r03}
cp "$(ls -t gpurun_out/${tag}_stats/*/*kernel_stats.csv | head -1)" profiles/${tag}_kernel_stats_200frames.csv
cp gpurun_out/${tag}_bench_under_rocprof.json profiles/${tag}_bench_under_rocprof.json
cp gpurun_out/${tag}_bench.json profiles/${tag}_bench.json
cp gpurun_out/${tag}_pmc_traffic.json profiles/${tag}_pmc_traffic.json  # (folded on the box by collect.sh, before its bench run)
cp "$(ls -t gpurun_out/${tag}_sor_stats/*/*kernel_stats.csv | head -1)" profiles/${tag}_sor_kernel_stats_50frames.csv
cp gpurun_out/${tag}_sor_bench_under_rocprof.json profiles/${tag}_sor_bench_under_rocprof.json
python profiles/sq_table.py gpurun_out/${tag}_sq > profiles/${tag}_sq_counters.txt
echo folded
