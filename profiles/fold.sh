#!/bin/bash
# Copies the summaries of `profiles/collect.sh <tag>` from gpurun_out/ into profiles/ (run in the build container).
set -e
tag=${1:-r02}
cp "$(ls -t gpurun_out/${tag}_stats/*/*kernel_stats.csv | head -1)" profiles/${tag}_kernel_stats_200frames.csv
cp gpurun_out/${tag}_bench_under_rocprof.json profiles/${tag}_bench_under_rocprof.json
cp gpurun_out/${tag}_bench.json profiles/${tag}_bench.json
python profiles/make_pmc_traffic.py gpurun_out/${tag}_fetch gpurun_out/${tag}_write profiles/${tag}_pmc_traffic.json > /dev/null
python profiles/sq_table.py gpurun_out/${tag}_sq > profiles/${tag}_sq_counters.txt
echo folded
