#!/bin/bash
# Kernel-trace + stats profile of bench.py on the GPU box (run through gpurun from the repo root):
#   gpurun -- 'bash tools/rocprof_bench.sh r01'
# Writes gpurun_out/prof_<tag>/ and prints the kernel stats summary; copy the summary into profiles/.
set -e
TAG=${1:-r01}
EXTRA=${2:-}        # e.g. "--dtype f16"
R=$PWD
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o $TAG -- \
    python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-f32-compare --no-f16-compare --no-small-batch --in-flight 1 $EXTRA > $R/gpurun_out/prof_${TAG}_bench.json 2> $R/gpurun_out/prof_${TAG}.err
cd $R
STATS=$(find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1)
cp "$STATS" gpurun_out/prof_${TAG}_kernel_stats.csv
head -30 gpurun_out/prof_${TAG}_kernel_stats.csv
