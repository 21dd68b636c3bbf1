#!/bin/bash
# Everything the round's profiles/ files come from, on ONE GPU box:  gpurun --timeout 3000 -- 'bash tools/final_profiles.sh r02'
# (PMC passes first, copied into profiles/ so that the bench lines that follow carry `traffic` for the library that ran.)
TAG=${1:-r04}
mkdir -p gpurun_out/final
O=gpurun_out/final
for D in "" "--dtype f32" "--dtype f16"; do
  bash tools/pmc_traffic.sh "$D" > $O/pmc_$(echo $D | tr -d ' -').txt 2>&1
done
cp gpurun_out/pmc_traffic_f32_split.json profiles/${TAG}_pmc_traffic_f32_split.json
cp gpurun_out/pmc_traffic.json profiles/${TAG}_pmc_traffic.json
cp gpurun_out/pmc_traffic_f16.json profiles/${TAG}_pmc_traffic_f16.json
cp profiles/${TAG}_pmc_traffic*.json $O/
timeout 600 python bench.py --layers > $O/${TAG}_bench.json 2> $O/${TAG}_layers.txt
timeout 600 python bench.py --dtype f32 --no-cpu-baseline --no-small-batch --layers > $O/${TAG}_bench_f32_operands.json 2> $O/${TAG}_layers_f32_operands.txt
timeout 600 python bench.py --heads sparse --no-cpu-baseline --no-f16-compare --no-small-batch > $O/${TAG}_bench_sparse_heads.json 2> /dev/null
timeout 600 python bench.py --heads allpass --no-cpu-baseline --no-f32-compare --no-f16-compare --no-small-batch > $O/${TAG}_bench_allpass_heads.json 2> /dev/null
timeout 600 python bench.py --batch 1 --latency-mode --layers --no-cpu-baseline --no-f32-compare --no-f16-compare --no-small-batch --no-extras > $O/${TAG}_bench_bs1_latency_mode.json 2> $O/${TAG}_layers_bs1_latency_mode.txt
timeout 600 python bench.py --dtype f16 --no-cpu-baseline --no-small-batch --layers > $O/${TAG}_bench_f16.json 2> $O/${TAG}_layers_f16.txt
timeout 600 python tools/latency_ab.py --bs 1 2 4 8 --ksplit 1 8 --iters 100 > $O/${TAG}_latency_ab.txt 2>&1
bash tools/rocprof_bench.sh $TAG > $O/rocprof.txt 2>&1
cp gpurun_out/prof_${TAG}_kernel_stats.csv $O/${TAG}_rocprofv3_kernel_stats.csv
timeout 600 python tools/split_error.py > $O/${TAG}_split_error.json 2> /dev/null
timeout 900 python tools/soak_in_flight.py f32_split 300 32 > $O/soak.txt 2>&1
bash tools/batch_sweep.sh 1 2 8 > $O/${TAG}_batch_sweep.txt 2>&1
OM_BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 5 --no-cpu-baseline --no-extras --no-f32-compare --no-f16-compare --no-small-batch > $O/force_dist.json 2> $O/force_dist.err; echo "force-dist rc $?" >> $O/soak.txt
cat $O/soak.txt | tail -3; cat $O/${TAG}_batch_sweep.txt; cut -c1-250 $O/${TAG}_bench.json
