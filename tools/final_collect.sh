#!/bin/bash
# Turns gpurun_out/final/ (tools/final_measure.sh) into the summaries committed under profiles/.  Usage: bash tools/final_collect.sh <version tag, e.g. v5>
set -e
V=${1:-v5}
F=gpurun_out/final
for n in f32 bf16 tile_f32 tile_bf16; do cp $F/bench_$n.json profiles/r01_bench_$n.json; done
cp $F/bench_f32_under_rocprof.json profiles/r01_bench_f32_under_rocprof.json
cp $F/bench_bf16_under_rocprof.json profiles/r01_bench_bf16_under_rocprof.json
{
  echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline  (fp32, configs[1])"
  python tools/rocpd_stats.py $F/prof_f32/bench_results.db
  echo; echo "# same command with --dtype bf16"
  python tools/rocpd_stats.py $F/prof_bf16/bench_results.db
  echo; echo "# python3 bench.py --tile --steps 2 --warmup 1 (one test_brn tile per step, fp32)"
  python tools/rocpd_stats.py $F/prof_tile_f32/bench_results.db
  echo; echo "# python3 bench.py --tile --dtype bf16 --steps 3 --warmup 1"
  python tools/rocpd_stats.py $F/prof_tile_bf16/bench_results.db
} > profiles/r01_bench_kernel_stats_$V.txt
{
  echo "## fp32, configs[1]"
  python tools/pmc_summary.py $F/pmc_f32_FETCH_SIZE/pmc_counter_collection.csv $F/pmc_f32_WRITE_SIZE/pmc_counter_collection.csv $F/pmc_f32_SQ/pmc_counter_collection.csv profiles/conv27_traffic.json "conv3d_mfma<2,"
  echo; echo "## bf16, configs[1] shape"
  python tools/pmc_summary.py $F/pmc_bf16_FETCH_SIZE/pmc_counter_collection.csv $F/pmc_bf16_WRITE_SIZE/pmc_counter_collection.csv - profiles/conv27_traffic_bf16.json conv27_bf16
  echo; echo "## fp32, test_brn tile"
  python tools/pmc_summary.py $F/pmc_tile_f32_FETCH_SIZE/pmc_counter_collection.csv $F/pmc_tile_f32_WRITE_SIZE/pmc_counter_collection.csv - profiles/conv27_traffic_tile.json "conv3d_mfma<2,"
  echo; echo "## bf16, test_brn tile"
  python tools/pmc_summary.py $F/pmc_tile_bf16_FETCH_SIZE/pmc_counter_collection.csv $F/pmc_tile_bf16_WRITE_SIZE/pmc_counter_collection.csv - profiles/conv27_traffic_bf16_tile.json conv27_bf16
} > profiles/r01_pmc_summary_$V.txt
grep "HBM traffic per launch" profiles/r01_pmc_summary_$V.txt
