#!/bin/bash
# Round-end measurement refresh on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1200 -- 'bash tools/final_measure.sh'
# Writes everything under gpurun_out/final/; tools/rocpd_stats.py and tools/pmc_summary.py turn the
# outputs into the summaries committed under profiles/.
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT

timeout -k 10 300 python3 bench.py > $OUT/bench_f32.json 2> $OUT/bench_f32.err &&
timeout -k 10 200 python3 bench.py --dtype bf16 --no-cpu-baseline > $OUT/bench_bf16.json 2> $OUT/bench_bf16.err &&
timeout -k 10 200 python3 bench.py --tile --no-cpu-baseline > $OUT/bench_tile_f32.json 2> $OUT/bench_tile_f32.err &&
timeout -k 10 200 python3 bench.py --tile --dtype bf16 --no-cpu-baseline > $OUT/bench_tile_bf16.json 2> $OUT/bench_tile_bf16.err &&
echo "bench done" &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_f32 -o bench -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_f32_under_rocprof.json 2> $OUT/prof_f32.err &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_bf16 -o bench -- python3 bench.py --dtype bf16 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_bf16_under_rocprof.json 2> $OUT/prof_bf16.err &&
echo "trace done" &&
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  TAG=$(echo $C | cut -d' ' -f1)
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$TAG -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_$TAG.json 2> $OUT/pmc_$TAG.err || exit 1
  echo "pmc $TAG done"
done
