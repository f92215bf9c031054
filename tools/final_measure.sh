#!/bin/bash
# Round-end measurement refresh on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1200 -- 'bash tools/final_measure.sh'
# Writes everything under gpurun_out/final/; tools/rocpd_stats.py and tools/pmc_summary.py turn the
# outputs into the summaries committed under profiles/ (tools/final_collect.sh).
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
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_tile_f32 -o bench -- python3 bench.py --tile --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/prof_tile_f32.err &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_tile_bf16 -o bench -- python3 bench.py --tile --dtype bf16 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/prof_tile_bf16.err &&
echo "trace done" || exit 1
# PMC: one counter set per pass (gfx950 slot limits; never together with the trace domains)
for W in "f32:" "bf16:--dtype bf16" "tile_f32:--tile" "tile_bf16:--tile --dtype bf16"; do
  TAGW=${W%%:*}; FL=${W#*:}
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_${TAGW}_$C -o pmc -- python3 bench.py $FL --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_${TAGW}_$C.err || exit 1
  done
  echo "pmc $TAGW done"
done
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_f32_SQ -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_f32_SQ.err || exit 1
echo "pmc SQ done"
