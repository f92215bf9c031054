#!/bin/bash
# Turns gpurun_out/final/ (tools/final_measure.sh) into the summaries committed under profiles/.
# Usage: bash tools/final_collect.sh r03        (run in the build container, from the repo root, on the measured commit)
set -e
R=${1:-r02}
F=gpurun_out/final
COMMIT=$(git rev-parse --short HEAD)
SHA_BOX=$(cat $F/src_sha.txt)
SHA_HERE=$(python3 tools/pmc_stamp.py)
if [ "$SHA_BOX" != "$SHA_HERE" ]; then echo "kernel sources changed since the measurement ($SHA_BOX vs $SHA_HERE): re-measure"; exit 1; fi
for n in f32 bf16 f16 tile_f32 tile_bf16 tile_f16 sweep_bf16_b16 sweep_bf16_b16_noshare; do cp $F/bench_$n.json profiles/${R}_bench_$n.json; done
{
  for W in "f32:(fp32, configs[1])" "bf16:--dtype bf16 (configs[1] shape)" "tile_f32:--tile (one test_brn tile per step, fp32)" "tile_bf16:--tile --dtype bf16"; do
    TAGW=${W%%:*}; D=${W#*:}
    echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py ${D} --steps 3 --warmup 1 --no-cpu-baseline --no-sweep   @ $COMMIT"
    python tools/rocpd_stats.py $(find $F/prof_$TAGW -name "*.db" | head -1)
    echo
  done
} > profiles/${R}_bench_kernel_stats.txt
{
  echo "## fp32, configs[1]   @ $COMMIT"
  python tools/pmc_summary.py $F/pmc_f32_FETCH_SIZE/pmc_counter_collection.csv $F/pmc_f32_WRITE_SIZE/pmc_counter_collection.csv $F/pmc_f32_SQ/pmc_counter_collection.csv profiles/conv27_traffic.json "conv3d_mfma<2," $COMMIT
  echo; echo "## bf16, configs[1] shape"
  python tools/pmc_summary.py $F/pmc_bf16_FETCH_SIZE/pmc_counter_collection.csv $F/pmc_bf16_WRITE_SIZE/pmc_counter_collection.csv $F/pmc_bf16_SQ/pmc_counter_collection.csv profiles/conv27_traffic_bf16.json conv27_ $COMMIT
  echo; echo "## fp32, test_brn tile"
  python tools/pmc_summary.py $F/pmc_tile_f32_FETCH_SIZE/pmc_counter_collection.csv $F/pmc_tile_f32_WRITE_SIZE/pmc_counter_collection.csv $F/pmc_tile_f32_SQ/pmc_counter_collection.csv profiles/conv27_traffic_tile.json "conv3d_mfma<2," $COMMIT
  echo; echo "## bf16, test_brn tile"
  python tools/pmc_summary.py $F/pmc_tile_bf16_FETCH_SIZE/pmc_counter_collection.csv $F/pmc_tile_bf16_WRITE_SIZE/pmc_counter_collection.csv $F/pmc_tile_bf16_SQ/pmc_counter_collection.csv profiles/conv27_traffic_bf16_tile.json conv27_ $COMMIT
} > profiles/${R}_pmc_summary.txt
{
  echo "# per-kernel HBM table, bench.py --tile --dtype bf16 (one test_brn tile per step), kernels >= 0.5 % of the step's kernel time   @ $COMMIT"
  echo "# durations: rocprofv3 --kernel-trace; bytes: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (FETCH_SIZE x 2, gfx950); peak 8 TB/s"
  python tools/hbm_table.py $(find $F/prof_tile_bf16 -name "*.db" | head -1) profiles/conv27_traffic_bf16_tile.json
  echo
  echo "# the same for bench.py --dtype bf16 (configs[1] shape)"
  python tools/hbm_table.py $(find $F/prof_bf16 -name "*.db" | head -1) profiles/conv27_traffic_bf16.json
} > profiles/${R}_kernel_hbm_table.txt
grep -E "HBM traffic per launch|mfma_busy" profiles/${R}_pmc_summary.txt
