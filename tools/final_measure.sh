#!/bin/bash
# Round-end measurement refresh on the GPU box (run through gpurun from the repo root):
#   gpurun --timeout 1200 -- 'bash tools/final_measure.sh'
# Writes everything under gpurun_out/final/; tools/final_collect.sh turns it into the summaries committed under profiles/.
# One step per line; a step that is killed / times out ends the script (no GPU step is started behind a hung one).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/final
rm -rf "$OUT"; mkdir -p "$OUT"
run() {   # name timeout command...
  local name=$1 tmo=$2; shift 2
  timeout -k 10 "$tmo" "$@" > "$OUT/$name.json" 2> "$OUT/$name.err"
  local rc=$?
  echo "== $name rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 143 ]; then echo "killed: stopping"; exit $rc; fi
}
run bench_f32 400 python3 bench.py
run bench_bf16 200 python3 bench.py --dtype bf16 --no-cpu-baseline --no-sweep
run bench_f16 200 python3 bench.py --dtype f16 --no-cpu-baseline --no-sweep
run bench_tile_f32 200 python3 bench.py --tile --no-cpu-baseline --no-sweep --steps 5
run bench_tile_bf16 200 python3 bench.py --tile --dtype bf16 --no-cpu-baseline --no-sweep --steps 5
run bench_tile_f16 200 python3 bench.py --tile --dtype f16 --no-cpu-baseline --no-sweep --steps 5
run bench_sweep_bf16_b16 300 python3 bench.py --sweep --sweep-hnm 2 --sweep-wnm 16 --sweep-batch-tiles 16 --steps 2 --warmup 1 --no-cpu-baseline
run bench_sweep_bf16_b16_noshare 300 python3 bench.py --sweep --sweep-hnm 2 --sweep-wnm 16 --sweep-batch-tiles 16 --sweep-share-halo 0 --steps 2 --warmup 1 --no-cpu-baseline
echo "bench done"
# kernel traces (the program itself after `--`: no env / shell hop under the profiler)
for W in "f32:" "bf16:--dtype bf16" "tile_f32:--tile" "tile_bf16:--tile --dtype bf16"; do
  TAGW=${W%%:*}; FL=${W#*:}
  run trace_$TAGW 300 rocprofv3 --kernel-trace --stats -d "$OUT/prof_$TAGW" -o bench -- python3 bench.py $FL --steps 3 --warmup 1 --no-cpu-baseline --no-sweep
done
echo "trace done"
# PMC: one counter set per pass (gfx950 slot limits; never together with the trace domains)
for W in "f32:" "bf16:--dtype bf16" "tile_f32:--tile" "tile_bf16:--tile --dtype bf16"; do
  TAGW=${W%%:*}; FL=${W#*:}
  run pmc_${TAGW}_FETCH 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_${TAGW}_FETCH_SIZE" -o pmc -- python3 bench.py $FL --steps 1 --warmup 1 --no-cpu-baseline --no-sweep
  run pmc_${TAGW}_WRITE 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_${TAGW}_WRITE_SIZE" -o pmc -- python3 bench.py $FL --steps 1 --warmup 1 --no-cpu-baseline --no-sweep
  run pmc_${TAGW}_SQ 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_${TAGW}_SQ" -o pmc -- python3 bench.py $FL --steps 1 --warmup 1 --no-cpu-baseline --no-sweep
  echo "pmc $TAGW done"
done
python3 tools/pmc_stamp.py > "$OUT/src_sha.txt"
echo "all done"
