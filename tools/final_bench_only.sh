#!/bin/bash
# The bench legs of tools/final_measure.sh alone (after tools/final_collect.sh regenerated profiles/conv27_traffic*.json for the
# current kernel sources, so that the lines carry roofline.traffic):  gpurun --timeout 900 -- 'bash tools/final_bench_only.sh'
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=$GRAFT_REPO_ROOT/gpurun_out/final_bench
rm -rf "$OUT"; mkdir -p "$OUT"
run() {
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
run bench_tile_bf16_two_streams 200 python3 bench.py --tile --dtype bf16 --no-cpu-baseline --no-sweep --steps 5 --overlap-streams 2
python3 tools/pmc_stamp.py > "$OUT/src_sha.txt"
echo "bench done"
