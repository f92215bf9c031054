#!/bin/bash
# Runs GPU steps one after another under their own timeouts; stops at the first step that was KILLED or timed out
# (exit 124 / 137 / 143: never start another GPU step behind a hung one), carries on after an ordinary failure.
#   usage: tools/gpu_step.sh OUTDIR "name|timeout_s|command" ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/$1; shift
mkdir -p "$OUT"
for spec in "$@"; do
  name=${spec%%|*}; rest=${spec#*|}; tmo=${rest%%|*}; cmd=${rest#*|}
  echo "== $name (timeout ${tmo}s): $cmd"
  timeout -k 10 "$tmo" bash -c "$cmd" > "$OUT/$name.out" 2> "$OUT/$name.err"
  rc=$?
  echo "== $name rc=$rc"; tail -n 4 "$OUT/$name.out"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 143 ]; then echo "== $name was killed: stopping"; exit $rc; fi
done
exit 0
