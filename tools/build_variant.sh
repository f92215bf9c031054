#!/bin/bash
# Builds libteramind_hip.so of a git ref (default: the working tree) into ab/<name>.so for same-box A/B timing:
#   tools/build_variant.sh A HEAD          # the committed kernels
#   tools/build_variant.sh B               # the working tree
#   gpurun -- 'for r in 1 2; do for v in A B; do TM_LIB_PATH=ab/$v.so python bench.py ...; done; done'
# (timings taken on different gpurun boxes differ by several percent: never compare across calls)
set -e
NAME=$1; REF=$2; EXTRA=$3
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$ROOT/ab"
if [ -z "$REF" ]; then
  make -C "$ROOT/tera-mind_amd/csrc" -j6 > /dev/null
  cp "$ROOT/tera-mind_amd/csrc/libteramind_hip.so" "$ROOT/ab/$NAME.so"
else
  T=$(mktemp -d)
  (cd "$ROOT" && git archive "$REF" tera-mind_amd/csrc include | tar -x -C "$T")
  make -C "$T/tera-mind_amd/csrc" -j6 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -ffp-contract=fast $EXTRA" > /dev/null
  cp "$T/tera-mind_amd/csrc/libteramind_hip.so" "$ROOT/ab/$NAME.so"
  rm -rf "$T"
fi
ls -la "$ROOT/ab/$NAME.so"
