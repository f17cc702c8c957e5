#!/bin/bash
# Build exploration variants of the library (one attention instantiation each) under
# cann-ops-adv-vllm_amd/lib_var/<name>/ ; usage: tools/build_variants.sh name "flags" [name "flags" ...]
set -e
cd "$(dirname "$0")/../cann-ops-adv-vllm_amd/csrc"
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  out=../lib_var/$name
  mkdir -p $out
  make -s -j6 OUT_DIR=$out EXTRA="-DSPFA_EXPLORE $flags" 2>&1 | grep -E "error" -A4 || true
  echo "built $name: $flags"
done
