#!/bin/bash
# A/B of bench flags on one library, interleaved: tools/abflags.sh lib_variant reps "flagsA" "flagsB" [common bench args]
v=$1; reps=$2; fa=$3; fb=$4; shift 4
for i in $(seq $reps); do
  for f in "$fa" "$fb"; do
    SPFA_LIB=$([ "$v" = main ] && echo cann-ops-adv-vllm_amd/lib || echo cann-ops-adv-vllm_amd/lib_var/$v)/libcust_opapi.so timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-dropin $f "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$f]', '%.4f' % d['roofline']['kernel_ms'])" || exit 1
  done
done
