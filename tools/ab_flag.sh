#!/bin/bash
# A/B of one bench flag on one library variant, interleaved: tools/ab_flag.sh variant "flag" reps [bench args]
v=$1; flag=$2; reps=$3; shift 3
for i in $(seq $reps); do
  for f in "" "$flag"; do
    SPFA_LIB=cann-ops-adv-vllm_amd/lib_var/$v/libcust_opapi.so timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-dropin $f "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$f]', '%.4f' % d['roofline']['kernel_ms'])" || exit 1
  done
done
