#!/bin/bash
# A/B of (library variant, bench flags) pairs, interleaved: tools/abmix.sh reps "variant|flags" "variant|flags" ... -- [common bench args]
reps=$1; shift
pairs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do pairs+=("$1"); shift; done
[ "$1" = "--" ] && shift
for i in $(seq $reps); do
  for pr in "${pairs[@]}"; do
    v=${pr%%|*}; f=${pr#*|}
    lib=$([ "$v" = main ] && echo cann-ops-adv-vllm_amd/lib || echo cann-ops-adv-vllm_amd/lib_var/$v)/libcust_opapi.so
    SPFA_LIB=$lib timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-dropin $f "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$pr]', '%.4f' % d['roofline']['kernel_ms'])" || exit 1
  done
done
