#!/bin/bash
# A/B timing of library variants on the bench workload, interleaved: tools/ab.sh "name1 name2" reps [bench args]
names=$1; reps=$2; shift 2
for i in $(seq $reps); do
  for n in $names; do
    SPFA_LIB=cann-ops-adv-vllm_amd/lib_var/$n/libcust_opapi.so timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-dropin "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', '%.4f' % d['roofline']['kernel_ms'])" || exit 1
  done
done
