#!/bin/bash
# A/B of one environment setting on the shipped library, interleaved: tools/ab_env.sh VAR "v1 v2" reps [bench args]
var=$1; vals=$2; reps=$3; shift 3
for i in $(seq $reps); do
  for v in $vals; do
    env $var=$v timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-dropin "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$var=$v', '%.4f' % d['roofline']['kernel_ms'])" || exit 1
  done
done
