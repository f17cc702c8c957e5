#!/bin/bash
# kernel time of the bench on launch shapes around the planner's choices (geometry, items scored ahead, queue)
for cfg in "" "--batch 128" "--seqlen 16384" "--batch 128 --seqlen 16384" "--batch 32" "--batch 16" "--seqlen 8192 --batch 256" "--batch 100" "--kv int8 --seqlen 65536"; do
  echo "### $cfg"
  timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-dropin $cfg 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('kernel_ms %.4f frac %.3f' % (d['roofline']['kernel_ms'], d['roofline']['frac']))" || exit 1
done
