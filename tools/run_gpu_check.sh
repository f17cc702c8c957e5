#!/bin/bash
# GPU regression pass: parity tests, then the bench on the headline and GQA shapes (ms/step, tokens/s, GB/s, frac, kernel ms).
set -e
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/t.log 2>&1 || { tail -30 gpurun_out/t.log; exit 1; }
tail -2 gpurun_out/t.log
run() {
  echo "### $*"
  timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-dropin "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ms/step %.4f  tok/s %.0f  GB/s %.0f  frac %.3f  kernel_ms %.4f' % (d['ms_per_step'], d['value'], r['achieved'], r['frac'], r['kernel_ms']))"
}
run
run
run --kv-heads 8
run --kv-heads 8 --seqlen 131072 --topk 64
run --kv int8 --seqlen 65536
run --kv-heads 8 --kv int8 --seqlen 65536
if [ "$1" = "small" ]; then
  run --batch 1
  run --batch 8
  run --batch 8 --kv-heads 8
fi
