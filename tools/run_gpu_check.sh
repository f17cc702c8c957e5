set -e
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/t.log 2>&1 || { tail -30 gpurun_out/t.log; exit 1; }
tail -2 gpurun_out/t.log
for args in "" "--kv-heads 8" "--kv-heads 8 --seqlen 131072 --topk 64"; do
  echo "### $args"
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --cpu-sample-batch 2 $args 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['achieved'], d['roofline']['frac'], d.get('parity'))"
done
