#!/bin/bash
# rehearsals of the multi-GPU path on one GPU: RCCL with a one-rank group, gloo with two ranks on the device (RCCL refuses
# two ranks on one device), each with and without the collective; then the launch-shape table
set -o pipefail
run() { name=$1; shift; echo "### $name: $*"; env "$@" > gpurun_out/$name.json 2> gpurun_out/$name.err || { grep -v "amdgpu.ids\|socket.cpp" gpurun_out/$name.err | tail -25; return 1; }; python -c "
import json,sys
d=json.load(open('gpurun_out/$name.json')); r=d['ranks']
print('ms/step %.4f kernel_ms %s backend %s collective_ms %s submitted %s' % (d['ms_per_step'], ['%.4f' % x for x in r['kernel_ms_per_rank']], r['backend'], r.get('collective_ms'), r.get('collectives_submitted')))"; }
run r3_rccl1_c3 BENCH_WATCHDOG=150 BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --no-cpu-baseline || exit 1
run r3_rccl1_c3_single BENCH_SINGLE_OUTPUTS=1 BENCH_WATCHDOG=150 BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --no-cpu-baseline || exit 1
run r3_rccl1_c3_nocoll BENCH_WATCHDOG=150 BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --no-cpu-baseline --allreduce-every 0 || exit 1
run r3_rccl1_c3_b BENCH_WATCHDOG=150 BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --no-cpu-baseline || exit 1
run r3_rccl1_c3_nocoll_b BENCH_WATCHDOG=150 BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --no-cpu-baseline --allreduce-every 0 || exit 1
run r3_n1_c5_b8 timeout -k 10 300 python bench.py --workload c5 --batch 8 --no-cpu-baseline --no-dropin || exit 1
run r3_gloo2_c5_b8_nocoll BENCH_WATCHDOG=150 BENCH_SINGLE_DEVICE=1 BENCH_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --workload c5 --batch 8 --no-cpu-baseline --allreduce-every 0 || exit 1
run r3_gloo2_c5_b8 BENCH_WATCHDOG=150 BENCH_SINGLE_DEVICE=1 BENCH_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --workload c5 --batch 8 --no-cpu-baseline || exit 1
bash tools/shapes_check.sh > gpurun_out/r3_shapes0.log 2>&1; cat gpurun_out/r3_shapes0.log
