#!/bin/bash
# first GPU pass of the round: parity suite, bench with the drop-in leg, two-rank rehearsals, launch-shape table
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r3_t1.log 2>&1
rc=$?; tail -15 gpurun_out/r3_t1.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py > gpurun_out/r3_bench1.json 2> gpurun_out/r3_bench1.err || { tail -20 gpurun_out/r3_bench1.err; exit 1; }
cat gpurun_out/r3_bench1.json
BENCH_SINGLE_DEVICE=1 BENCH_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --workload c5 --batch 8 --no-cpu-baseline > gpurun_out/r3_gloo2_c5.json 2> gpurun_out/r3_gloo2_c5.err || { tail -20 gpurun_out/r3_gloo2_c5.err; exit 1; }
cat gpurun_out/r3_gloo2_c5.json
BENCH_SINGLE_DEVICE=1 BENCH_DIST_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --workload c5 --batch 8 --no-cpu-baseline --allreduce-every 0 > gpurun_out/r3_gloo2_c5_nocoll.json 2> gpurun_out/r3_gloo2_c5_nocoll.err || { tail -20 gpurun_out/r3_gloo2_c5_nocoll.err; exit 1; }
cat gpurun_out/r3_gloo2_c5_nocoll.json
timeout -k 10 300 python bench.py --workload c5 --batch 8 --no-cpu-baseline > gpurun_out/r3_c5_b8.json 2>/dev/null; cat gpurun_out/r3_c5_b8.json
bash tools/shapes_check.sh > gpurun_out/r3_shapes0.log 2>&1; cat gpurun_out/r3_shapes0.log
