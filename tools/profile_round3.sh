#!/bin/bash
# Measurement evidence (run on the GPU box from the repository root): everything lands in gpurun_out/prof_r03/ as small
# summaries, to be copied into profiles/.  Parts: bench lines (c3 with the drop-in leg, c4, c5, rehearsals of the ranks),
# rocprofv3 kernel stats of every launch shape the design notes quote, HBM traffic counters (FETCH_SIZE and WRITE_SIZE in
# separate passes) for c3 / c4 / c5, matrix-core counters of the MFMA-form attention kernel and of the scoring kernels,
# the SQ / TCP / TCC comparison of c3 and c4, the compiled caller's repeated launch.
tag=r03
root=$(pwd)
out=$root/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
B="$root/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-dropin"
step() { echo "== $*" >> $out/progress.log; }

# ---- bench lines
step bench
timeout -k 10 400 python3 bench.py > $out/bench_c3.json 2> $out/bench_c3.err || exit 1
timeout -k 10 400 python3 bench.py --workload c4 > $out/bench_c4.json 2> $out/bench_c4.err || exit 1
timeout -k 10 400 python3 bench.py --workload c5 > $out/bench_c5.json 2> $out/bench_c5.err || exit 1
BENCH_FORCE_DIST=1 timeout -k 10 300 python3 bench.py --workload c5 --no-cpu-baseline > $out/bench_c5_rccl_world1.json 2> $out/rccl1.err || exit 1
BENCH_FORCE_DIST=1 timeout -k 10 300 python3 bench.py --workload c5 --no-cpu-baseline --allreduce-every 1 > $out/bench_c5_rccl_world1_every_step.json 2>> $out/rccl1.err || exit 1
BENCH_SINGLE_DEVICE=1 BENCH_DIST_BACKEND=gloo timeout -k 10 300 python3 bench.py --gpus 2 --workload c5 --batch 8 --no-cpu-baseline > $out/bench_c5_b8_gloo_2ranks.json 2> $out/gloo2.err || exit 1
BENCH_SINGLE_DEVICE=1 BENCH_DIST_BACKEND=gloo timeout -k 10 300 python3 bench.py --gpus 2 --workload c5 --batch 8 --no-cpu-baseline --allreduce-every 1 > $out/bench_c5_b8_gloo_2ranks_every_step.json 2>> $out/gloo2.err || exit 1
timeout -k 10 300 python3 bench.py --workload c5 --batch 8 --no-cpu-baseline --no-dropin > $out/bench_c5_b8_1rank.json 2>/dev/null || exit 1
bash tools/shapes_check.sh > $out/shapes.log 2>&1
TOKEN_ROWS=kv timeout -k 10 200 python3 tools/token_bench.py --kv-heads 8 > $out/token_bench_gqa.log 2>&1
timeout -k 10 200 python3 tools/token_bench.py > $out/token_bench_mha.log 2>&1
TOKEN_ROWS=kv timeout -k 10 200 python3 tools/token_bench.py --kv-heads 8 --kv int8 --seqlen 65536 > $out/token_bench_gqa_int8_s64k.log 2>&1
TOKEN_ROWS=kv timeout -k 10 200 python3 tools/token_bench.py --kv-heads 8 --seqlen 131072 > $out/token_bench_gqa_s128k.log 2>&1
timeout -k 10 120 python3 tools/cent_only.py > $out/cent_only.log 2>&1

cd /tmp && export TMPDIR=/tmp
# ---- kernel stats per launch shape
stats() { name=$1; shift; step stats $name; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/raw_$name -- python3 $B "$@" > $out/stats_$name.log 2>&1 && python3 $root/tools/pmc_summary.py stats $out/raw_$name $out/${tag}_${name}_kernel_stats.csv; rm -rf $out/raw_$name; }
stats bench_c3
stats bench_c4_int8 --kv int8 --seqlen 65536
stats bench_c5_share --kv-heads 8 --seqlen 131072 --topk 64
stats gqa_s32k --kv-heads 8
stats gqa_int8_s64k --kv-heads 8 --kv int8 --seqlen 65536
stats b128 --batch 128
stats s16k --seqlen 16384
stats b128_s16k --batch 128 --seqlen 16384
stats b32 --batch 32
stats b16 --batch 16
stats b256_s8k --seqlen 8192 --batch 256
stats b100 --batch 100
step stats cent
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/raw_cent -- python3 $root/tools/cent_only.py --steps 5 > $out/stats_cent.log 2>&1 && python3 $root/tools/pmc_summary.py stats $out/raw_cent $out/${tag}_cent_ops_kernel_stats.csv; rm -rf $out/raw_cent
# the compiled caller: five launches on one executor and one workspace
step caller
python3 $root/tools/make_caller_inputs.py $out/caller_in > /dev/null 2>&1
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $out/raw_caller -- $root/tests/cpp/ref_caller $out/caller_in > $out/caller.log 2>&1 && python3 $root/tools/pmc_summary.py stats $out/raw_caller $out/${tag}_ref_caller_kernel_stats.csv; rm -rf $out/raw_caller $out/caller_in
# the drop-in Python call, repeated (one kernel per call after the first)
# ---- counters, one set per pass
pmc() { name=$1; ctr=$2; shift 2; step pmc $name; timeout -s KILL 300 rocprofv3 --pmc $ctr --output-format csv -d $out/pmc_$name -- python3 "$@" > $out/pmc_$name.log 2>&1; }
pmc c3_fetch FETCH_SIZE $B
pmc c3_write WRITE_SIZE $B
pmc c3_l2 "TCC_HIT_sum TCC_MISS_sum" $B
pmc c4_fetch FETCH_SIZE $B --kv int8 --seqlen 65536
pmc c4_write WRITE_SIZE $B --kv int8 --seqlen 65536
pmc c5_fetch FETCH_SIZE $B --kv-heads 8 --seqlen 131072 --topk 64
pmc c5_write WRITE_SIZE $B --kv-heads 8 --seqlen 131072 --topk 64
pmc c5_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVES" $B --kv-heads 8 --seqlen 131072 --topk 64
pmc cent_mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVES" $root/tools/cent_only.py --steps 5
pmc cent_l2 "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" $root/tools/cent_only.py --steps 5
cd $root
args=""
for n in c3_fetch c3_write c3_l2 c4_fetch c4_write c5_fetch c5_write c5_mfma cent_mfma cent_l2; do args="$args $n=$out/pmc_$n"; done
python3 tools/pmc_summary.py pmc $out/${tag}_pmc.json $args
rm -rf $out/pmc_c3_* $out/pmc_c4_* $out/pmc_c5_* $out/pmc_cent_mfma $out/pmc_cent_l2
# ---- SQ / TCP / TCC comparison of c3 (f16) and c4 (int8)
step compare
bash tools/pmc_compare.sh c3_c4 "" "--kv int8 --seqlen 65536" > $out/pmc_compare.log 2>&1 && cp gpurun_out/pmc_c3_c4/summary.json $out/${tag}_pmc_compare_c3_c4.json
step done
ls -la $out
