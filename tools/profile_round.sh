#!/bin/bash
# rocprofv3 evidence for one round: kernel stats of the bench command, HBM traffic counters (separate passes) and
# MFMA counters of the standalone scoring kernels.  Usage (on the GPU box): tools/profile_round.sh r01
set -e
tag=${1:-r02}
root=$(pwd)
out=$root/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="$root/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-dropin"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_c3 -- python3 $B > $out/stats_c3.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_c5 -- python3 $B --kv-heads 8 --seqlen 131072 --topk 64 > $out/stats_c5.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_c4 -- python3 $B --kv int8 --seqlen 65536 > $out/stats_c4.log 2>&1
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | tr ' ' '_')
  timeout -s KILL 300 rocprofv3 --pmc $c --output-format csv -d $out/pmc_c3_$n -- python3 $B > $out/pmc_c3_$n.log 2>&1
done
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_c5_FETCH_SIZE -- python3 $B --kv-heads 8 --seqlen 131072 --topk 64 > $out/pmc_c5_FETCH.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_c4_FETCH_SIZE -- python3 $B --kv int8 --seqlen 65536 > $out/pmc_c4_FETCH.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVES --output-format csv -d $out/pmc_cent_mfma -- python3 $root/tools/cent_only.py --steps 5 > $out/pmc_cent.log 2>&1
timeout -s KILL 300 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_cent_l2 -- python3 $root/tools/cent_only.py --steps 5 > $out/pmc_cent_l2.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_cent -- python3 $root/tools/cent_only.py --steps 5 > $out/stats_cent.log 2>&1
cd $root
python3 tools/pmc_summary.py stats $out/stats_c3 $out/${tag}_bench_c3_kernel_stats.csv
python3 tools/pmc_summary.py stats $out/stats_c5 $out/${tag}_bench_c5_share_kernel_stats.csv
python3 tools/pmc_summary.py stats $out/stats_c4 $out/${tag}_bench_c4_int8_kernel_stats.csv
python3 tools/pmc_summary.py stats $out/stats_cent $out/${tag}_cent_ops_kernel_stats.csv
python3 tools/pmc_summary.py pmc $out/${tag}_pmc.json c3_fetch=$out/pmc_c3_FETCH_SIZE c3_write=$out/pmc_c3_WRITE_SIZE c3_l2=$out/pmc_c3_TCC_HIT_sum_TCC_MISS_sum c5_fetch=$out/pmc_c5_FETCH_SIZE c4_fetch=$out/pmc_c4_FETCH_SIZE cent_mfma=$out/pmc_cent_mfma cent_l2=$out/pmc_cent_l2
ls -la $out/*.csv $out/*.json
# raw rocprof trees are large: keep only the summaries in gpurun_out
rm -rf $out/stats_c3 $out/stats_c4 $out/stats_c5 $out/stats_cent $out/pmc_c3_* $out/pmc_c5_* $out/pmc_c4_* $out/pmc_cent_mfma $out/pmc_cent_l2
