#!/bin/bash
# Counter comparison of two bench shapes (diagnostic): tools/pmc_compare.sh tag "bench flags A" "bench flags B"
# Three counter sets, one rocprofv3 pass each per shape (<= 8 SQ, 4 TCP + 2 TA, 4 TCC counters per pass).
set -e
tag=$1; fa=$2; fb=$3
root=$(pwd)
out=$root/gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="$root/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-dropin"
sets=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VALU SQ_INSTS_VMEM_RD"
      "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum"
      "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_BUSY_sum")
args=""
for i in 0 1 2; do
  timeout -s KILL 120 rocprofv3 --pmc ${sets[$i]} --output-format csv -d $out/a$i -- python3 $B $fa > $out/a$i.log 2>&1
  timeout -s KILL 120 rocprofv3 --pmc ${sets[$i]} --output-format csv -d $out/b$i -- python3 $B $fb > $out/b$i.log 2>&1
  args="$args a$i=$out/a$i b$i=$out/b$i"
done
cd $root
python3 tools/pmc_summary.py pmc $out/summary.json $args
rm -rf $out/a? $out/b?
