#!/bin/bash
# SQ / cache counters of the default bench workload's kernels (run on the GPU box): bash tools/pmc_quick.sh [outdir]
# BENCH_ARGS='--obs-type lidar_occupancy' KERNEL=patch bash tools/pmc_quick.sh   for another config / kernel
out=${1:-gpurun_out/pmcq}; R=$(pwd); rm -rf $out; mkdir -p $out; export TMPDIR=/tmp; i=0
for group in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" \
             "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VALU" \
             "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $group -d $R/$out/pass_$i -o pmc -- python3 $R/bench.py --no-cpu-baseline --no-ftg --no-configs --steps 10 --warmup 2 $BENCH_ARGS > /dev/null 2> $R/$out/err_$i.log) || exit 1
done
python tools/rocpd_summary.py pmc $(find $out -name "*.db" | sort) | grep "${KERNEL:-raycast_car}" | sed 's/"void (anonymous namespace):://; s/(Rc[^"]*"//'
rm -rf $out/pass_*
