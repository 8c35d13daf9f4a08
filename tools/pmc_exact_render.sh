#!/bin/bash
# SQ / LDS counters of the exact render's two kernels (GPU box), three --pmc passes of tools/time_exact_render.py 16384:
#   bash tools/pmc_exact_render.sh
out=gpurun_out/pmc_exact; R=$(pwd); rm -rf $out; mkdir -p $out; export TMPDIR=/tmp; i=0
for group in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" \
             "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU" \
             "SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $group -d $R/$out/pass_$i -o pmc -- python3 $R/tools/time_exact_render.py 16384 > /dev/null 2> $R/$out/err_$i.log) || { tail -3 $out/err_$i.log; exit 1; }
done
python tools/rocpd_summary.py pmc $(find $out -name "*.db" | sort) | grep -i "exact"
rm -rf $out/pass_*
