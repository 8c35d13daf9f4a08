#!/bin/bash
# rocprofv3 evidence for the exact render (obs_type lidar_occupancy_reference): kernel-trace stats, then FETCH_SIZE / WRITE_SIZE in
# passes of their own.   bash tools/profile_exact_render.sh gpurun_out/prof_exact
out=${1:-gpurun_out/prof_exact}; R=$(pwd); mkdir -p $out; export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$out/kt -o kt -- python3 $R/tools/time_exact_render.py 16384 > $R/$out/time_exact.txt 2> $R/$out/err_kt.log) || { tail -3 $out/err_kt.log; exit 1; }
db=$(find $out/kt -name "*.db" | head -1)
python tools/rocpd_summary.py stats $db > $out/kernel_stats_exact_render_16384_austria.csv
rm -rf $out/kt
i=0
for group in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $group -d $R/$out/pass_$i -o pmc -- python3 $R/tools/time_exact_render.py 16384 > /dev/null 2> $R/$out/err_pass_$i.log) || { tail -3 $out/err_pass_$i.log; exit 1; }
done
python tools/rocpd_summary.py pmc $(find $out -path "*pass_*" -name "*.db" | sort) | grep -i "exact\|kernel" > $out/pmc_counters_exact_render_16384_austria.csv
rm -rf $out/pass_*
cat $out/time_exact.txt | tail -2; cat $out/kernel_stats_exact_render_16384_austria.csv | head -8; cat $out/pmc_counters_exact_render_16384_austria.csv | head -8
