#!/bin/bash
# rocprofv3 evidence for profiles/ (run on the GPU box from the repo root): kernel-trace stats of the bench command for
# the single-GPU configs, then PMC counters of the default workload in separate passes (no tracing combined with --pmc).
#   bash tools/profile_round.sh gpurun_out/prof_x
out=${1:-gpurun_out/prof}; R=$(pwd); mkdir -p $out; export TMPDIR=/tmp
run_stats() {   # name, bench args...
  name=$1; shift
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $R/$out/kt_$name -o kt -- python3 $R/bench.py --no-cpu-baseline --no-ftg --no-configs "$@" > $R/$out/bench_prof_$name.json 2> $R/$out/err_$name.log)
  db=$(find $out/kt_$name -name "*.db" | head -1)
  python tools/rocpd_summary.py stats $db > $out/kernel_stats_$name.csv && python tools/rocpd_summary.py gaps $db > $out/gaps_$name.csv
  rm -rf $out/kt_$name
}
run_stats 65536_austria_lidar &&
run_stats 65536_austria_lidar_occupancy --obs-type lidar_occupancy &&
run_stats 32768x2_treitlstrasse_v2 --envs 32768 --cars 2 --track treitlstrasse_v2 &&
run_stats 4096_columbia_lidar --envs 4096 --track columbia --steps 1000 --warmup 100 || exit 1
i=0
for group in "FETCH_SIZE" "WRITE_SIZE" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" \
             "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
             "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $group -d $R/$out/pass_a$i -o pmc -- python3 $R/bench.py --no-cpu-baseline --no-ftg --no-configs --steps 10 --warmup 2 > /dev/null 2> $R/$out/err_pass_a$i.log) || exit 1
done
python tools/rocpd_summary.py pmc $(find $out -path "*pass_a*" -name "*.db" | sort) > $out/pmc_counters_65536_austria.csv
# the same FETCH/WRITE passes for the lidar_occupancy config (patch kernel traffic)
for group in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $group -d $R/$out/pass_b$i -o pmc -- python3 $R/bench.py --no-cpu-baseline --no-ftg --no-configs --obs-type lidar_occupancy --steps 10 --warmup 2 > /dev/null 2> $R/$out/err_pass_b$i.log) || exit 1
done
python tools/rocpd_summary.py pmc $(find $out -path "*pass_b*" -name "*.db" | sort) > $out/pmc_counters_65536_austria_lidar_occupancy.csv
rm -rf $out/pass_a* $out/pass_b*
ls -la $out
