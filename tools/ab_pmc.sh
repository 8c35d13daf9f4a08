#!/bin/bash
# SQ counters of the scan for every variant in racing_dreamer_amd/lib/ab/ (GPU box): bash tools/ab_pmc.sh > gpurun_out/ab_pmc.log
R=$(pwd); export TMPDIR=/tmp
export RC_ALLOW_STALE_LIBRARY=1      # (variant builds take the library's place: racing_dreamer_amd/_lib.py)
lib=racing_dreamer_amd/lib/libracecar_hip.so
cp $lib /tmp/ab_pmc_original.so
# whatever ends this script - Ctrl-C, a time-out, a failing step - the shipped library is put back
trap 'cp /tmp/ab_pmc_original.so $lib' EXIT INT TERM
for v in racing_dreamer_amd/lib/ab/*.so; do
  cp $v $lib
  echo "== $(basename $v .so)"
  i=0
  for group in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU" \
               "SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"; do
    i=$((i+1)); out=gpurun_out/abpmc_$i; rm -rf $out
    (cd /tmp && rocprofv3 --pmc $group -d $R/$out -o pmc -- python3 $R/bench.py --no-cpu-baseline --no-ftg --no-configs --steps 10 --warmup 2 > /dev/null 2> $R/gpurun_out/abpmc_err_$i.log) || { echo "rocprof failed"; tail -3 $R/gpurun_out/abpmc_err_$i.log; }
    python tools/rocpd_summary.py pmc $(find $out -name "*.db" | sort) | grep "raycast_car" | sed 's/"void (anonymous namespace):://; s/(Rc[^"]*"//'
    rm -rf $out
  done
done
cp /tmp/ab_pmc_original.so $lib
