#!/bin/bash
# A/B timing of library builds ON ONE BOX in ONE call (run on the GPU box): box-to-box spread of the scan time is about
# +-1.5 %, more than most single changes are worth, so variants are compared by alternating them on the same device.
#   build each variant, copy its libracecar_hip.so to racing_dreamer_amd/lib/ab/<name>.so, then
#   bash tools/ab_bench.sh [reps] [bench args...]      prints the scan time of every variant, `reps` times in turn
reps=${1:-3}; shift
export RC_ALLOW_STALE_LIBRARY=1      # (variant builds take the library's place: racing_dreamer_amd/_lib.py)
lib=racing_dreamer_amd/lib/libracecar_hip.so
cp $lib /tmp/ab_original.so
# whatever ends this script - Ctrl-C, a time-out, a failing step - the shipped library is put back
trap 'cp /tmp/ab_original.so $lib' EXIT INT TERM
for r in $(seq $reps); do
  for v in racing_dreamer_amd/lib/ab/*.so; do
    cp $v $lib
    python bench.py --no-cpu-baseline --no-ftg --no-configs --steps 300 --warmup 30 "$@" 2>/dev/null | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rep $r  %-28s scan %.4f ms  step %.4f ms' % ('$(basename $v .so)', d['roofline']['avg_launch_ms'], d['ms_per_step']))"
  done
done
cp /tmp/ab_original.so $lib
