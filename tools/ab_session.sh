#!/bin/bash
# One A/B session on the GPU box: parity of every variant in racing_dreamer_amd/lib/ab/, then tools/ab_bench.sh.
#   bash tools/ab_session.sh [reps] [bench args...] > gpurun_out/ab.log
export RC_ALLOW_STALE_LIBRARY=1      # (variant builds take the library's place: racing_dreamer_amd/_lib.py)
lib=racing_dreamer_amd/lib/libracecar_hip.so
cp $lib /tmp/ab_session_original.so
# whatever ends this script - Ctrl-C, a time-out, a failing step - the shipped library is put back
trap 'cp /tmp/ab_session_original.so $lib' EXIT INT TERM
for v in racing_dreamer_amd/lib/ab/*.so; do
  cp $v $lib
  timeout -k 10 300 python tools/ab_check.py $(basename $v .so) 2>&1 | grep ab_check || echo "ab_check $(basename $v .so): FAILED"
done
cp /tmp/ab_session_original.so $lib
bash tools/ab_bench.sh "$@"
