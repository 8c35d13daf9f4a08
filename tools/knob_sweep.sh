#!/bin/bash
# Sweep of the scan's experiment knobs (run on the GPU box): prints ms_per_step and kernel times.
#   bash tools/knob_sweep.sh            the default bench workload (65 536 envs, austria)
#   bash tools/knob_sweep.sh small      BASELINE.json configs[1] (4 096 envs, columbia): waves per car 1..6, 9, 17
if [ "$1" = small ]; then args="--envs 4096 --track columbia --steps 600 --warmup 60"; splits="1 2 3 4 5 6 9 17"; thrs="64"; else args="--steps 100 --warmup 10"; splits="1 2"; thrs="64 128 256"; fi
for split in $splits; do for thr in $thrs; do
  python bench.py --no-cpu-baseline --no-ftg --no-configs $args --debug-knob ray_split=$split --debug-knob ray_threads=$thr 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('split $split threads $thr', round(d['ms_per_step'],4), d['kernels_ms'])"
done; done
