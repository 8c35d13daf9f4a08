#!/bin/bash
# Sweep of the scan's experiment knobs on the default bench workload (run on the GPU box): prints ms_per_step and kernel times.
for split in 1 2; do for thr in 64 128 256; do
  python bench.py --debug-knob ray_split=$split --debug-knob ray_threads=$thr --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('split $split threads $thr', round(d['ms_per_step'],4), d['kernels_ms'])"
done; done
