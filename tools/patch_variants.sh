#!/bin/bash
# lidar_occupancy render experiments (run on the GPU box): run layout x store flavour of rc_patch_kernel
for v in 0 1 2 4 5; do
  python bench.py --no-cpu-baseline --no-ftg --no-configs --obs-type lidar_occupancy --steps 60 --warmup 10 --debug-knob patch_variant=$v 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('patch_variant $v (bit0 row-major, bit1 plain stores, bit2 LDS-transposed stores)', round(d['ms_per_step'],4), d['kernels_ms'])"
done
