#!/bin/bash
# Render kernel A/B on one box: bench.py with lidar_occupancy per variant in racing_dreamer_amd/lib/ab/, patch kernel time (GPU box)
lib=racing_dreamer_amd/lib/libracecar_hip.so; cp $lib /tmp/ab_patch_original.so
for r in 1 2 3; do for v in racing_dreamer_amd/lib/ab/*.so; do cp $v $lib
  python bench.py --no-cpu-baseline --no-ftg --no-configs --obs-type lidar_occupancy --steps 200 --warmup 20 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rep $r  %-28s render %.4f ms  scan %.4f  step %.4f ms' % ('$(basename $v .so)', d['kernels_ms']['rc_patch_kernel'], d['roofline']['avg_launch_ms'], d['ms_per_step']))"
done; done
cp /tmp/ab_patch_original.so $lib
