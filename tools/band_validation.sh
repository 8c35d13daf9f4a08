#!/bin/bash
# Sensitivity check of the parity tests to the scan's exact-count band (run on the GPU box): with the band shrunk
# below the rounding bound (racecar_kernels.hip, cast_ray_rects) the corner-aimed test must fail; at the shipped
# width (max(w, h) * 2^-21) it passes.
for l2 in -40 -26 -24 -22 -21; do
  echo "== band log2 $l2"
  RC_TEST_BAND_LOG2=$l2 python -m pytest tests/test_gpu_parity.py -q -k "aimed_at_wall_corners or 2040" 2>&1 | grep -E "passed|failed|^E  +assert|AssertionError" | cut -c1-220
done
