#!/bin/bash
# The exact render's sample kernel against its tile size (PX_T x PX_T pixels per staged tile; LDS per workgroup falls with it, so
# more workgroups - more waves to hide the binary64 chains - fit a CU).  Builds each variant ON the GPU box and times it; the tree's
# header is put back on any exit.   bash tools/exact_tile_sweep.sh
h=racing_dreamer_amd/csrc/racecar_patch_exact.h
cp $h /tmp/px_header_original.h
trap 'cp /tmp/px_header_original.h $h; python -m racing_dreamer_amd.build > /dev/null 2>&1' EXIT INT TERM
for cfg in "40 64 65" "25 41 41" "20 34 35" "10 20 21"; do
  set -- $cfg
  sed -i "s/^#define PX_T .*/#define PX_T $1/; s/^#define PX_TILE_N .*/#define PX_TILE_N $2/; s/^#define PX_TILE_PITCH .*/#define PX_TILE_PITCH $3/" $h
  python -m racing_dreamer_amd.build > /dev/null 2>&1 || { echo "build failed for PX_T $1"; continue; }
  echo "PX_T $1 (tile $2, pitch $3): $(python tools/time_exact_render.py 16384 2>/dev/null | tail -1)"
done
