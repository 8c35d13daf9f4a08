#!/bin/bash
# The exact render's prefilter against the lines a wave filters at a time (PX_NBW: LDS per workgroup = 4 x NBW x 221 x 8 B + 7 KB, so
# workgroups per CU; lines in flight per CU stay ~64, the instructions issued per line fall with NBW).  Builds each variant ON the
# GPU box; the tree's header is put back on any exit.   bash tools/exact_nbw_sweep.sh
h=racing_dreamer_amd/csrc/racecar_patch_exact.h
cp $h /tmp/px_header_original.h
trap 'cp /tmp/px_header_original.h $h; python -m racing_dreamer_amd.build > /dev/null 2>&1' EXIT INT TERM
for cfg in "4 10" "4 8" "2 16" "1 32" "5 8" "2 20" "8 5"; do set -- $cfg; nbw=$2
  sed -i "s/^#define PX_NBW .*/#define PX_NBW $2/; s/^#define PX_PF_WAVES .*/#define PX_PF_WAVES $1/" $h
  python -m racing_dreamer_amd.build > /dev/null 2>&1 || { echo "build failed for PX_NBW $nbw"; continue; }
  echo "waves $1, lines per wave $2: $(python tools/time_exact_render.py 16384 2>/dev/null | tail -1)"
done
