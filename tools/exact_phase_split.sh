#!/bin/bash
# what the prefilter costs without its recursions (fills and drains only) - timing experiment, results are wrong by construction
h=racing_dreamer_amd/csrc/racecar_patch_exact.h
cp $h /tmp/px_header_original.h
trap 'cp /tmp/px_header_original.h $h; python -m racing_dreamer_amd.build > /dev/null 2>&1' EXIT INT TERM
sed -i '1i #define PX_EXP_NO_CHAIN 1' $h
python -m racing_dreamer_amd.build > /dev/null 2>&1 || { echo build failed; exit 1; }
echo "no recursions: $(python tools/time_exact_render.py 16384 2>/dev/null | tail -1)"
