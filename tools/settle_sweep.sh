#!/bin/bash
# The N = 1 headline at the driver's flags and at 200 steps for several --settle values (GPU box).   bash tools/settle_sweep.sh [values...]
for s in ${@:-0 20 150 300}; do
python bench.py --no-cpu-baseline --no-ftg --no-configs --settle $s --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('settle $s steps 20:', round(d['value']/1e6,1), 'M', round(d['ms_per_step'],4), d['kernels_ms'])"
python bench.py --no-cpu-baseline --no-ftg --no-configs --settle $s 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('settle $s steps 200:', round(d['value']/1e6,1), 'M', round(d['ms_per_step'],4), d['kernels_ms'])"
done
