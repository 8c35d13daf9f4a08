#!/bin/bash
# 4 096 envs on columbia (BASELINE.json configs[1]): waves per car x one / two rays per lane (run on the GPU box)
for f in 1 0; do for split in 2 3 4 5 6 9; do
  python bench.py --no-cpu-baseline --no-ftg --no-configs --envs 4096 --track columbia --steps 600 --warmup 60 --debug-knob ray_split=$split --debug-knob scan_flags=$f 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('two rays per lane:', 'no ' if $f else 'yes', ' waves per car $split ', round(d['ms_per_step'],4), d['kernels_ms'])"
done; done
