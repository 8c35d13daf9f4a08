#!/bin/bash
# The bench lines of BASELINE.json's single-GPU configs (run on the GPU box); $1 = output directory
out=${1:-gpurun_out/configs}; mkdir -p $out
python bench.py > $out/bench_65536_austria_lidar.json 2> $out/err.log &&
python bench.py --obs-type lidar_occupancy --no-cpu-baseline > $out/bench_65536_austria_lidar_occupancy.json 2>> $out/err.log &&
python bench.py --envs 4096 --track columbia --steps 1000 --warmup 100 --no-cpu-baseline > $out/bench_4096_columbia_lidar.json 2>> $out/err.log &&
python bench.py --envs 32768 --cars 2 --track treitlstrasse_v2 --no-cpu-baseline > $out/bench_32768x2_treitlstrasse_v2.json 2>> $out/err.log &&
for t in barcelona gbr columbia; do python bench.py --track $t --no-cpu-baseline --steps 100 > $out/bench_65536_${t}_lidar.json 2>> $out/err.log; done
for f in $out/bench_*.json; do python -c "
import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f'.split('/')[-1], round(d['value']/1e6,1), 'M/s', round(d['ms_per_step'],4), d['kernels_ms'], 'frac', round(d['roofline']['frac'],4), 'r4', round(d['action_repeat_4']['env_steps_per_s']/1e6,1))"; done
