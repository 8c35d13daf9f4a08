#!/bin/bash
# bench.py --gpus 2 over gloo on one GPU with the other headlines / transports / staging (functional sweep; GPU box)
for extra in "--mixed-tracks" "--gather full-u16" "--gather none" "--gather-via abi" "--gather summary --gather-every 4"; do
  python bench.py --gpus 2 --backend gloo --steps 6 --warmup 2 --envs 1024 $extra 2>gpurun_out/n2.err | python -c "
import sys, json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('$extra ->', d['config']['gather'], d['config']['track'][:20], round(d['ms_per_step'],3), 'check', d.get('gather_check',{}).get('ok'), sorted(d.get('gather_modes',{})), 'aborted' in d, d.get('leg_errors'))" || { echo "FAILED: $extra"; tail -5 gpurun_out/n2.err; }
done
for extra in "--obs-type lidar_occupancy" "--cars 2 --track treitlstrasse_v2" "--repeat 4" "--gather-via p2p --gather full" "--no-gather-modes --no-gather-check"; do
  python bench.py --gpus 2 --backend gloo --steps 6 --warmup 2 --envs 1024 $extra 2>gpurun_out/n2.err | python -c "
import sys, json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('$extra ->', d['config']['gather'], round(d['ms_per_step'],3), 'check', d.get('gather_check',{}).get('ok'), sorted(d.get('gather_modes',{})), 'aborted' in d, d.get('leg_errors'))" || { echo "FAILED: $extra"; grep -n "Error" -B6 gpurun_out/n2.err | head -30; }
done
python bench.py --envs 2048 --cars 2 --track treitlstrasse_v2 --obs-type lidar_occupancy --repeat 4 --steps 6 --warmup 2 --no-cpu-baseline 2>gpurun_out/n1.err | python -c "
import sys, json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('N=1 2 cars occupancy repeat 4:', round(d['ms_per_step'],3), d['kernels_ms'], d.get('leg_errors'), [t['track'] for t in d['tracks']])" || tail -5 gpurun_out/n1.err
