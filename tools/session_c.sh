#!/bin/bash
out=gpurun_out/r3c; mkdir -p $out
for i in 1 2 3; do
  timeout -k 10 300 python bench.py --gpus 2 --backend gloo --gather-via p2p --steps 6 --warmup 2 --envs 2048 > $out/p2p_$i.out 2> $out/p2p_$i.err; echo "run $i rc=$?"
  grep -n "Error\|error\|Traceback" $out/p2p_$i.err | head -5
done
timeout -k 10 600 python -m pytest tests/test_gpu_gather.py -m gpu -x -q 2>&1 | tail -5
