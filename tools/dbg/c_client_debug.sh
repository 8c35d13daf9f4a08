cd /root/repo
gcc -std=c99 -O2 -Wall -Iinclude examples/c_rollout.c -o /tmp/c_rollout -Lracing_dreamer_amd/lib -lracecar_hip -Wl,-rpath,$PWD/racing_dreamer_amd/lib -lm
( time timeout -k 5 100 /tmp/c_rollout 512 120 ) > gpurun_out/dbg_c.out 2>&1
echo "rc=$?" >> gpurun_out/dbg_c.out
tail -12 gpurun_out/dbg_c.out
( time timeout -k 5 100 /tmp/c_rollout 512 120 < /dev/null | cat ) > gpurun_out/dbg_c2.out 2>&1
tail -8 gpurun_out/dbg_c2.out
timeout 300 python -m pytest tests/test_gpu_api.py -k plain_c -x -q 2>&1 | tail -5
