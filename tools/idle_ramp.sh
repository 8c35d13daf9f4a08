#!/bin/bash
R=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out/sh; rm -rf gpurun_out/sh/kt
(cd /tmp && rocprofv3 --kernel-trace -d $R/gpurun_out/sh/kt -o kt -- python3 $R/tools/idle_ramp_trace.py) > /dev/null 2>&1
python - <<PY
import sqlite3, glob
db=glob.glob("gpurun_out/sh/kt/**/*.db", recursive=True)[0]
c=sqlite3.connect(db)
scan=[(s,e) for nm,s,e in c.execute("select name, start, end from kernels order by start") if "raycast_car" in nm and "false, false" in nm]
t0=scan[0][0]
for i,(s,e) in enumerate(scan):
    if i % 10 == 0 or (i and s - scan[i-1][1] > 1e8): print("%3d start %9.1f ms dur %6.1f us"%(i,(s-t0)/1e6,(e-s)/1e3))
PY
rm -rf gpurun_out/sh/kt
