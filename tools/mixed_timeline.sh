#!/bin/bash
# Kernel timeline of MixedTrackEnv steps (GPU box): rocprofv3 --kernel-trace of tools/mixed_trace.py, the last steps printed.
R=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out/mx
(cd /tmp && rocprofv3 --kernel-trace -d $R/gpurun_out/mx/kt -o kt -- python3 $R/tools/mixed_trace.py) 2>&1 | grep "ms per step"
python - <<PY
import sqlite3, glob
db=glob.glob("gpurun_out/mx/kt/**/*.db", recursive=True)[0]
c=sqlite3.connect(db)
rows=[r for r in c.execute("select name, start, end from kernels order by start") if "raycast" in r[0] or "dynamics" in r[0]]
last=rows[-2*30:]; t0=last[0][1]
for n,s,e in last[:8]:
    print("%-28s start %8.1f us  end %8.1f us  dur %7.1f us"%(n.split("::")[-1][:28], (s-t0)/1e3, (e-t0)/1e3, (e-s)/1e3))
PY
rm -rf gpurun_out/mx/kt
