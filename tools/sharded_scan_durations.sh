#!/bin/bash
# Scan duration per step over a window of the sharded loop, one rank (GPU box): does the scan get faster as the window goes on?
R=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out/sh; rm -rf gpurun_out/sh/kt
(cd /tmp && TRACE_STEPS=${1:-120} rocprofv3 --kernel-trace -d $R/gpurun_out/sh/kt -o kt -- python3 $R/tools/sharded_trace.py) 2>&1 | grep "ms per step"
python - <<PY
import sqlite3, glob, os
db=glob.glob("gpurun_out/sh/kt/**/*.db", recursive=True)[0]
c=sqlite3.connect(db)
rows=list(c.execute("select name, start, end, queue_id from kernels order by start"))
n=int(os.environ.get("TRACE_STEPS", "${1:-120}"))
scan=[(s,e) for nm,s,e,q in rows if "raycast" in nm][-n-4-int(os.environ.get("BEFORE", "0")):]
copies=[(s,e) for nm,s,e,q in rows if "copyBuffer" in nm]
t0=scan[0][0]
out=[]
for i,(s,e) in enumerate(scan):
    ov=[(cs,ce) for cs,ce in copies if cs<e and ce>s]
    out.append("%3d start %8.1f us dur %6.1f us  copy kernels beside it: %s"%(i,(s-t0)/1e3,(e-s)/1e3, ", ".join("%.0f us"%((ce-cs)/1e3) for cs,ce in ov)))
print("\n".join(out[::int(os.environ.get("EVERY", "1"))]))
PY
rm -rf gpurun_out/sh/kt
