#!/bin/bash
# Kernel timeline of the N > 1 headline's loop with one rank (GPU box): the last steps, every kernel with its queue.
R=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out/sh; rm -rf gpurun_out/sh/kt
(cd /tmp && rocprofv3 --kernel-trace -d $R/gpurun_out/sh/kt -o kt -- python3 $R/tools/sharded_trace.py) 2>&1 | grep "ms per step"
python - <<PY
import sqlite3, glob
db=glob.glob("gpurun_out/sh/kt/**/*.db", recursive=True)[0]
c=sqlite3.connect(db)
cols=[r[1] for r in c.execute("pragma table_info(kernels)")]
print(cols)
q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows=list(c.execute(f"select name, start, end{', ' + q if q else ''} from kernels order by start"))
import os
n=int(os.environ.get("TRACE_STEPS", "40"))
scan=[i for i,r in enumerate(rows) if "raycast" in r[0]]
i0=scan[-n-4]; t0=rows[scan[-n-4]][1]          # (the window's steps, then 4 of the closing pass)
prev_end={}
for r in rows[i0:]:
    nm,s,e=r[0],r[1],r[2]
    short=nm.replace("void ","").replace("(anonymous namespace)::","").replace("at::native::","")[:40]
    print("%-40s q %-3s start %9.1f us  dur %7.1f us"%(short, r[3] if q else "-", (s-t0)/1e3, (e-s)/1e3))
PY
rm -rf gpurun_out/sh/kt
