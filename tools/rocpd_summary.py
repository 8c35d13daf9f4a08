"""Summaries of rocprofv3 rocpd (.db) output, in the layout of its CSV reports.

    python tools/rocpd_summary.py stats <results.db>      # kernel_stats.csv columns
    python tools/rocpd_summary.py pmc <results.db> ...    # kernel, counter, mean value per dispatch (summed over instances)
"""
import sqlite3
import sys
from collections import defaultdict


def stats(db):
    c = sqlite3.connect(db)
    rows = defaultdict(list)
    for name, dur in c.execute("select name, duration from kernels"):
        rows[name].append(dur)
    total = sum(sum(v) for v in rows.values())
    out = ['"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"']
    for name, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
        n, s = len(v), sum(v)
        mean = s / n
        sd = (sum((x - mean) ** 2 for x in v) / (n - 1)) ** 0.5 if n > 1 else 0.0
        out.append(f'"{name}",{n},{s},{mean:.6f},{100.0 * s / total:.2f},{min(v)},{max(v)},{sd:.6f}')
    return "\n".join(out)


def pmc(dbs):
    acc = defaultdict(lambda: defaultdict(float))
    for db in dbs:
        c = sqlite3.connect(db)
        for kernel, disp, counter, value in c.execute(
                "select kernel_name, dispatch_id, counter_name, value from counters_collection"):
            acc[(kernel, counter)][(db, disp)] += value          # one row per hardware instance: sum them
    out = ["kernel,counter,mean_per_dispatch,dispatches"]
    for (kernel, counter), d in sorted(acc.items()):
        out.append(f'"{kernel}",{counter},{sum(d.values()) / len(d):.1f},{len(d)}')
    return "\n".join(out)


if __name__ == "__main__":
    print(stats(sys.argv[2]) if sys.argv[1] == "stats" else pmc(sys.argv[2:]))
