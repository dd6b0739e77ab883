#!/usr/bin/env python3
"""Summarise rocprofv3 (rocpd sqlite) outputs for this repo's kernels.

  tools/rocprof_summary.py stats  <results.db>            -> per-kernel calls / total / average (like --stats)
  tools/rocprof_summary.py pmc    <results.db> <COUNTER>  -> per-kernel sum and per-launch mean of a PMC counter (KB)
"""
import sqlite3
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = name.split("(")[0].strip()
    if name.startswith("void "):            # template instantiations are reported with their return type
        name = name[5:]
    return name


def main():
    mode, path = sys.argv[1], sys.argv[2]
    db = sqlite3.connect(path)
    if mode == "stats":
        print("kernel,calls,total_ms,average_ms,percent")
        for name, calls, total, avg, pct in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            if short(name).startswith("k_"):
                print(f"{short(name)},{calls},{total / 1e3:.3f},{avg / 1e3:.3f},{pct:.3f}")
    else:
        counter = sys.argv[3]
        print("kernel,launches,%s_sum_KB,%s_per_launch_KB,avg_duration_ms" % (counter, counter))
        q = ("select kernel_name, count(*), sum(value), avg(value), avg(duration) from counters_collection "
             "where counter_name=? group by kernel_name")
        for name, n, s, a, d in db.execute(q, (counter,)):
            if short(name).startswith("k_"):
                print(f"{short(name)},{n},{s:.1f},{a:.1f},{d / 1e6:.3f}")


if __name__ == "__main__":
    main()
