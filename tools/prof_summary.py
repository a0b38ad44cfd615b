#!/usr/bin/env python
"""Summarise a rocprofv3 results .db (kernel-trace) into a CSV like --stats would print."""
import sqlite3
import sys


def summarise(db_path, out_path=None, header=""):
    cur = sqlite3.connect(db_path).cursor()
    rows = list(cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    lines = [f"# {header}", "name,calls,total_us,avg_us,pct"]
    lines += ['"%s",%d,%.3f,%.3f,%.2f' % r for r in rows]
    txt = "\n".join(lines) + "\n"
    if out_path:
        open(out_path, "w").write(txt)
    return txt


if __name__ == "__main__":
    print(summarise(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None, " ".join(sys.argv[3:])))
