#!/usr/bin/env python
"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into the per-kernel HBM traffic table bench.py reads.

    python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.csv> [command text]

Units / correction follow /opt/skills/guides/MI355X_MICROARCH.md (HBM, rocprofv3): both counters are KB per dispatch;
on gfx950 FETCH_SIZE under-reports a wide coalesced stream by 2x, so fetch_MB_corrected = 2 x raw.
"""
import csv
import sys
from collections import OrderedDict


def per_kernel(path, counter):
    acc = OrderedDict()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        s = acc.setdefault(r["Kernel_Name"], [0, 0.0])
        s[0] += 1
        s[1] += float(r["Counter_Value"])
    return acc


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    cmd = " ".join(sys.argv[4:]) or "python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    lines = [f"# rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separate pass, --pmc WRITE_SIZE) -- {cmd}",
             f"# csrc_sha16: {bench.csrc_sha16()}",
             "# raw counters are KB per dispatch; on gfx950 FETCH_SIZE reads 1/2 of a wide coalesced stream "
             "(MI355X_MICROARCH.md, HBM) -> fetch_MB_corrected = 2 x raw",
             "kernel,dispatches,FETCH_SIZE_KB_avg,fetch_MB_corrected,WRITE_SIZE_KB_avg,write_MB"]
    order = sorted(fetch, key=lambda n: -fetch[n][1])
    for name in order:
        n, tot = fetch[name]
        f_kb = tot / n
        wn, wtot = write.get(name, (1, 0.0))
        w_kb = wtot / max(wn, 1)
        lines.append('"%s",%d,%.1f,%.1f,%.1f,%.1f' % (name, n, f_kb, 2 * f_kb * 1024 / 1e6, w_kb, w_kb * 1024 / 1e6))
    open(sys.argv[3], "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
