#!/usr/bin/env python
"""Per-kernel averages of every counter in one or more rocprofv3 `--pmc ... --output-format csv` result files, with the
derived quantities the roofline discussion needs:
    effective clock      = GRBM_GUI_ACTIVE / 8 / kernel duration      (MI355X_MICROARCH.md, "DVFS give-back"; rocprofv3 reports the
                           counter summed over the 8 XCDs' GRBMs: the raw quotient would be 15 GHz)
    MFMA utilisation     = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)
                           (SQ_VALU_MFMA_BUSY_CYCLES = 32 cycles x number of 32x32x16 MFMA wave-instructions, summed over the chip)

    python tools/pmc_counters.py out.csv kernel_substring file1_counter_collection.csv [file2 ...]
"""
import csv
import sys
from collections import OrderedDict, defaultdict


def main():
    out, key = sys.argv[1], sys.argv[2]
    acc = defaultdict(lambda: [0, 0.0])          # counter -> [dispatches, sum]
    dur = [0, 0.0]
    name = None
    for path in sys.argv[3:]:
        seen = set()
        for r in csv.DictReader(open(path)):
            if key not in r["Kernel_Name"]:
                continue
            name = r["Kernel_Name"]
            a = acc[r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                dur[0] += 1
                dur[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    avg = OrderedDict((c, v[1] / v[0]) for c, v in sorted(acc.items()))
    avg_ns = dur[1] / max(dur[0], 1)
    lines = [f"# rocprofv3 --kernel-trace --pmc <counters> (separate passes), kernel '{name}', averages per dispatch over {dur[0]} dispatches",
             "counter,average_per_dispatch"]
    lines += [f"{c},{v:.6g}" for c, v in avg.items()]
    lines.append(f"kernel_duration_ns_under_pmc,{avg_ns:.6g}")
    if "GRBM_GUI_ACTIVE" in avg:
        cyc = avg["GRBM_GUI_ACTIVE"] / 8.0
        lines.append(f"elapsed_cycles_per_xcd,{cyc:.6g}")
        lines.append(f"effective_clock_GHz,{cyc / avg_ns:.4f}")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
            lines.append(f"mfma_pipe_busy_fraction,{avg['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * cyc):.4f}")
    if "TCC_HIT_sum" in avg and "TCC_MISS_sum" in avg:
        lines.append(f"l2_hit_rate,{avg['TCC_HIT_sum'] / (avg['TCC_HIT_sum'] + avg['TCC_MISS_sum']):.4f}")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
