"""Per-kernel summary of a rocprofv3 --kernel-trace csv: launches, average duration, ms per step.

    python profiles/trace_summary.py <kernel_trace.csv> <steps traced>   (steps = warm-up + eager + replayed steps of the run)
"""
import collections
import csv
import sys

from pmc_summary import norm_name


def main():
    path, steps = sys.argv[1], float(sys.argv[2])
    acc = collections.defaultdict(list)
    with open(path, newline="") as fh:
        for r in csv.DictReader(fh):
            acc[norm_name(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
    tot = sum(sum(v) for v in acc.values())
    print(f"total kernel time {tot / 1e3:.2f} ms over {steps:g} steps = {tot / 1e3 / steps:.2f} ms per step; {sum(len(v) for v in acc.values()) / steps:.0f} launches per step")
    print(f"{'ms/step':>9s} {'n/step':>7s} {'avg us':>9s}  kernel")
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        print(f"{sum(v) / 1e3 / steps:9.3f} {len(v) / steps:7.1f} {sum(v) / len(v):9.1f}  {k[:110]}")


if __name__ == "__main__":
    main()
