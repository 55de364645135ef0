"""Concurrency picture of ONE replayed step from a rocprofv3 --kernel-trace csv: which kernels ran beside which
(the side-stream weight gradients of ops.WgradSide), wall time against the sum of kernel durations, and the kernels of the
step in start order with what overlapped them.

    python profiles/overlap_timeline.py <kernel_trace.csv> [--list]
"""
import collections
import csv
import sys

from pmc_summary import norm_name


def main():
    path = sys.argv[1]
    rows = []
    with open(path, newline="") as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), norm_name(r["Kernel_Name"]), r.get("Queue_Id", "?")))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if r[2].startswith("adamw_k")]
    assert len(ends) >= 3, "need at least three optimizer steps in the trace"
    lo, hi = ends[-3] + 1, ends[-2] + 1          # the last-but-one step (the last ones may be eager timer steps)
    # prefer the last REPLAYED step: replayed steps are the shortest ones
    best = None
    for a, b in zip(ends[:-1], ends[1:]):
        span = rows[b][1] - rows[a][1]
        if best is None or span < best[0]:
            best = (span, a + 1, b + 1)
    span, lo, hi = best
    step = rows[lo:hi]
    t0 = min(r[0] for r in step)
    wall = (max(r[1] for r in step) - t0) * 1e-3
    tot = sum(r[1] - r[0] for r in step) * 1e-3
    # union of busy intervals and time with >= 2 kernels in flight
    evs = []
    for s, e, _n, _q in step:
        evs.append((s, 1)); evs.append((e, -1))
    evs.sort()
    depth, last, busy, multi = 0, evs[0][0], 0, 0
    for t, d in evs:
        if depth >= 1: busy += t - last
        if depth >= 2: multi += t - last
        depth += d; last = t
    print(f"step: {len(step)} kernels, wall {wall / 1e3:.3f} ms, sum of durations {tot / 1e3:.3f} ms, device busy {busy * 1e-6:.3f} ms, "
          f">= 2 kernels in flight {multi * 1e-6:.3f} ms")
    byq = collections.defaultdict(lambda: [0, 0.0])
    for s, e, n, q in step:
        byq[q][0] += 1; byq[q][1] += (e - s) * 1e-6
    for q, (n, ms) in sorted(byq.items(), key=lambda kv: -kv[1][1]):
        first = min(s for s, _e, _n, qq in step if qq == q)
        lastq = max(e for _s, e, _n, qq in step if qq == q)
        print(f"  queue {q}: {n} kernels, {ms:.3f} ms, first start {(first - t0) * 1e-3:.1f} us, last end {(lastq - t0) * 1e-3:.1f} us")
    # per kernel name: duration alone vs overlapped
    def overlapped(i):
        s, e = step[i][0], step[i][1]
        o = 0
        for j in range(max(0, i - 12), min(len(step), i + 12)):
            if j != i:
                o += max(0, min(e, step[j][1]) - max(s, step[j][0]))
        return o
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    ov = []
    for i, (s, e, n, q) in enumerate(step):
        o = overlapped(i)
        ov.append(o)
        a = agg[n]; a[0] += 1; a[1] += (e - s) * 1e-3; a[2] += min(o, e - s) * 1e-3
    print(f"{'n':>4s} {'total us':>10s} {'overlapped':>10s}  kernel")
    for n, (c, us, ous) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"{c:4d} {us:10.1f} {ous:10.1f}  {n[:100]}")
    if "--list" in sys.argv:
        for i, (s, e, n, q) in enumerate(step):
            print(f"{(s - t0) * 1e-3:10.1f} {(e - s) * 1e-3:8.1f} ov {min(ov[i], e - s) * 1e-3:8.1f} q{q} {n[:90]}")


if __name__ == "__main__":
    main()
