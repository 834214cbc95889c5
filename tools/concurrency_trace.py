#!/usr/bin/env python
"""How many kernels are resident while the workers run concurrently -- from a rocprofv3 --kernel-trace CSV (plain or .gz).

    python tools/concurrency_trace.py TRACE.csv[.gz] [--from S] [--to S]

Prints, for the window [from, to) in seconds after the first dispatch: the share of the wall clock with 0 / 1 / 2 / ...
kernels in flight, the average number in flight, and per kernel name: launches, mean duration, the time it ran ALONE
(nothing else in flight: only making THAT kernel shorter, or overlapping it, recovers this time), and the time-integral
of 1 / (kernels in flight) -- its share of the wall clock if co-resident kernels split the machine evenly.
Measurement tooling only (tools/gpu_r4_q.sh collects the trace of the default bench configuration).
"""
import argparse
import collections
import csv
import gzip
import re


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:44]


def load(path):
    op = gzip.open if path.endswith(".gz") else open
    rows = []
    with op(path, "rt") as f:
        for x in csv.DictReader(f):
            rows.append((int(x["Start_Timestamp"]), int(x["End_Timestamp"]), short(x["Kernel_Name"]), int(x["Thread_Id"]),
                         int(x["Grid_Size_X"]) * int(x["Grid_Size_Y"]) * int(x["Grid_Size_Z"]),
                         int(x["Workgroup_Size_X"]) * int(x["Workgroup_Size_Y"]) * int(x["Workgroup_Size_Z"])))
    return rows


def analyse(rows, lo, hi):
    t0 = min(r[0] for r in rows)
    lo_ns, hi_ns = t0 + int(lo * 1e9), t0 + int(hi * 1e9)
    sel = [r for r in rows if r[1] > lo_ns and r[0] < hi_ns]
    ev = []
    for i, r in enumerate(sel):
        ev.append((max(r[0], lo_ns), 1, i))
        ev.append((min(r[1], hi_ns), -1, i))
    ev.sort()
    level = collections.Counter()
    alone = collections.Counter()
    share = collections.Counter()
    live = set()
    prev = lo_ns
    for t, s, i in ev:
        dt = t - prev
        if dt > 0:
            n = len(live)
            level[n] += dt
            if n == 1:
                alone[sel[next(iter(live))][2]] += dt
            if n:
                w = dt / n
                for j in live:
                    share[sel[j][2]] += w
        prev = t
        if s > 0:
            live.add(i)
        else:
            live.discard(i)
    if hi_ns > prev:
        level[0] += hi_ns - prev
    return sel, level, alone, share, hi_ns - lo_ns


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--from", dest="lo", type=float, default=0.0)
    ap.add_argument("--to", dest="hi", type=float, default=1e9)
    a = ap.parse_args()
    rows = load(a.trace)
    span = (max(r[1] for r in rows) - min(r[0] for r in rows)) / 1e9
    hi = min(a.hi, span)
    sel, level, alone, share, wall = analyse(rows, a.lo, hi)
    print(f"window {a.lo:.2f} .. {hi:.2f} s: {len(sel)} dispatches from {len(set(r[3] for r in sel))} host threads, "
          f"wall {wall / 1e6:.1f} ms, kernel time {sum(min(r[1], 0) or (r[1] - r[0]) for r in sel) / 1e6:.1f} ms")
    tot = sum(level.values())
    avg = sum(k * v for k, v in level.items()) / tot
    print("kernels in flight -> share of the wall clock: " +
          "  ".join(f"{k}: {100 * level[k] / tot:.1f}%" for k in sorted(level) if level[k] / tot >= 0.002) + f"   mean {avg:.2f}")
    cnt = collections.Counter(r[2] for r in sel)
    dur = collections.Counter()
    wgs = collections.Counter()
    for r in sel:
        dur[r[2]] += r[1] - r[0]
        wgs[r[2]] += r[4] // max(r[5], 1)
    print(f"{'kernel':46s} {'launches':>8s} {'mean us':>8s} {'WGs':>6s} {'sum ms':>8s} {'alone ms':>9s} {'1/n share':>9s}")
    for k, v in share.most_common(24):
        print(f"{k:46s} {cnt[k]:8d} {dur[k] / cnt[k] / 1e3:8.1f} {wgs[k] // cnt[k]:6d} {dur[k] / 1e6:8.1f} {alone[k] / 1e6:9.1f} "
              f"{100 * v / wall:8.1f}%")


if __name__ == "__main__":
    main()
