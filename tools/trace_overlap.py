"""Concurrency summary of a rocprofv3 --kernel-trace CSV (several batches in flight on separate HIP streams):
how many kernels run at the same time, per-queue busy time, per-kernel total time.  Runs on the GPU box right after
the traced command; prints a small JSON so that only the summary has to travel back."""
import csv
import json
import sys
from collections import defaultdict


def main(path, t_lo_frac=0.0, t_hi_frac=1.0):
    rows = []
    with open(path, newline="") as f:
        rd = csv.DictReader(f)
        for r in rd:
            try:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]))
            except (KeyError, ValueError):
                continue
    if not rows:
        print(json.dumps({"error": "no kernel rows", "path": path}))
        return
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "stream_copy_kernel" in r[3]]
    if len(marks) >= 2:  # bench.py SBK_TRACE_MARK=1: exactly the timed region (a mark is a few launches of the marker kernel)
        lo_i = max(range(len(marks) - 1), key=lambda i: marks[i + 1] - marks[i])
        rows = rows[marks[lo_i] + 1: marks[lo_i + 1]]
    elif t_lo_frac > 0.0 or t_hi_frac < 1.0:  # the kernels between two percentiles of the launch order: the steady state of a timed region
        rows = rows[int(len(rows) * t_lo_frac): max(int(len(rows) * t_hi_frac), int(len(rows) * t_lo_frac) + 1)]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    # sweep: time-weighted histogram of the number of kernels in flight
    ev = []
    for s, e, _, _ in rows:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    hist = defaultdict(int)
    cur, last = 0, ev[0][0]
    for t, d in ev:
        hist[cur] += t - last
        last = t
        cur += d
    span = t1 - t0
    busy = span - hist.get(0, 0)
    kern_ns = sum(e - s for s, e, _, _ in rows)
    per_q = defaultdict(int)
    per_k = defaultdict(lambda: [0, 0])
    for s, e, q, k in rows:
        per_q[q] += e - s
        short = k.replace("void ", "").replace("(anonymous namespace)::", "").split("<")[0].split("(")[0][-48:]
        per_k[short][0] += e - s
        per_k[short][1] += 1
    out = {
        "kernels": len(rows), "span_ms": round(span / 1e6, 2), "gpu_idle_frac": round(hist.get(0, 0) / span, 4),
        "sum_kernel_ms": round(kern_ns / 1e6, 2), "mean_kernels_in_flight_when_busy": round(kern_ns / max(busy, 1), 3),
        "time_frac_by_kernels_in_flight": {str(k): round(v / span, 4) for k, v in sorted(hist.items()) if v / span > 0.002},
        "queues": {q: round(v / 1e6, 1) for q, v in sorted(per_q.items(), key=lambda kv: -kv[1])},
        "top_kernels_ms": {k: [round(v[0] / 1e6, 2), v[1], round(v[0] / v[1] / 1e3, 1)] for k, v in
                           sorted(per_k.items(), key=lambda kv: -kv[1][0])[:24]},
    }
    print(json.dumps(out))


if __name__ == "__main__":
    main(sys.argv[1], *(float(v) for v in sys.argv[2:4]))
