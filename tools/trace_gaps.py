"""Who ends the GPU's idle gaps?  From a rocprofv3 --kernel-trace CSV of the eight-worker timed region: every interval
with NO kernel in flight is charged to the kernel that starts next (and, separately, to the one that ended last), and the
idle time is split by gap length.  Tells launch-latency gaps (short, between dependent kernels of one search) from
host-side starvation (long, before an encoder kernel issued from Python)."""
import csv
import json
import sys
from collections import defaultdict


def short(k):
    return k.replace("void ", "").replace("(anonymous namespace)::", "").split("<")[0].split("(")[0][-40:]


def main(path, lo=0.3, hi=0.9):
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            try:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")))
            except (KeyError, ValueError):
                continue
    rows.sort()
    marks = [i for i, r in enumerate(rows) if "stream_copy_kernel" in r[2]]
    if len(marks) >= 2:  # bench.py SBK_TRACE_MARK=1: exactly the timed region (a mark is a few launches of the marker kernel)
        lo_i = max(range(len(marks) - 1), key=lambda i: marks[i + 1] - marks[i])
        rows = rows[marks[lo_i] + 1: marks[lo_i + 1]]
    else:
        rows = rows[int(len(rows) * lo): int(len(rows) * hi)]
    span = max(r[1] for r in rows) - rows[0][0]
    by_next, by_prev, by_len = defaultdict(float), defaultdict(float), defaultdict(float)
    n_gaps, idle = 0, 0
    cur_end, prev_name = rows[0][1], rows[0][2]
    for s, e, name, q in rows[1:]:
        if s > cur_end:
            g = s - cur_end
            idle += g
            n_gaps += 1
            by_next[name] += g
            by_prev[prev_name] += g
            by_len["<2us" if g < 2000 else "2-5us" if g < 5000 else "5-20us" if g < 20000 else "20-100us" if g < 100000 else ">100us"] += g
        if e > cur_end:
            cur_end, prev_name = e, name
    top = lambda d: {k: round(v / 1e6, 2) for k, v in sorted(d.items(), key=lambda kv: -kv[1])[:10]}
    print(json.dumps({"span_ms": round(span / 1e6, 1), "idle_ms": round(idle / 1e6, 1), "idle_frac": round(idle / span, 4),
                      "gaps": n_gaps, "idle_ms_by_gap_length": {k: round(v / 1e6, 2) for k, v in by_len.items()},
                      "cut": "markers" if len(marks) >= 2 else "percentiles", "idle_ms_by_next_kernel": top(by_next), "idle_ms_by_previous_kernel": top(by_prev)}))


if __name__ == "__main__":
    main(sys.argv[1], *(float(v) for v in sys.argv[2:4]))
