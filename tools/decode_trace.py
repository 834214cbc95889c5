"""Per-kernel TRUE durations and the gaps between dependent launches of a single-stream grouped search, from a
rocprofv3 --kernel-trace CSV of tools/decode_probe.py (no HIP events in the stream): where a decoding step's time goes.

    python tools/decode_trace.py <kernel_trace.csv> [steps]
"""
import csv
import sys
from collections import defaultdict


def short(k):
    return k.replace("void ", "").replace("(anonymous namespace)::", "").split("<")[0].split("(")[0][-44:]


def main(path, steps=48):
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            try:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
            except (KeyError, ValueError):
                continue
    rows.sort()
    # the decoding steps: from the first to the last embed_pos launch of the LAST search (the timed repetitions)
    idx = [i for i, r in enumerate(rows) if "embed_pos" in r[2]]
    idx = idx[-steps:] if len(idx) > steps else idx
    rows = rows[idx[0]: idx[-1]]
    n_steps = len(idx) - 1
    dur, cnt, gap_after = defaultdict(float), defaultdict(int), defaultdict(float)
    busy, gaps, end = 0.0, 0.0, rows[0][0]
    for s, e, name in rows:
        dur[name] += e - s
        cnt[name] += 1
        if s > end:
            gaps += s - end
            gap_after[name] += s - end
        busy += max(0, e - max(s, end))
        end = max(end, e)
    span = end - rows[0][0]
    print(f"decode trace: {n_steps} steps, {span / n_steps / 1e3:.1f} us per step: kernels {busy / n_steps / 1e3:.1f} us, "
          f"gaps {gaps / n_steps / 1e3:.1f} us, {len(rows) / n_steps:.1f} launches per step")
    for k, v in sorted(dur.items(), key=lambda kv: -kv[1]):
        print(f"  {k:44s} {cnt[k] / n_steps:5.1f} per step {v / cnt[k] / 1e3:8.1f} us each {v / n_steps / 1e3:8.1f} us per step"
              f"   gap in front {gap_after[k] / max(cnt[k], 1) / 1e3:5.1f} us")


if __name__ == "__main__":
    main(sys.argv[1], *(int(v) for v in sys.argv[2:3]))
