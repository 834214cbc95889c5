"""The launches of ONE decoding step in issue order, from a rocprofv3 --kernel-trace CSV of tools/decode_probe.py: per position in the
step the kernel, its mean duration over the traced steps and the mean gap in front of it (device timestamps; no HIP events in the stream).

    python tools/decode_timeline.py <kernel_trace.csv> [steps] [anchor kernel: embed_pos (default) | decoder_step_persist ...]
"""
import csv
import sys
from collections import Counter


def short(k):
    k = k.replace("void ", "").replace("(anonymous namespace)::", "")
    head = k.split("(")[0]
    return head[-46:]


def main(path, steps=24, anchor="embed_pos"):
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            try:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]),
                             r.get("Grid_Size", r.get("Grid_Size_X", "?")), r.get("LDS_Block_Size", "?")))
            except (KeyError, ValueError):
                continue
    rows.sort()
    idx = [i for i, r in enumerate(rows) if anchor in r[2]]
    idx = idx[-(steps + 1):]
    seqs = [rows[a:b] for a, b in zip(idx[:-1], idx[1:])]
    n = Counter(len(s) for s in seqs).most_common(1)[0][0]
    seqs = [s for s in seqs if len(s) == n]
    prev_end = [rows[i - 1][1] for i, s in zip(idx[:-1], [rows[a:b] for a, b in zip(idx[:-1], idx[1:])]) if len(s) == n]
    print(f"{len(seqs)} steps of {n} launches; step = {sum(s[-1][1] - s[0][0] for s in seqs) / len(seqs) / 1e3:.1f} us "
          f"(kernels {sum(sum(e - b for b, e, *_ in s) for s in seqs) / len(seqs) / 1e3:.1f} us)")
    for k in range(n):
        name = seqs[0][k][2]
        dur = sum(s[k][1] - s[k][0] for s in seqs) / len(seqs) / 1e3
        gap = sum((s[k][0] - (s[k - 1][1] if k else pe)) for s, pe in zip(seqs, prev_end)) / len(seqs) / 1e3
        print(f"  {k:3d} {name:46s} grid {seqs[0][k][3]:>8s} {dur:8.1f} us   gap in front {gap:6.1f} us")


if __name__ == "__main__":
    main(sys.argv[1], *(int(v) for v in sys.argv[2:3]), *sys.argv[3:4])
