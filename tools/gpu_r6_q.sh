#!/bin/bash
# Round 6, visit Q: the single-utterance encoder (3.05 ms for ~200 launches issued from Python): its device-side kernel sum (timeline) and
# the feasibility of replaying it from a captured graph (tools/latency_probe.py).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
{
  timeout 200 python tools/latency_probe.py --runs 9 --overlap 3 2>&1 | grep -E "latency probe|graphed"
  timeout 200 python tools/latency_probe.py --runs 9 --overlap 3 --seconds 20 2>&1 | grep -E "latency probe|graphed"
  (cd /tmp && rm -rf /tmp/tq && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tq -o t -- python $R/tools/latency_probe.py --runs 3 --overlap 3 2>&1 | grep "latency probe")
  f=$(find /tmp/tq -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))), key=lambda r: r[0])
# the encoder of the LAST transcribe/encode call: from the last fbank kernel to the first decoder kernel after it
idx = [i for i, r in enumerate(rows) if "fbank_frames" in r[2]]
i0 = idx[-1]
i1 = next(i for i in range(i0, len(rows)) if "decoder_step_persist" in rows[i][2] or "embed_pos" in rows[i][2] or i == len(rows) - 1)
seg = rows[i0:i1]
busy = sum(e - s for s, e, _ in seg)
print(f"  encoder of one 10-s utterance: {len(seg)} launches, span {(seg[-1][1] - seg[0][0]) / 1e3:.1f} us, kernels {busy / 1e3:.1f} us, mean {busy / len(seg) / 1e3:.1f} us each")
PY
} 2>&1 | tee gpurun_out/r6_q.log
