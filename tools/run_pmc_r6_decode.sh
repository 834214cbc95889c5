#!/bin/bash
# Fabric traffic of the memory-bound decode-step kernels (cross_attn_ring, ctc_score_step, self_attn_step) at the headline's
# decode shape (tools/decode_probe.py: 4 x 32 utterances, T' 430, beam 10 + CTC), FETCH_SIZE and WRITE_SIZE in separate --pmc
# passes beside --kernel-trace only (MI355X_MICROARCH.md).  Output: gpurun_out/pmc_r6_decode_{fetch,write}.csv + a summary.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
# (90 s per pass: a healthy pass takes 10-25 s; in visit R rocprofv3 died at start-up and sat in its signal handler until the
#  timeout -- two 300-second waits were the round's last GPU minutes, profiles/r05_r_*)
CMD="python $PWD/tools/decode_probe.py --steps 8 --reps 1"
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1)); tag=$(echo fetch write | cut -d' ' -f$i)
  (cd /tmp && rm -rf /tmp/pmcd6 && timeout 90 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d /tmp/pmcd6 -o x -- $CMD > $OLDPWD/gpurun_out/pmc_r6_decode_$tag.log 2>&1)
  f=$(find /tmp/pmcd6 -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && cp "$f" gpurun_out/pmc_r6_decode_$tag.csv
done
python - <<'PY'
import csv, collections
# algorithmic bytes per launch at this shape: cross K/V 8 len d per utterance (T' 430..400, 128 utterances, one layer);
# CTC posteriors 4 T' V per utterance; self-attention cache 8 d (step + 1) per hypothesis row (1 280 rows), mean over the steps run
T = [430, 420, 410, 400]
# (lengths 0.85 .. 1.0 of T' within a batch: mean 0.925)
alg = {"cross_attn_ring_kernel": int(0.925 * sum(32 * t for t in T)) * 8 * 512, "ctc_score_step_kernel": int(0.925 * sum(32 * t for t in T)) * 4 * 5000,
       "self_attn_step_kernel": None}
vals = {}
for tag in ("fetch", "write"):
    try:
        rows = list(csv.DictReader(open(f"gpurun_out/pmc_r6_decode_{tag}.csv")))
    except Exception as e:
        print(tag, "missing", e); continue
    for k in alg:
        rs = [r for r in rows if k in r["Kernel_Name"]]
        rs = rs[len(rs) // 2:]  # (the second half: the warm-up search comes first)
        if not rs:
            continue
        v = sum(float(r["Counter_Value"]) for r in rs) / len(rs)
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs if r.get("End_Timestamp")]
        vals.setdefault(k, {})[tag] = (v, (sum(d) / len(d) / 1e3) if d else float("nan"), len(rs))
for k, v in vals.items():
    f, w = v.get("fetch", (0, 0, 0)), v.get("write", (0, 0, 0))
    total = (2 * f[0] + w[0]) * 1024  # KiB units; the fabric-side counter tallies 128-byte requests at 64 (MI355X_MICROARCH.md)
    a = alg[k]
    print(f"{k:26s} launches={f[2]:4d} avg_us={f[1]:8.1f} FETCH_SIZE={f[0]:12.1f} WRITE_SIZE={w[0]:10.1f} bytes_per_launch={total/1e6:9.1f} MB"
          + (f" algorithmic={a/1e6:9.1f} MB ratio={total/a:5.2f} rate_under_counters={total/f[1]/1e3:7.1f} GB/s" if a else ""))
PY
