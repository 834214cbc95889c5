#!/bin/bash
# Round 6, visit G: clocks and power during the 8-worker headline run (is the chip power-limited?): rocm-smi sampled every 0.25 s
# beside `bench.py --steps 16`; the same beside the one-stream decode probe.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
sample() { while true; do /opt/rocm/bin/rocm-smi --showclocks --showpower --showuse --showmemuse --json 2>/dev/null | tr -d '\n'; echo; sleep 0.25; done; }
summ() { python - "$1" <<'PY'
import json, sys, re
rows = []
for line in open(sys.argv[1]):
    try:
        d = json.loads(line)["card0"]
    except Exception:
        continue
    g = lambda pat: next((v for k, v in d.items() if re.search(pat, k)), None)
    rows.append((g(r"sclk clock speed"), g(r"mclk clock speed"), g(r"Power \(W\)|Graphics Package Power"), g(r"GPU use"), g(r"fclk clock speed")))
print(f"{len(rows)} samples")
def num(x):
    m = re.search(r"[-+]?\d+\.?\d*", str(x)); return float(m.group()) if m else None
for i, name in enumerate(("sclk MHz", "mclk MHz", "power W", "GPU use %", "fclk MHz")):
    v = [num(r[i]) for r in rows if num(r[i]) is not None]
    if v:
        v2 = sorted(v)
        print(f"  {name:10s} min {v2[0]:8.1f}  p25 {v2[len(v2)//4]:8.1f}  median {v2[len(v2)//2]:8.1f}  p75 {v2[3*len(v2)//4]:8.1f}  max {v2[-1]:8.1f}")
if rows: print("  first keys:", list(json.loads(open(sys.argv[1]).readline())["card0"].keys())[:12])
PY
}
{
  echo "== 8-worker bench"
  sample > gpurun_out/smi_bench.jsonl & SP=$!
  timeout 300 python bench.py --steps 16 --no-extras --no-roofline --no-cpu-baseline --latency-runs 0 2>/dev/null | tail -1 | cut -c1-160
  kill $SP; wait $SP 2>/dev/null
  summ gpurun_out/smi_bench.jsonl
  echo "== one-stream decode probe"
  sample > gpurun_out/smi_probe.jsonl & SP=$!
  timeout 200 python tools/decode_probe.py --steps 60 --reps 8 2>&1 | grep "decode probe"
  kill $SP; wait $SP 2>/dev/null
  summ gpurun_out/smi_probe.jsonl
} 2>&1 | tee gpurun_out/r6_g.log
