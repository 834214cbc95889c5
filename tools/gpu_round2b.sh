#!/bin/bash
# Round-2, second half: new parity tests, A/B of the grouped encoder pass, Whisper encoder leg.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  echo "== pytest -m gpu (${TEST_K:-all})"
  if [ -n "${TEST_K:-}" ]; then
    timeout ${TEST_TIMEOUT:-600} python -m pytest tests -q -m gpu -x -k "$TEST_K" 2>&1 | tail -8
  else
    timeout ${TEST_TIMEOUT:-1200} python -m pytest tests -q -m gpu -x 2>&1 | tail -8
  fi
  echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
  i=0
  while IFS= read -r line; do
    [ -z "$line" ] && continue
    i=$((i+1))
    echo "== bench case $i: $line"
    timeout ${CASE_TIMEOUT:-420} python bench.py $line > gpurun_out/b2_case_$i.json 2> gpurun_out/b2_case_$i.err
    tail -2 gpurun_out/b2_case_$i.err; python - "$i" <<'PY'
import json, sys
try:
    d = json.loads(open(f"gpurun_out/b2_case_{sys.argv[1]}.json").read().strip().splitlines()[-1])
    keep = {k: d.get(k) for k in ("value", "value_batch128", "value_encoder_gemms_bf16", "p50_latency_ms", "ms_per_step")}
    keep["roofline"] = {k: (d.get("roofline") or {}).get(k) for k in ("kernel", "achieved", "frac", "avg_launch_ms")}
    keep["top3"] = [(e["kernel"], e["achieved"], e["frac"]) for e in d.get("roofline_top3", [])]
    keep["e2e"] = d.get("roofline_end_to_end")
    keep["whisper"] = d.get("config5_whisper_encoder")
    print(json.dumps(keep))
except Exception as e:
    print("no json:", e)
PY
  done < tools/b2_cases.txt
} 2>&1 | tee gpurun_out/round2b.log
