#!/bin/bash
# Round 5, visit K: knob 46 -- the few-row projection's waves touch the lines of their later k steps behind their first loads.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
bench() { timeout 120 python bench.py --steps 12 --no-extras --no-roofline --no-cpu-baseline --latency-runs 0 "$@" 2>>gpurun_out/r5k.err | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print(d['value'])
except Exception as e: print('no result', e)"; }
{
  timeout 300 python -m pytest tests/ -q -m gpu -k "test_gemm_x3r or test_gemm_ln_x3r or x3r_route" 2>&1 | tail -2
  for k in 0 1 0 1; do echo "== decode probe, knob 46 = $k"; timeout 60 python tools/decode_probe.py --steps 16 --reps 3 --knob 46=$k 2>&1 | grep "decode probe"; done
  echo "== decode probe report, knob 46 = 1"; timeout 60 python tools/decode_probe.py --steps 16 --reps 2 --knob 46=1 --report 2>&1 | grep -A4 "decode probe"
  echo "== decode probe report, knob 46 = 0"; timeout 60 python tools/decode_probe.py --steps 16 --reps 2 --knob 46=0 --report 2>&1 | grep -A4 "decode probe"
  for rep in 1 2 3; do for k in 0 1; do echo "== bench, knob 46 = $k"; bench --knob 46=$k; done; done
} 2>&1 | tee gpurun_out/r5_k.log
