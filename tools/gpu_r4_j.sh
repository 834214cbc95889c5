#!/bin/bash
# Round 4, visit J: is the eight-worker bench bound by the host's launch rate?  hipGraph replay of the decoding steps
# (--graph-mode 1) against plain launches, at the defaults settled by visit I (x3r with fp32 A)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
short() { tee -a gpurun_out/r4_j_bench.jsonl | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d.get('parity_check', {}).get('ids_equal'), d.get('p50_latency_ms'), d.get('p50_latency_ms_by_mode'), {k: round(v, 1) for k, v in list(d.get('kernel_breakdown_ms', {}).items())[:8]})"; }
{
  echo "== tests"; timeout 900 python -m pytest tests/test_kernels.py tests/test_model_parity.py -q -m gpu -p no:cacheprovider -x -k "layernorm_x3p or x3r or device_side_step or panel_route" 2>&1 | tail -5
  B="python bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-extras --no-roofline"
  echo "== bench plain launches"; timeout 400 $B --latency-runs 3 2>/dev/null | short
  echo "== bench graph-mode 1"; timeout 400 $B --latency-runs 0 --graph-mode 1 2>/dev/null | short
  echo "== bench graph-mode 1, 12 workers x 3"; timeout 400 $B --latency-runs 0 --graph-mode 1 --streams 12 --group 3 2>/dev/null | short
  echo "== bench graph-mode 1, 6 x 6"; timeout 400 $B --latency-runs 0 --graph-mode 1 --streams 6 --group 6 2>/dev/null | short
  echo "== bench plain, 12 x 3"; timeout 400 $B --latency-runs 0 --streams 12 --group 3 2>/dev/null | short
  echo "== bench plain launches again"; timeout 400 $B --latency-runs 0 2>/dev/null | short
} 2>&1 | tee gpurun_out/r4_j.log
