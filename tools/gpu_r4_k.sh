#!/bin/bash
# Round 4, visit K: x3r v3 (rolled step loop, K never split across workgroups: the 2 048-deep feed-forward projection in one
# launch): parity, microbench, decode-step trace, two 16-step bench runs (run-to-run spread)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
short() { tee -a gpurun_out/r4_k_bench.jsonl | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d.get('parity_check', {}).get('ids_equal'), d.get('p50_latency_ms'), {k: round(v, 1) for k, v in list(d.get('kernel_breakdown_ms', {}).items())[:10]}); print(json.dumps(d.get('roofline'))[:400])"; }
trace() { (cd /tmp && rm -rf /tmp/dtr && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/dtr -o t -- python $OLDPWD/tools/decode_probe.py --steps 16 --reps 3 "$@" 2>&1 | grep "decode probe"); f=$(find /tmp/dtr -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/decode_trace.py "$f" 48 | head -${TRACE_LINES:-14}; }
{
  echo "== tests"; timeout 900 python -m pytest tests/test_kernels.py tests/test_model_parity.py -q -m gpu -p no:cacheprovider -x -k "gemm_x3r or x3r_route or golden_model or grouped_search" 2>&1 | tail -5
  echo "== microbench"; timeout 300 python tools/microbench.py --x3r 2>&1 | grep -v amdgpu.ids | grep -E "M=1280|M=320 N=512 K=2048|M=2560 N=512 K=2048"
  echo "== decode trace"; TRACE_LINES=20 trace
  B="python bench.py --warmup 1 --no-cpu-baseline --no-extras"
  echo "== bench 16 steps"; timeout 400 $B --latency-runs 3 2>/dev/null | short
  echo "== bench 16 steps again"; timeout 400 $B --latency-runs 0 --no-roofline 2>/dev/null | short
} 2>&1 | tee gpurun_out/r4_k.log
