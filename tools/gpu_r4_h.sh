#!/bin/bash
# Round 4, visit H: the decode step's projections on the bf16 matrix pipe (sbk_gemm_nt_x3r): parity, microbench at the
# decode shapes, kernel trace of a decoding step, bench A/B
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
short() { tee -a gpurun_out/r4_h_bench.jsonl | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d.get('parity_check', {}).get('ids_equal'), {k: round(v, 1) for k, v in list(d.get('kernel_breakdown_ms', {}).items())[:14]}); print(json.dumps(d.get('roofline_top3'))[:900]); print(d['config'].get('gpu_memory_reserved_gb'), d.get('p50_latency_ms'))"; }
trace() { (cd /tmp && rm -rf /tmp/dtr && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/dtr -o t -- python $OLDPWD/tools/decode_probe.py --steps 16 --reps 3 "$@" 2>&1 | grep "decode probe"); f=$(find /tmp/dtr -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/decode_trace.py "$f" 48 | head -${TRACE_LINES:-16}; }
{
  echo "== tests"; timeout 1200 python -m pytest tests/test_kernels.py tests/test_model_parity.py tests/test_full_size_gpu.py -q -m gpu -p no:cacheprovider -x -k "gemm_x3r or x3r_route or fused_scoring or golden_model or grouped_search or headline or decoder_logprobs or lm_scorer or greedy or properties or wide_beam" 2>&1 | tail -12
  echo "== microbench"; timeout 300 python tools/microbench.py --x3r 2>&1 | grep -v amdgpu.ids
  echo "== decode trace, x3r mode 2 (default)"; TRACE_LINES=24 trace
  echo "== decode trace, x3r mode 1"; trace --knob 41=1
  echo "== decode trace, x3r off"; trace --knob 41=0
  echo "== decode trace, x3r mode 2, vocabulary projection on the 128-wide kernel"; trace --knob 43=0
  B="python bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-extras"
  echo "== bench default"; timeout 400 $B 2>/dev/null | short
  echo "== bench SBK_X3R=0"; SBK_X3R=0 timeout 400 $B --no-roofline --latency-runs 0 2>/dev/null | short
  echo "== bench x3r mode 1"; timeout 400 $B --no-roofline --latency-runs 0 --knob 41=1 2>/dev/null | short
} 2>&1 | tee gpurun_out/r4_h.log
