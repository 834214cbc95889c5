#!/bin/bash
# Round 4, visit I: x3r v2 (panel A operands, 16-byte epilogue, panel hand-over of the feed-forward hidden layer), the CTC
# prefix score on the matrix cores (knob 7 = 8): parity, microbench, decode-step traces, bench A/B
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
short() { tee -a gpurun_out/r4_i_bench.jsonl | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d.get('parity_check', {}).get('ids_equal'), {k: round(v, 1) for k, v in list(d.get('kernel_breakdown_ms', {}).items())[:14]}); print(json.dumps(d.get('roofline_top3'))[:700]); print(d['config'].get('gpu_memory_reserved_gb'), d.get('p50_latency_ms'))"; }
trace() { (cd /tmp && rm -rf /tmp/dtr && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/dtr -o t -- python $OLDPWD/tools/decode_probe.py --steps 16 --reps 3 "$@" 2>&1 | grep "decode probe"); f=$(find /tmp/dtr -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/decode_trace.py "$f" 48 | head -${TRACE_LINES:-14}; }
{
  echo "== tests"; timeout 900 python -m pytest tests/test_kernels.py tests/test_model_parity.py -q -m gpu -p no:cacheprovider -x -k "gemm_x3r or x3r_route or fused_scoring or golden_model or grouped_search or ctc_score_tokens or wide_beam or long_utterance" 2>&1 | tail -12
  echo "== microbench"; timeout 300 python tools/microbench.py --x3r 2>&1 | grep -v amdgpu.ids | grep -E "M=1280|M=320 N=512 K=512|M=2560 N=2048"
  echo "== decode trace, default (x3r with panel A)"; TRACE_LINES=20 trace
  echo "== decode trace, x3r with fp32 A"; trace --knob 44=0
  echo "== decode trace, CTC score on the matrix cores"; trace --knob 7=8
  B="python bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-extras"
  echo "== bench default"; timeout 400 $B 2>/dev/null | short
  echo "== bench knob 44=0 (fp32 A)"; timeout 400 $B --no-roofline --latency-runs 0 --knob 44=0 2>/dev/null | short
  echo "== bench knob 7=8 (CTC mfma)"; timeout 400 $B --no-roofline --latency-runs 0 --knob 7=8 2>/dev/null | short
} 2>&1 | tee gpurun_out/r4_i.log
