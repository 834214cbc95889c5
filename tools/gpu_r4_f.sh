#!/bin/bash
# Round 4, visit F: panel chain of the encoder (LayerNorm -> panel, feed-forward hidden layer as a panel), VALU arg-max
# rounds in the scoring kernels, explicit (non-contractable) scoring arithmetic: parity, microbench, bench A/B
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
short() { tee -a gpurun_out/r4_f_bench.jsonl | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d.get('parity_check', {}).get('ids_equal'), {k: round(v, 1) for k, v in list(d.get('kernel_breakdown_ms', {}).items())[:12]}); print(json.dumps(d.get('roofline')))"; }
{
  echo "== tests"; timeout 1200 python -m pytest tests/test_kernels.py tests/test_model_parity.py tests/test_full_size_gpu.py -q -m gpu -p no:cacheprovider -x -k "x3p or fused_scoring or panel_route or golden_model or grouped_search or device_side_step or wide_beam or headline or decoder_logprobs or lm_scorer or greedy or properties or transformerlm or encoder_vs_oracle or partial_ctc or attention_window" 2>&1 | tail -30
  echo "== microbench"; timeout 300 python tools/microbench.py --ln-x3p 2>&1 | grep -v amdgpu.ids
  echo "== decode probe, fused / separate (no events)"; timeout 300 python tools/decode_probe.py --steps 16 --reps 3 2>&1 | grep -v amdgpu.ids; timeout 300 python tools/decode_probe.py --steps 16 --reps 3 --knob 40=0 2>&1 | grep -v amdgpu.ids
  echo "== decode probe, fused, events"; timeout 300 python tools/decode_probe.py --steps 16 --report 2>&1 | grep -v amdgpu.ids | grep -E "probe|score_topk|beam_update|beam_topk"
  B="python bench.py --steps 8 --warmup 1 --no-cpu-baseline --no-extras --latency-runs 0"
  echo "== bench default (fused scoring, panel chain)"; timeout 300 $B 2>/dev/null | short
  echo "== bench default again"; timeout 300 $B --no-roofline 2>/dev/null | short
  echo "== bench SBK_X3P=0"; SBK_X3P=0 timeout 300 $B 2>/dev/null | short
  echo "== bench separate scoring kernels"; timeout 300 $B --no-roofline --knob 40=0 2>/dev/null | short
  echo "== bench streams 6 group 6"; timeout 300 $B --no-roofline --streams 6 --group 6 2>/dev/null | short
  echo "== bench streams 6 group 8"; timeout 300 $B --no-roofline --streams 6 --group 8 2>/dev/null | short
} 2>&1 | tee gpurun_out/r4_f.log
