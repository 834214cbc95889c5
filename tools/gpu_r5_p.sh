#!/bin/bash
# Round 5, visit P: whole-prefix self-attention (knob 50 = 4 / 8 / 16 positions x 4 per pass) and the few-row projection's bias /
# residual loads in front of its partial-tile exchange (knob 46), per kernel class at 60 decoding steps; the unconditional
# up-front loads of the fused scoring kernel / LayerNorm kernels / ring query rows are in every run (no knob).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
probe() { timeout 120 python tools/decode_probe.py --steps 60 --reps 2 --report "$@" 2>&1 | grep -E "decode probe|self_attn|gemm_x3r|gemm_ln_x3r|score_topk|cross_attn|layernorm|ctc_score"; }
{
  timeout 400 python -m pytest tests/test_kernels.py tests/test_model_parity.py -q -m gpu -x -k "whole_prefix or register_ring or fused_scoring or layernorm or x3r or cross_attention or wide_beam or persistent" 2>&1 | tail -2
  for k in "50=0" "50=4" "50=8" "50=16" "50=0" "50=8"; do echo "== knob $k"; probe --knob $k; done
  for k in "46=0" "46=1" "46=0" "46=1"; do echo "== knob $k (50=0)"; probe --knob $k | grep -E "decode probe|x3r"; done
} 2>&1 | tee gpurun_out/r5_p.log
