#!/bin/bash
# Round 5, visit O: runs per utterance of the register-ring cross-attention (default ~1 024 waves; knob 8 = 5: ~2 048, 6: ~512) from
# 2 to 64 utterances per search, against the frame-per-thread kernel (knob 4 = 0); knob 47 = 0 keeps the <= 16-row cases off the persistent step.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
probe() { timeout 90 python tools/decode_probe.py --steps 16 --reps 3 --report --knob 47=0 "$@" 2>&1 | grep -E "decode probe|cross_"; }
{
  for u in 2 8 16 32; do
    for k in "4=0" "4=6"; do echo "== 1 x $u utterances, knob $k"; probe --batches 1 --utts $u --knob $k; done
    echo "== 1 x $u utterances, ring ~2048 waves"; probe --batches 1 --utts $u --knob 4=6 --knob 8=5
    echo "== 1 x $u utterances, ring ~512 waves"; probe --batches 1 --utts $u --knob 4=6 --knob 8=6
  done
  echo "== 2 x 32, ring ~1024"; probe --batches 2 --knob 4=6
  echo "== 2 x 32, ring ~512"; probe --batches 2 --knob 4=6 --knob 8=6
} 2>&1 | tee gpurun_out/r5_o.log
