#!/bin/bash
# Round 5, visit N: the register-ring cross-attention at smaller searches (1 / 2 batches of 32) against the frame-per-thread kernel,
# ring depth 2 / 3 / 4 (knob 4 = 8 / 6 / 9), runs per utterance (knob 8 = 3: one, 5: ~4 096 waves, 6: ~1 024 waves).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
probe() { timeout 90 python tools/decode_probe.py --steps 16 --reps 3 --report "$@" 2>&1 | grep -E "decode probe|cross_"; }
{
  for nb in 1 2; do for k in "4=7" "4=6" "4=0"; do echo "== $nb batches, knob $k"; probe --batches $nb --knob $k; done; done
  echo "== 1 batch, ring, ~4096 waves"; probe --batches 1 --knob 4=6 --knob 8=5
  echo "== 1 batch, ring, ~1024 waves"; probe --batches 1 --knob 4=6 --knob 8=6
  echo "== 4 batches, ring depth 2"; probe --knob 4=8
  echo "== 4 batches, ring depth 3"; probe --knob 4=6
  echo "== 4 batches, ring depth 3, 2 runs + merge"; probe --knob 4=6 --knob 8=5
  echo "== 8 batches (256 utterances), ring depth 3"; probe --batches 8 --knob 4=6
  echo "== 8 batches (256 utterances), default"; probe --batches 8 --knob 4=7
} 2>&1 | tee gpurun_out/r5_n.log
