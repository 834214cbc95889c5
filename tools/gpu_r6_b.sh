#!/bin/bash
# Round 6, visit B: what does a projection of the decoding step lose to COLD operands?  (gemm_x3r: 28 us per launch in the step, 13-19 in
# isolation; visit A: non-temporal loads on the streamed K/V and CTC data do not move it.)  Measurement-only knob 54: 1 = every
# projection launched twice (the repeat finds its operands where the first launch left them), 2 = the weight panel read into every XCD's
# L2 by a launch in front, 3 = the panel and the A rows.  Then nt-load masks 0 / 3 / 7 again (the CTC reading of visit A was ambiguous).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  for k in 0 1 2 3; do echo "-- knob 54=$k"; timeout 120 python tools/decode_probe.py --steps 24 --reps 2 --report --knob 54=$k 2>&1 | grep -v amdgpu.ids | head -12; done
  for k in 0 3 7 0 3 7; do echo "-- knob 53=$k"; timeout 120 python tools/decode_probe.py --steps 24 --reps 2 --report --knob 53=$k 2>&1 | grep -v amdgpu.ids | head -6; done
} 2>&1 | tee gpurun_out/r6_b.log
