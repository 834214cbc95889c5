#!/bin/bash
# Round 6, visit O: the single-utterance latency path -- the persistent step's grid barriers on a two-level arrival counter (knob 59) and
# the step as a plain instead of a cooperative launch (knob 47 = 2; same grid, same residency): tests, wall clock per step, timeline.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
{
  timeout 600 python -m pytest tests/test_model_parity.py tests/test_full_size_gpu.py -q -m gpu -x -k "persistent" 2>&1 | tail -3
  for cfg in "47=1 59=0" "47=1 59=1" "47=2 59=0" "47=2 59=1" "47=1 59=0" "47=1 59=1"; do
    set -- $cfg
    for ov in 0 3; do
      echo "-- knobs $1 $2 overlap $ov"
      timeout 120 python tools/latency_probe.py --runs 9 --overlap $ov --knob $1 --knob $2 2>&1 | grep "latency probe"
    done
  done
  for g in 64 96 128 192; do echo "-- knob 59=1 48=$g"; timeout 120 python tools/latency_probe.py --runs 9 --overlap 3 --knob 59=1 --knob 48=$g 2>&1 | grep "latency probe"; done
  echo "== timeline, knobs 47=2 59=1, overlap 0"
  (cd /tmp && rm -rf /tmp/to && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/to -o t -- python $R/tools/latency_probe.py --runs 3 --overlap 0 --knob 47=2 --knob 59=1 2>&1 | grep "latency probe")
  f=$(find /tmp/to -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python tools/decode_timeline.py "$f" 30 decoder_step_persist
} 2>&1 | tee gpurun_out/r6_o.log
