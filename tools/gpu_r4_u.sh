#!/bin/bash
# Round 4, visit U (staged change): kernel trace of a decoding step with the LayerNorms inside the projections
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
{
  (cd /tmp && rm -rf /tmp/dtr && timeout 45 rocprofv3 --kernel-trace --output-format csv -d /tmp/dtr -o t -- python $R/tools/decode_probe.py --steps 16 --reps 2 --knob 45=1 2>&1 | grep "decode probe")
  f=$(find /tmp/dtr -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/decode_trace.py "$f" 32 | head -16
} 2>&1 | tee gpurun_out/r4_u.log
