#!/bin/bash
# Round 6, visit C: the device timeline of a decoding step launch by launch (rocprofv3 --kernel-trace of tools/decode_probe.py, no HIP
# events in the stream), with the projections as shipped (knob 54 = 0) and launched twice (knob 54 = 1): which launches are slow in the
# step, by how much against their repeat, and what the gaps are.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
{
  for k in 0 1; do
    echo "== knob 54=$k"
    (cd /tmp && rm -rf /tmp/tl$k && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl$k -o t -- python $R/tools/decode_probe.py --steps 24 --reps 1 --knob 54=$k 2>&1 | grep "decode probe")
    f=$(find /tmp/tl$k -name "*kernel_trace.csv" | head -1)
    [ -n "$f" ] && python tools/decode_timeline.py "$f" 20
  done
} 2>&1 | tee gpurun_out/r6_c.log
