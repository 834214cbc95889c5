#!/bin/bash
# Round 6, visit H: the single-utterance latency path launch by launch (device timeline of a B = 1, 10-s search: persistent decoder
# step + scoring / CTC / beam launches), with the CTC scorer on the search's stream and on the helper stream; the wall-clock numbers
# of tools/latency_probe.py beside it; grid sizes of the persistent step (knob 48).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$PWD
{
  for ov in 0 3; do
    echo "== overlap $ov"
    timeout 120 python tools/latency_probe.py --runs 9 --overlap $ov 2>&1 | grep "latency probe"
    (cd /tmp && rm -rf /tmp/lt$ov && timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/lt$ov -o t -- python $R/tools/latency_probe.py --runs 3 --overlap $ov 2>&1 | grep "latency probe")
    f=$(find /tmp/lt$ov -name "*kernel_trace.csv" | head -1)
    [ -n "$f" ] && python tools/decode_timeline.py "$f" 30 decoder_step_persist
  done
  for g in 64 96 128 160 192 256; do echo "-- knob 48=$g"; timeout 120 python tools/latency_probe.py --runs 9 --overlap 3 --knob 48=$g 2>&1 | grep "latency probe"; done
} 2>&1 | tee gpurun_out/r6_h.log
