#!/bin/bash
# Round 5, visit D (short): where the persistent step's 550 us go -- phase time stamps of workgroup 0 (knob 49).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  for g in 128 64; do timeout 120 python tools/latency_probe.py --runs 3 --knob 48=$g --stamps 2>&1 | grep -A14 "latency probe"; done
} 2>&1 | tee gpurun_out/r5_d.log
