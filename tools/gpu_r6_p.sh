#!/bin/bash
# Round 6, visit P: where the GPU suite's time goes (the driver's tier has a 1 200-s limit; the suite took 874 s after this round's additions).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  timeout 1700 python -m pytest tests/ -q -m gpu --durations=40 2>&1 | tail -60
} 2>&1 | tee gpurun_out/r6_p.log
