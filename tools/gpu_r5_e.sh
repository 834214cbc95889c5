#!/bin/bash
# Round 5, visit E (short): the persistent step, third version (group-parallel LayerNorm, a workgroup per (row, head) in both
# attentions, weight prefetch across tiles): parity test, phase stamps, latency.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  timeout 300 python -m pytest tests/ -q -m gpu -k "persistent_few_row" 2>&1 | tail -3
  for g in 128 96; do timeout 120 python tools/latency_probe.py --runs 5 --knob 48=$g --stamps 2>&1 | grep -A14 "latency probe"; done
  timeout 120 python tools/latency_probe.py --runs 5 --overlap 3 2>&1 | grep "latency probe"
  timeout 120 python tools/latency_probe.py --runs 5 --overlap 3 --seconds 28 2>&1 | grep "latency probe"
  timeout 120 python tools/latency_probe.py --runs 5 --overlap 3 --seconds 28 --knob 47=0 2>&1 | grep "latency probe"
} 2>&1 | tee gpurun_out/r5_e.log
