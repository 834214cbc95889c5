#!/bin/bash
# Round 5, visit J: (1) several threads run single-utterance searches (cooperative persistent steps) at once beside a contraction load:
# equal results, no hang (under timeout); (2) the bench with the latency leg in a process of its own.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
  echo "== cooperative launches from several threads"
  timeout 150 python tools/coop_concurrency_check.py --threads 4 --rounds 12 2>&1 | tail -2; echo "rc $?"
  timeout 150 python tools/coop_concurrency_check.py --threads 8 --rounds 6 2>&1 | tail -2; echo "rc $?"
  echo "== bench, 8 steps, latency leg in its own process"
  timeout 900 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>>gpurun_out/r5j.err | tail -1 > gpurun_out/r5j_bench.json
  python -c "
import json; d = json.load(open('gpurun_out/r5j_bench.json'))
for k in ('value', 'p50_latency_ms', 'p50_latency_ms_by_mode', 'decode_step_ms', 'launches_per_decode_step'): print(k, d.get(k))
print(d['config'].get('gpu_memory_reserved_gb'), d['config'].get('latency_leg'), {k: v for k, v in d['config'].items() if 'error' in k})
r = d.get('roofline') or {}; print(r.get('frac'), r.get('traffic'), r.get('per_shape_isolated'))"
} 2>&1 | tee gpurun_out/r5_j.log
