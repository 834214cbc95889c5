#!/bin/bash
# A/B of batches in flight x hardware queues (GPU_MAX_HW_QUEUES) on the bench; CASES = lines of "ENV|ARGS"
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
show='import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"].get("batch"), d["config"].get("batches_in_flight_per_gpu"))'
while IFS='|' read -r envs line; do
  [ -z "$line" ] && continue
  echo "== env[$envs] bench $line"
  env $envs timeout 600 python bench.py --warmup 1 --no-cpu-baseline --no-roofline --latency-runs 0 $line 2>&1 | grep '^{' | tee -a gpurun_out/streams_ab.jsonl | python -c "$show"
done <<< "$CASES"
