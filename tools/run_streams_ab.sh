#!/bin/bash
# A/B of workers x group (and env settings) on the bench; CASES_FILE (default tools/ab_cases.txt) = lines of "ENV|ARGS"
# (never feed cases through stdin: the remote shell has none and a `cat /dev/stdin` hangs until the time limit)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out
show='import sys,json; d=json.loads(sys.stdin.read()); c=d["config"]; print(d["value"], d["ms_per_step"], c.get("max_batch"), c.get("workers_per_gpu"), c.get("batches_per_grouped_search"))'
while IFS='|' read -r envs line; do
  [ -z "$line" ] && continue
  echo "== env[$envs] bench $line"
  env $envs timeout 600 python bench.py --warmup 1 --no-cpu-baseline --no-roofline --no-extras --latency-runs 0 $line 2> gpurun_out/ab_last.err | grep '^{' | tee -a gpurun_out/streams_ab.jsonl | python -c "$show" || tail -5 gpurun_out/ab_last.err
done < "${CASES_FILE:-tools/ab_cases.txt}"
