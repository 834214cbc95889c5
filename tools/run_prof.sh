cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
(cd /tmp && rm -rf /tmp/prof && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o r1 -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-roofline --latency-runs 0 > "$GRAFT_REPO_ROOT/gpurun_out/rocprof_run.log" 2>&1)
find /tmp/prof -name "*stats*.csv" -exec cp {} gpurun_out/ \;
grep '^{' gpurun_out/rocprof_run.log | cut -c1-200
