# rocprofv3 kernel trace of the default bench (10 timed + 3 warm-up + 3 profiling steps) -> compact per-call CSV + per-kernel stats
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-trace}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/raw -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/bench.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_digest.py trace $O/raw $O/kernel_trace.csv
python tools/rocprof_digest.py stats $O/raw $O/kernel_stats.csv
rm -rf $O/raw
grep '^{' $O/bench.log | cut -c1-200
