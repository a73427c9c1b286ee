# rocprofv3 kernel trace of the batch-256 train step (the reference's stock config) -> per-dispatch CSV for a timeline / gap analysis
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-trace_b256}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/raw -- python $GRAFT_REPO_ROOT/bench.py --batch ${2:-256} --steps 20 --warmup 5 --no-cpu-baseline --no-also ${3:-} > $GRAFT_REPO_ROOT/$O/bench.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_digest.py trace $O/raw $O/kernel_trace.csv
python tools/rocprof_digest.py stats $O/raw $O/kernel_stats.csv
rm -rf $O/raw
grep '^{' $O/bench.log | cut -c1-300
