cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r06t
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-also"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw_trace -- $B --steps 10 --warmup 3 > $O/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_digest.py trace $O/raw_trace $O/kernel_trace.csv
python tools/rocprof_digest.py stats $O/raw_trace $O/kernel_stats.csv
rm -rf $O/raw_trace
bash tools/trace_b256.sh r06t/b256_graph 256 --graph > /dev/null 2>&1; cp $O/b256_graph/kernel_trace.csv $O/b256_graph_kernel_trace.csv; rm -rf $O/b256_graph
python tools/narrow_gemms.py 2>&1 | grep -v amdgpu > $O/narrow_gemms.txt
cat $O/narrow_gemms.txt
