# GPU box: the weight-stationary split-bf16 GRU forward probe over group counts, store policies and the BPTT stash (profiles/r06_rec_split_probe.txt)
cd $GRAFT_REPO_ROOT
O=gpurun_out/rec_probe.txt
hipcc --offload-arch=gfx950 -O3 -w tools/rec_split_probe.hip -o /tmp/rec_split_probe || exit 1
: > $O
for args in "8 30 1 0 -40 36" "16 30 1 0 -40 4" "16 30 1 0 -40 36" "24 30 1 0 -40 4" "32 30 1 0 -40 36" "32 30 0 0 -40 32"; do
  timeout 120 /tmp/rec_split_probe $args 2>&1 | grep -v amdgpu.ids >> $O
done
cat $O
