set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/probe
O=gpurun_out/probe
VAME_LIB=tools/libvame_hip_probe.so timeout 300 python tools/probe_clock.py > $O/gru_phase.log 2>&1
cat $O/gru_phase.log
