set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/probe
O=gpurun_out/probe
timeout 120 ./tools/probe_gemm calib > $O/calib.log 2>&1
VAME_LIB=tools/libvame_hip_probe.so timeout 300 python tools/probe_clock.py > $O/gru_clock.log 2>&1
cat $O/calib.log $O/gru_clock.log
