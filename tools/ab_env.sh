# A/B of one environment switch on the headline bench (interleaved runs on one box): bash tools/ab_env.sh VAR valA valB [reps]
cd $GRAFT_REPO_ROOT
V=$1; A=$2; B=$3; R=${4:-3}
for i in $(seq 1 $R); do
  for x in $A $B; do
    env $V=$x timeout 200 python bench.py --no-also --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('$V=$x', j['value'], j['ms_per_step'], j['roofline']['non_mfma_ms_per_step'])"
  done
done
