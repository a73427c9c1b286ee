# small-batch A/B of the step's side-stream overlaps: bash tools/ab_small.sh <batch>
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for x in "0 0" "1 1"; do
    set -- $x
    VAME_AMD_NUC_SIDE=$1 VAME_AMD_BWD_OVERLAP=$2 timeout 200 python bench.py --batch ${BATCH:-256} --no-cpu-baseline --steps 60 --warmup 15 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.readline()); print('overlaps=$1', j['value'], j['ms_per_step'])"
  done
done
