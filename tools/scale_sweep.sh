#!/bin/bash
# One call for the first visit to a multi-GPU node (BASELINE configs[2] and configs[4]): the train line at every rank count, then the
# window-index-sharded embedding line on all of them.  One JSON line per run on stdout; everything else goes to stderr.
#   bash tools/scale_sweep.sh [max_gpus] [embed windows per GPU] [-- extra bench.py arguments for the train lines]
#   default: 1 2 4 8 ranks (as many as are visible), 1,250,000 windows per GPU (10 M on 8 GPUs: configs[4])
# bench.py starts its N ranks itself (python -m torch.distributed.run ... --nproc-per-node N, one process per GPU, RCCL over xGMI);
# VAME_SCALE_PYTHON overrides the launcher (the CPU test-suite runs this script on its 2-rank gloo harness with a tiny model).
set -u
cd "$(dirname "$0")/.."
MAX=${1:-8}; [ $# -gt 0 ] && shift
EW=${1:-1250000}; [ $# -gt 0 ] && shift
[ "${1:-}" = "--" ] && shift
PY=${VAME_SCALE_PYTHON:-python}
NG=${VAME_SCALE_VISIBLE:-$($PY -c "import torch; print(torch.cuda.device_count())" 2>/dev/null || echo 1)}
[ "$NG" -lt 1 ] && NG=1
[ "$MAX" -gt "$NG" ] && MAX=$NG
for n in 1 2 4 8; do
  [ "$n" -gt "$MAX" ] && break
  echo "== train, $n GPU(s)" >&2
  $PY bench.py --gpus $n --no-also "$@" || echo "{\"error\": \"train line failed at $n ranks\"}"
done
echo "== embedding, $MAX GPU(s), $EW windows each" >&2
$PY bench.py --mode embed --gpus $MAX --embed-windows $EW ${VAME_SCALE_EMBED_ARGS:-} || echo "{\"error\": \"embedding line failed at $MAX ranks\"}"
