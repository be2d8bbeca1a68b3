#!/bin/bash
# the driver's N > 1 launch line at FULL size with N ranks sharing the one GPU (MI_SPMM_SHARE_GPU=1: gloo + HIP IPC; a rehearsal of the code path, never a result)
# usage: scripts/gpu/rehearse_n.sh N [steps] [warmup]     (N <= 6: the box allows six processes on the GPU)
set -o pipefail
n=${1:-4}; steps=${2:-10}; warm=${3:-3}
out=gpurun_out; mkdir -p $out
MI_SPMM_WATCHDOG_S=${MI_SPMM_WATCHDOG_S:-60} MI_SPMM_SHARE_GPU=1 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus $n --steps $steps --warmup $warm > $out/${TAG:-r05}_launch_line_rehearsal_n$n.json 2> $out/${TAG:-r05}_launch_line_rehearsal_n$n.err
rc=$?
echo "rc=$rc"; grep "\[bench\]" $out/${TAG:-r05}_launch_line_rehearsal_n$n.err | tail -30
python - "$out/${TAG:-r05}_launch_line_rehearsal_n$n.json" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print({k: d.get(k) for k in ("n_gpus", "ms_per_step", "value", "speedup_vs_one_gpu", "rehearsal")}, d.get("exchange_selection"), (d.get("multi_gpu_breakdown") or {}).get("links"))
except Exception as e:
    print("no line:", e)
PY
exit $rc
