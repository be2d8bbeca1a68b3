#!/bin/bash
# variants of scripts/debug/ipc_open_probe.py in one call: one size per launch (a size that hangs costs its own timeout only)
# usage: ipc_probe_runs.sh WORLD size_MiB ...
out=gpurun_out; mkdir -p $out
echo "HSA_ENABLE_IPC_MODE_LEGACY=$HSA_ENABLE_IPC_MODE_LEGACY"
w=$1; shift
run() { tag=$1; shift; timeout -k 10 60 python -m torch.distributed.run --nnodes=1 --nproc-per-node $w --master-addr 127.0.0.1 --master-port 29533 scripts/debug/ipc_open_probe.py "$@" > $out/ipc_probe_$tag.log 2>&1; echo "== $tag rc=$?"; grep -o "rank [0-9]: [0-9]* MiB: rank [0-9]'s opened in [0-9.]* s (status [0-9]*)" $out/ipc_probe_$tag.log | sort | head -60; grep -o "rank [0-9]: [0-9]* MiB: rank [0-9].s first.*" $out/ipc_probe_$tag.log | sort | head -3; grep -o "rank [0-9]: [0-9]* MiB: 2-D copy.*" $out/ipc_probe_$tag.log | sort | head -40; }
for s in "$@"; do run w${w}_$s $s; done
