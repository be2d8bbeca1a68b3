#!/bin/bash
# EXPERIMENTS (not part of the library): cross-compiles the micro-benchmarks for gfx950 into scripts/experiments/build/
# (git-ignored binaries; they travel to the GPU box with the snapshot).   scripts/experiments/build.sh [name ...]
set -e
here=$(cd "$(dirname "$0")" && pwd); root=$(cd "$here/../.." && pwd)
mkdir -p "$here/build"
python3 "$here/make_hub_micro.py" > /dev/null
python3 "$here/make_block_micro.py" > /dev/null
names=("$@"); [ ${#names[@]} -eq 0 ] && names=(hub_micro)
for n in "${names[@]}"; do
    extra=""
    [ "$n" = block_micro ] && extra="$root/hpc_amd/csrc/preprocess_gpu.hip"
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -I "$root/hpc_amd/csrc" -I "$root/include" \
        "$here/$n.hip" $extra -o "$here/build/$n"
    echo "built $here/build/$n"
    if [ "$n" = hub_micro ] && [ -n "$HUB_MICRO_VARIANTS" ]; then      # counter-attribution builds (see make_hub_micro.py)
        for v in NO_B128 NO_B32 NO_BPERM; do
            /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -D$v -I "$root/hpc_amd/csrc" -I "$root/include" "$here/$n.hip" -o "$here/build/${n}_$v"
        done
    fi
done
