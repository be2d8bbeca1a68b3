#!/bin/bash
# a soak of the three fuzz tests on a seed other than the suite's (MI_SPMM_FUZZ_SEED), after the whole -m gpu suite has passed
# usage: scripts/gpu/soak.sh [seed] [shape cases] [hub cases] [block cases]
set -o pipefail
out=gpurun_out; mkdir -p $out
seed=${1:-11}; shapes=${2:-1500}; hubs=${3:-300}; blocks=${4:-300}
timeout -k 10 700 python -m pytest tests -x -q -m gpu > $out/soak_suite.log 2>&1; rc=$?
tail -5 $out/soak_suite.log
[ $rc -ne 0 ] && exit $rc
MI_SPMM_FUZZ_SEED=$seed MI_SPMM_FUZZ_CASES=$shapes MI_SPMM_HUB_FUZZ_CASES=$hubs MI_SPMM_BLOCK_FUZZ_CASES=$blocks \
  timeout -k 10 420 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "fuzz or random_lengths" > $out/soak_fuzz_seed$seed.log 2>&1; rc=$?
tail -5 $out/soak_fuzz_seed$seed.log
exit $rc
