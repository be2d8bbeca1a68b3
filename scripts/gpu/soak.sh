#!/bin/bash
# a soak of the three fuzz tests on seeds other than the suite's (MI_SPMM_FUZZ_SEED); with SUITE=1 the whole -m gpu suite runs first
# usage: [SUITE=1] [FUSED=1] scripts/gpu/soak.sh "seed seed ..." [shape cases] [hub cases] [block cases]
#   FUSED_ORDER=1|2 (with FUSED=1): the small-step kernel's hub / segment workgroups lead its grid ("fused_order"; default auto)
#   FUSED=1: the fuzz cases' eligible steps go through the small-step kernel ("fused_step" = 1 as the session default: tests/conftest.py)
set -o pipefail
out=gpurun_out; mkdir -p $out
seeds=${1:-11}; shapes=${2:-1500}; hubs=${3:-300}; blocks=${4:-300}
if [ "${SUITE:-0}" = 1 ]; then
  timeout -k 10 700 python -m pytest tests -x -q -m gpu > $out/soak_suite.log 2>&1; rc=$?
  tail -5 $out/soak_suite.log
  [ $rc -ne 0 ] && exit $rc
fi
for seed in $seeds; do
  [ "${FUSED:-0}" = 1 ] && export MI_SPMM_TEST_FUSED=1
  [ -n "${FUSED_ORDER:-}" ] && export MI_SPMM_TEST_FUSED_ORDER=$FUSED_ORDER
  export MI_SPMM_FUZZ_SEED=$seed MI_SPMM_FUZZ_CASES=$shapes MI_SPMM_HUB_FUZZ_CASES=$hubs MI_SPMM_BLOCK_FUZZ_CASES=$blocks
  echo "seed $seed: $shapes shape cases, $hubs hub cases, $blocks block cases"
  timeout -k 10 500 python -m pytest tests/test_parity_gpu.py -x -q -m gpu --durations=3 -k "fuzz or random_lengths" > $out/soak_fuzz_seed$seed${FUSED:+_fused}${FUSED_ORDER:+_o$FUSED_ORDER}.log 2>&1; rc=$?
  tail -12 $out/soak_fuzz_seed$seed${FUSED:+_fused}${FUSED_ORDER:+_o$FUSED_ORDER}.log
  [ $rc -ne 0 ] && exit $rc
done
exit 0
