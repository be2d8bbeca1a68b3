#!/bin/bash
# round 4, call 2: chain_patterns micro (DPP a-values), bench contract tests (safe-first auto, watchdog)
set -o pipefail
out=gpurun_out; mkdir -p $out
echo "== chain_patterns"
timeout -k 10 120 scripts/experiments/build/chain_patterns > $out/c2_chain_patterns.log 2>&1 || { tail -30 $out/c2_chain_patterns.log; exit 1; }
cat $out/c2_chain_patterns.log
echo "== bench contract tests"
timeout -k 10 1000 python -m pytest tests/test_bench_contract_gpu.py -x -q > $out/c2_bench_tests.log 2>&1; rc=$?
tail -40 $out/c2_bench_tests.log
exit $rc
