#!/bin/bash
# round 4, call 3: the new chain loop (DPP a values, pair-wise flags): micro (two lane maps), hub tests, hub bench
set -o pipefail
out=gpurun_out; mkdir -p $out
for b in hub_micro hub_micro_map1; do
  echo "== $b"
  timeout -k 10 120 scripts/experiments/build/$b > $out/c3_$b.log 2>&1; rc=$?
  cat $out/c3_$b.log
  [ $rc -ne 0 ] && { echo "$b failed rc=$rc"; exit 1; }
done
echo "== hub tests"
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -x -q -k "hub or fuzz or capturable or special" > $out/c3_tests.log 2>&1; rc=$?
tail -15 $out/c3_tests.log
[ $rc -ne 0 ] && exit $rc
echo "== hub bench"
HUB_VARIANTS=hub,split timeout -k 10 600 python scripts/hub_bench.py am arxiv youtube rmat > $out/c3_hub_bench.log 2>&1 || { tail $out/c3_hub_bench.log; exit 1; }
cat $out/c3_hub_bench.log
