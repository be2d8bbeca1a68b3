#!/bin/bash
# round 4, call 4: LDS counters of the final hub kernel (once), use_graph tests, hub + capture tests, report table (with the graph column)
set -o pipefail
out=gpurun_out; mkdir -p $out
root=$(pwd)
echo "== hub_micro plain + LDS counters"
timeout -k 10 120 scripts/experiments/build/hub_micro > $out/c4_hub_micro.log 2>&1 || { cat $out/c4_hub_micro.log; exit 1; }
cat $out/c4_hub_micro.log
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $root/$out/c4_hub_lds -- $root/scripts/experiments/build/hub_micro > $root/$out/c4_hub_lds.log 2>&1 ) || { echo "LDS counter pass FAILED"; tail -20 $out/c4_hub_lds.log; exit 1; }
grep -h spmm_hub_stamped $(find $out/c4_hub_lds -name "*counter_collection.csv") | awk -F, '{print $2, $(NF-3), $(NF-2)}'
echo "== tests"
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -x -q -k "hub or fuzz or capturable or special or use_graph or panels or rmat" > $out/c4_tests.log 2>&1; rc=$?
tail -15 $out/c4_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python -m pytest tests/test_fullsize_gpu.py -x -q -k "hub_graphs or c2_power" > $out/c4_tests_full.log 2>&1; rc=$?
tail -5 $out/c4_tests_full.log
[ $rc -ne 0 ] && exit $rc
echo "== report table"
timeout -k 10 900 python scripts/report_table.py > $out/c4_report_table.md 2> $out/c4_report_table.err || { tail $out/c4_report_table.err; exit 1; }
cat $out/c4_report_table.md
