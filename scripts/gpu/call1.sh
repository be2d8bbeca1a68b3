#!/bin/bash
# round 4, first GPU call: new tests, hub_micro (fixed args) plain + LDS counters ONCE, baselines for C4 / am-shaped
set -o pipefail
out=gpurun_out; mkdir -p $out
echo "== new tests" | tee $out/c1_tests.log
timeout -k 10 900 python -m pytest tests/test_fullsize_gpu.py -x -q -k "planted or c2_power" >> $out/c1_tests.log 2>&1 || { tail -30 $out/c1_tests.log; exit 1; }
tail -3 $out/c1_tests.log
echo "== hub_micro plain"
timeout -k 10 120 scripts/experiments/build/hub_micro > $out/c1_hub_micro.log 2>&1 || { cat $out/c1_hub_micro.log; exit 1; }
cat $out/c1_hub_micro.log
echo "== hub_micro LDS counters (once)"
root=$(pwd)
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $root/$out/c1_hub_lds -- $root/scripts/experiments/build/hub_micro > $root/$out/c1_hub_lds.log 2>&1 ) || { echo "LDS counter pass FAILED"; tail -20 $out/c1_hub_lds.log; exit 1; }
tail -5 $out/c1_hub_lds.log
find $out/c1_hub_lds -name "*counter_collection.csv" | head -1 | xargs -r head -20
echo "== baselines"
timeout -k 10 300 python bench.py --config C4 --steps 20 --warmup 5 --no-cpu-baseline > $out/c1_bench_c4.json 2> $out/c1_bench_c4.err && python -c "import json;d=json.load(open('$out/c1_bench_c4.json'));print('C4',d['ms_per_step'],d['roofline']['frac'])" || exit 1
for n in 32 128; do
timeout -k 10 300 python bench.py --config am --N $n --steps 20 --warmup 5 --no-cpu-baseline > $out/c1_bench_am$n.json 2> $out/c1_bench_am$n.err && python -c "import json;d=json.load(open('$out/c1_bench_am$n.json'));print('am N=$n',d['ms_per_step'],d['roofline']['frac'],d['config']['options'])" || exit 1
done
