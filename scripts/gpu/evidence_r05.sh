#!/bin/bash
# Evidence, second half -- run AFTER scripts/gpu/profile_r05.sh's output has been condensed and committed (profiles/traffic_latest.json then carries the
# tree's source hash): default bench line, report table (with floor / use_graph columns), suite, native harness -> gpurun_out/r05_*
set -o pipefail
out=gpurun_out; mkdir -p $out
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $out/r05_bench_final.json 2> $out/r05_bench_final.err || { tail $out/r05_bench_final.err; exit 1; }
echo "bench done"
timeout -k 10 900 python scripts/report_table.py > $out/r05_report_table.md 2> $out/r05_report_table.err || { tail $out/r05_report_table.err; exit 1; }
echo "report table done"
timeout -k 10 900 python scripts/suite.py --ref --vendor > $out/r05_suite.jsonl 2> $out/r05_suite.err || { tail $out/r05_suite.err; exit 1; }
echo "suite done"
bash scripts/native_harness_log.sh > $out/r05_native_harness.log 2>&1
grep -E "bad =|ref_vs_oracle" $out/r05_native_harness.log | head
