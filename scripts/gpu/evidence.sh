#!/bin/bash
# Evidence, second half -- run AFTER scripts/gpu/profile.sh's output has been condensed (scripts/summarize_prof.py ... --traffic-key) and committed, so that
# profiles/traffic_latest.json carries the tree's kernel-source hash: default bench line, report table, suite, native harness -> gpurun_out/r04_*
set -o pipefail
out=gpurun_out; mkdir -p $out
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $out/r04_bench_final.json 2> $out/r04_bench_final.err || { tail $out/r04_bench_final.err; exit 1; }
echo "bench done"
timeout -k 10 900 python scripts/report_table.py > $out/r04_report_table.md 2> $out/r04_report_table.err || { tail $out/r04_report_table.err; exit 1; }
echo "report table done"
timeout -k 10 900 python scripts/suite.py --ref --vendor > $out/r04_suite.jsonl 2> $out/r04_suite.err || { tail $out/r04_suite.err; exit 1; }
echo "suite done"
bash scripts/native_harness_log.sh > $out/r04_native_harness.log 2>&1
grep -E "bad =|ref_vs_oracle" $out/r04_native_harness.log | head
