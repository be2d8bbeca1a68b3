#!/bin/bash
# Round 5, last pass: "fused_order" (which of the small-step kernel's first two roles leads the grid).  Tests of the small-step kernel, the A/B of the two
# orders on the short-step dataset shapes, the auto rule against hubs-first, and the report table's short-step rows -> gpurun_out/r05_fused_order_*
set -o pipefail
out=gpurun_out; mkdir -p $out
timeout -k 10 400 python -m pytest tests/test_small_step_gpu.py -x -q > $out/r05_fused_order_tests.txt 2>&1 || { tail -30 $out/r05_fused_order_tests.txt; exit 1; }
tail -2 $out/r05_fused_order_tests.txt
G=ddi-shuffled,ddi-community,collab-shuffled,collab-community,arxiv-shuffled,arxiv-community,youtube-shuffled,youtube-community,am-community
timeout -k 10 300 python scripts/debug/option_ab.py fused_order 1 2 --graphs $G --lens 32,128 > $out/r05_fused_order_ab.jsonl 2> $out/r05_fused_order_ab.err || { tail $out/r05_fused_order_ab.err; exit 1; }
echo "A/B 1 vs 2 done"
timeout -k 10 300 python scripts/debug/option_ab.py fused_order 1 0 --graphs $G --lens 32,128 > $out/r05_fused_order_auto.jsonl 2>> $out/r05_fused_order_ab.err || { tail $out/r05_fused_order_ab.err; exit 1; }
echo "A/B 1 vs auto done"
timeout -k 10 400 python scripts/report_table.py --only arxiv,collab,ddi,youtube --lens 32 > $out/r05_fused_order_report.md 2> $out/r05_fused_order_report.err || { tail $out/r05_fused_order_report.err; exit 1; }
cat $out/r05_fused_order_ab.jsonl $out/r05_fused_order_auto.jsonl | cut -c1-260
