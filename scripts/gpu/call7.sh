#!/bin/bash
set -o pipefail
out=gpurun_out; mkdir -p $out
for sk in "" graph vendor student graph,vendor,student; do
  echo "== report_table --only arxiv --skip '$sk'"
  timeout -k 10 300 python scripts/report_table.py --only arxiv --skip "$sk" 2>/dev/null | grep "^| arxiv" | cut -d'|' -f2,6,7,9,14,15
done
