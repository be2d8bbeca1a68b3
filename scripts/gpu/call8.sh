#!/bin/bash
set -o pipefail
out=gpurun_out; mkdir -p $out
root=$(pwd)
for gf in 0 1; do
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $root/$out/c8_trace_$gf -- python3 $root/scripts/debug/overlap_trace.py arxiv 256 $gf > $root/$out/c8_trace_$gf.log 2>&1 ) || { tail -5 $out/c8_trace_$gf.log; exit 1; }
  grep -v amdgpu $out/c8_trace_$gf.log | grep "plain\|graph"
  f=$(find $out/c8_trace_$gf -name "*kernel_trace.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [r for r in rows if "mi::spmm" in r["Kernel_Name"]]
ks = ks[-9:]     # the last three steps of the final (plain, N = 256) handle
t0 = int(ks[0]["Start_Timestamp"])
for r in ks:
    print(f'{r["Kernel_Name"][:38]:38s} queue {r["Queue_Id"]:>3s}  start {int(r["Start_Timestamp"]) - t0:9d}  end {int(r["End_Timestamp"]) - t0:9d}  dur {int(r["End_Timestamp"]) - int(r["Start_Timestamp"]):8d}')
PY
done
