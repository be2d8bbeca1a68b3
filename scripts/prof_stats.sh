#!/bin/bash
# kernel trace + stats only (one rocprofv3 pass): scripts/prof_stats.sh <tag> <program> [args...]
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
CMD=("$@")
for i in "${!CMD[@]}"; do [[ -f "$root/${CMD[$i]}" ]] && CMD[$i]="$root/${CMD[$i]}"; done
d=$out/${tag}_stats; rm -rf $d
rocprofv3 --kernel-trace --stats --output-format csv -d $d -- "${CMD[@]}" > $d.log 2>&1 || echo "pass failed (see $d.log)"
cd $root
f=$(find $d -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cut -d, -f1-8 "$f" | head -12
