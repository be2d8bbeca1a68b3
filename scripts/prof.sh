#!/bin/bash
# rocprofv3 passes for one command (GPU box): kernel trace + stats, then the PMC passes of MI355X_MICROARCH.md
# (separate --pmc runs, never combined with a trace).  Output: gpurun_out/<tag>_{stats,fetch,write,tcc,sq}/ (csv).
#   scripts/prof.sh r03_c4 python3 bench.py --config C4 --steps 20 --warmup 5 --no-cpu-baseline
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
run() { d=$out/${tag}_$1; shift; rm -rf $d; rocprofv3 "$@" --output-format csv -d $d -- "${CMD[@]}" > $d.log 2>&1 || echo "pass failed: $d (see $d.log)"; }
CMD=("$@")
# absolute path for the script argument (we run from /tmp)
for i in "${!CMD[@]}"; do [[ -f "$root/${CMD[$i]}" ]] && CMD[$i]="$root/${CMD[$i]}"; done
run stats --kernel-trace --stats
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
run tcc --pmc TCC_HIT_sum TCC_MISS_sum
run sq --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE
cd $root
find $out/${tag}_stats -name "*kernel_stats.csv" | head -2
