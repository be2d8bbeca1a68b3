#!/bin/bash
# extra SQ counters for the sweep experiment (A/B library)
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
export MI_SPMM_LIB=$root/hpc_amd/libmi_spmm_ablate.so
cd /tmp && export TMPDIR=/tmp
CMD=(python3 $root/scripts/c4_ab.py --one 2 4 --steps 10 --opt block_sweep=1 --opt block_sweep_cols=$1 --opt block_sweep_min_tracks=3)
run() { d=$out/swp_$1; shift; rm -rf $d; rocprofv3 "$@" --output-format csv -d $d -- "${CMD[@]}" > $d.log 2>&1 || echo "pass failed: $d"; }
run a --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE
run b --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU
run c --kernel-trace --stats
