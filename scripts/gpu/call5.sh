#!/bin/bash
set -o pipefail
out=gpurun_out; mkdir -p $out
root=$(pwd)
echo "== overlap probe"
timeout -k 10 300 python scripts/debug/overlap_probe.py youtube 32 2>&1 | grep -v amdgpu.ids | tee $out/c5_overlap_youtube32.log
timeout -k 10 300 python scripts/debug/overlap_probe.py am 256 2>&1 | grep -v amdgpu.ids | tee $out/c5_overlap_am256.log
echo "== chain_patterns LDS counters (which instruction conflicts?)"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $root/$out/c5_chain_lds -- $root/scripts/experiments/build/chain_patterns > $root/$out/c5_chain_lds.log 2>&1 ) || { echo "pass FAILED"; tail -20 $out/c5_chain_lds.log; exit 1; }
grep -h "k[0-9]*(" $(find $out/c5_chain_lds -name "*counter_collection.csv") | awk -F, '{print $2, $9, $(NF-3), $(NF-2)}' | head -120
echo "== new test"
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q -k "skip_the_hub or use_graph or panels" 2>&1 | tail -5
