#!/bin/bash
set -o pipefail
out=gpurun_out; root=$(pwd)
timeout -k 10 120 scripts/experiments/build/hub_micro 2>&1 | tee $out/c21_hub_micro.log || exit 1
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $root/$out/c21_hub_lds -- $root/scripts/experiments/build/hub_micro > $root/$out/c21_hub_lds.log 2>&1 ) || { echo "LDS pass failed"; exit 1; }
grep -h spmm_hub_stamped $(find $out/c21_hub_lds -name "*counter_collection.csv") | awk -F, '{print $2, $(NF-3), $(NF-2)}'
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q -k "hub or fuzz or special or negative_zero or flush or panels or rmat" 2>&1 | tail -4 || exit 1
HUB_VARIANTS=hub,hub16,hub32 timeout -k 10 600 python scripts/hub_bench.py am arxiv youtube rmat 2>&1 | grep -v amdgpu.ids | tee $out/c21_hub_bench.log
