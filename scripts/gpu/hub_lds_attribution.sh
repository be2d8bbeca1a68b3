#!/bin/bash
# Which LDS instruction of spmm_hub makes SQ_LDS_BANK_CONFLICT count?  hub_micro and its three builds that drop one instruction kind each
# (HUB_MICRO_VARIANTS=1 scripts/experiments/build.sh hub_micro), each under the LDS counters -> profiles/r04_hub_lds_counters.txt
out=gpurun_out; root=$(pwd)
for v in "" _NO_B128 _NO_B32 _NO_BPERM; do
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE --output-format csv -d $root/$out/c23_attr$v -- $root/scripts/experiments/build/hub_micro$v > $root/$out/c23_attr$v.log 2>&1 )
  echo "== hub_micro$v"; grep -h "ticks/stage" $out/c23_attr$v.log | tail -1 | cut -c1-120
  grep -h spmm_hub_stamped $(find $out/c23_attr$v -name "*counter_collection.csv") | awk -F, '$2==8 || $2==9 || $2==7 {print $(NF-3), $(NF-2)}' | tail -4
done
