#!/bin/bash
# the tuner with the hub threshold among its candidates: its test, then the regret of "autotune" = 1 on the fitted set and on the hold-out graphs
set -o pipefail
out=gpurun_out; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_parity_gpu.py -x -q -k "autotune" 2>&1 | tail -3 || exit 1
timeout -k 10 1000 python scripts/regret.py --holdout --autotune > $out/r05_regret_holdout_autotune.jsonl 2> $out/r05_regret_holdout_autotune.err || { tail $out/r05_regret_holdout_autotune.err; exit 1; }
echo holdout autotune done
