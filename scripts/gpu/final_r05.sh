#!/bin/bash
# round 5, last call: the driver's sequence (every -m gpu test, smoke, bench line), then the evidence set, then the regret table of the fitted graphs
set -o pipefail
bash scripts/gpu/full.sh || exit 1
bash scripts/gpu/evidence_r05.sh || exit 1
timeout -k 10 600 python scripts/regret.py > gpurun_out/r05_regret.jsonl 2> gpurun_out/r05_regret.err; tail -2 gpurun_out/r05_regret.err
