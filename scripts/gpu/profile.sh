#!/bin/bash
# round 4: the committed evidence, regenerated on the shipped kernels: kernel trace + stats and the PMC passes (separate runs) for C1 and C4,
# kernel stats for the am-shaped graph at N = 128 and 32 (the hub kernel on a chain-bound row)
set -o pipefail
bash scripts/prof.sh r04_c1 python3 bench.py --no-cpu-baseline --no-also --steps 20 --warmup 10 && \
bash scripts/prof.sh r04_c4 python3 bench.py --config C4 --no-cpu-baseline --steps 20 --warmup 5 && \
bash scripts/prof_stats.sh r04_am python3 bench.py --config am --no-cpu-baseline --steps 20 --warmup 5 && \
bash scripts/prof_stats.sh r04_am32 python3 bench.py --config am --N 32 --no-cpu-baseline --steps 20 --warmup 5 && \
bash scripts/prof_stats.sh r04_c2 python3 bench.py --config C2 --no-cpu-baseline --steps 20 --warmup 5 && \
ls gpurun_out | grep r04_ | head -40
