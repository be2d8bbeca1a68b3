#!/bin/bash
# round 5: the committed evidence on the shipped sources -- kernel trace + stats and the separate PMC passes (scripts/prof.sh) for the headline and every
# `also` entry of the default bench line -> gpurun_out/r05_*.  Launch from a CLEAN, COMMITTED tree (scripts/summarize_prof.py names a dirty one as such).
# Condense afterwards, in the container (<steps> = warm-up + timed + the 20 runs of the reference's protocol that bench.py adds at N = 1):
#   python scripts/summarize_prof.py r05_c1 50 mi::spmm --traffic-key C1 --feat 128
#   python scripts/summarize_prof.py r05_c2 45 mi::spmm --traffic-key C2 --feat 128
#   python scripts/summarize_prof.py r05_c4 45 mi::spmm --traffic-key C4 --feat 256
#   python scripts/summarize_prof.py r05_n1024 33 mi::spmm --traffic-key C1_N1024 --feat 1024
#   python scripts/summarize_prof.py r05_longrows 45 mi::spmm --traffic-key LONG_ROWS --feat 128 --rows 131072
#   python scripts/summarize_prof.py r05_am32 45 mi::spmm --traffic-key AM32 --feat 32 --rows 881680
set -o pipefail
rm -rf gpurun_out/r05_c1_* gpurun_out/r05_c2_* gpurun_out/r05_c4_* gpurun_out/r05_n1024_* gpurun_out/r05_longrows_* gpurun_out/r05_am32_*
bash scripts/prof.sh r05_c1 python3 bench.py --no-cpu-baseline --no-also --steps 20 --warmup 10 && \
bash scripts/prof.sh r05_c2 python3 bench.py --config C2 --no-cpu-baseline --steps 20 --warmup 5 && \
bash scripts/prof.sh r05_c4 python3 bench.py --config C4 --no-cpu-baseline --steps 20 --warmup 5 && \
bash scripts/prof.sh r05_n1024 python3 bench.py --N 1024 --no-cpu-baseline --no-also --steps 10 --warmup 3 && \
bash scripts/prof.sh r05_longrows python3 bench.py --config LONGROWS --no-cpu-baseline --steps 20 --warmup 5 && \
bash scripts/prof.sh r05_am32 python3 bench.py --config am --N 32 --no-cpu-baseline --steps 20 --warmup 5 && \
ls gpurun_out | grep r05_ | grep -v regret | head -60
