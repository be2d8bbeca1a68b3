#!/bin/bash
# the two `also` entries of the default bench line that had no measured traffic: C2 and C1's CSR at N = 1024 (configs[3]'s one-GPU leg):
# kernel trace + stats, then the separate PMC passes (scripts/prof.sh) -> gpurun_out/r04_c2_* and gpurun_out/r04_n1024_*; condense with
#   python scripts/summarize_prof.py r04_c2 45 mi::spmm --traffic-key C2 --feat 128
#   python scripts/summarize_prof.py r04_n1024 33 mi::spmm --traffic-key C1_N1024 --feat 1024
#   python scripts/summarize_prof.py r04_longrows 45 mi::spmm --traffic-key LONG_ROWS --feat 128 --rows 131072
# (steps = warm-up + timed + the 20 runs of the reference-protocol timing)
set -o pipefail
bash scripts/prof.sh r04_c2 python3 bench.py --config C2 --no-cpu-baseline --steps 20 --warmup 5 && \
bash scripts/prof.sh r04_n1024 python3 bench.py --N 1024 --no-cpu-baseline --no-also --steps 10 --warmup 3 && \
bash scripts/prof.sh r04_longrows python3 bench.py --config LONGROWS --no-cpu-baseline --steps 20 --warmup 5 && \
ls gpurun_out | grep -E "r04_(c2|n1024|longrows)_" | head -40
