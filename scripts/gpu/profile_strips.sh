#!/bin/bash
# column strips of the exact segments under the profiler: protein-shaped graph, N = 128, strips by the auto rule and strips off --
# kernel trace + stats, then the separate PMC passes (scripts/prof.sh) -> gpurun_out/r04_strips_on_*, r04_strips_off_*; condense with
#   python scripts/summarize_prof.py r04_strips_on 45 mi::spmm ;  python scripts/summarize_prof.py r04_strips_off 45 mi::spmm
set -o pipefail
bash scripts/prof.sh r04_strips_on python3 bench.py --config protein --N 128 --no-cpu-baseline --steps 20 --warmup 5 && \
bash scripts/prof.sh r04_strips_off python3 bench.py --config protein --N 128 --no-cpu-baseline --steps 20 --warmup 5 --opt col_strips=1 && \
ls gpurun_out | grep -E "r04_strips_" | head -40
