#!/bin/bash
set -o pipefail
out=gpurun_out; mkdir -p $out
timeout -k 10 300 python scripts/debug/overlap_probe.py youtube 32 2>&1 | grep -v amdgpu.ids | tee $out/c6_overlap_youtube32.log
echo "== C4 early stores"
for i in 1 2; do timeout -k 10 300 python bench.py --config C4 --steps 20 --warmup 5 --no-cpu-baseline --check 2> $out/c6_c4.err | python -c "import sys,json;d=json.loads(sys.stdin.read());print('C4',d['ms_per_step'],d['roofline']['frac'],d['check'])"; done
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q -k "block" 2>&1 | tail -3
timeout -k 10 600 python -m pytest tests/test_dist_shared_gpu.py -x -q 2>&1 | tail -3
