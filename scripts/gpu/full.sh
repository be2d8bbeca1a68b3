#!/bin/bash
# the driver's round-end sequence: every gpu test, smoke, the default bench line
set -o pipefail
out=gpurun_out; mkdir -p $out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $out/full_tests.log 2>&1; rc=$?
tail -8 $out/full_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $out/full_bench.json 2> $out/full_bench.err || { tail $out/full_bench.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/full_bench.json"))
print("C1", d["ms_per_step"], d["roofline"]["frac"], "traffic_stale", d["roofline"]["traffic_source"]["traffic_stale"], "cpu", d["cpu_baseline"]["value"])
for k, e in d["also"].items():
    print(k, e.get("ms_per_step"), e.get("roofline", {}).get("frac"), (e.get("roofline", {}).get("traffic_source") or {}).get("traffic_stale"))
PY
