#!/bin/bash
# Debug: the native harness on a power-law graph, with the library's CSR-rejection detail
set -x
D=$(mktemp -d)
python - <<PY
import sys; sys.path.insert(0, "$GRAFT_REPO_ROOT")
from hpc_amd import synth, graph_io
ptr, idx = synth.csr_powerlaw(30000, 20.0, 1500, seed=12)
graph_io.write_graph("$D", "syn", ptr, idx, text=True, dumps=False)
PY
MI_SPMM_DEBUG=1 $GRAFT_REPO_ROOT/tests/native/unit_tests --dataset syn --datadir $D --len 32 2>&1 | tail -15
