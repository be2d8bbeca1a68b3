#!/bin/bash
# The native harness (tests/native/unit_tests = the reference's test/main.cpp + test/test_spmm.cu over include/spmm_adapter.hpp)
# on an arxiv-shaped graph in the course file format, kLen 32 and 256 -> stdout (for profiles/rNN_native_harness.log).
root=${GRAFT_REPO_ROOT:-$(pwd)}
D=$(mktemp -d)
python3 - <<PY
import sys; sys.path.insert(0, "$root")
from hpc_amd import synth, graph_io
ptr, idx = synth.csr_powerlaw(169343, 1166243 / 169343, 13155, seed=1, force_max=True)
graph_io.write_graph("$D", "arxiv_shaped", ptr, idx, text=True, dumps=False)
PY
for len in 32 256; do
  echo "\$ unit_tests --dataset arxiv_shaped --datadir <tmp> --len $len"
  $root/tests/native/unit_tests --dataset arxiv_shaped --datadir $D --len $len > $D/out.txt 2> $D/err.txt; echo "(exit $?)"
  cat $D/out.txt; echo "--- stderr ---"; cat $D/err.txt
done
rm -rf $D
