"""Run tests/native/unit_tests (the reference's harness re-stated over include/spmm_adapter.hpp) on an
arxiv-shaped graph written in the course file format, and keep its log (gtest lines on stdout, dbg-shaped
lines on stderr) as evidence of log compatibility.   python scripts/native_log_sample.py > gpurun_out/native_harness.log"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hpc_amd import graph_io, synth
exe = os.path.join(ROOT, "tests", "native", "unit_tests")
with tempfile.TemporaryDirectory() as d:
    ptr, idx = synth.csr_powerlaw(169_343, 1_166_243 / 169_343, 13_155, seed=7, force_max=True)
    graph_io.write_graph(d, "arxiv_shaped", ptr, idx, text=True, dumps=False)
    for n in (32, 256):
        r = subprocess.run([exe, "--dataset", "arxiv_shaped", "--datadir", d, "--len", str(n)], capture_output=True, text=True, timeout=600)
        print(f"$ unit_tests --dataset arxiv_shaped --datadir <tmp> --len {n}   (exit {r.returncode})")
        print(r.stdout.rstrip()); print("--- stderr ---"); print(r.stderr.replace(d, "<tmp>").rstrip()); print()
