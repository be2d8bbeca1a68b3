"""The course graph format (PA4/workspace/src/data.cu:3-66): text parse, dump caches, error paths."""
import os

import numpy as np
import pytest

from hpc_amd import graph_io, synth


def test_text_roundtrip_and_cache(tmp_path):
    ptr, idx = synth.csr_uniform(200, 0, 9, seed=3)
    graph_io.write_graph(str(tmp_path), "toy", ptr, idx)
    nv, ne, p, i = graph_io.load_graph(str(tmp_path), "toy")
    assert (nv, ne) == (200, idx.size) and np.array_equal(p, ptr) and np.array_equal(i, idx)
    # first load wrote the caches next to the text file (data.cu:33-38, 60-64)
    assert os.path.exists(tmp_path / "toy.graph.ptrdump") and os.path.exists(tmp_path / "toy.graph.edgedump")
    assert np.array_equal(np.fromfile(tmp_path / "toy.graph.ptrdump", dtype="<i4"), ptr)
    # caches alone are enough (util.cu:64-69)
    os.remove(tmp_path / "toy.graph")
    nv2, ne2, p2, i2 = graph_io.load_graph(str(tmp_path), "toy")
    assert np.array_equal(p2, ptr) and np.array_equal(i2, idx)


def test_only_ptrdump_cached_rereads_text(tmp_path):
    """The state in which the reference reads from a closed FILE* (SURVEY.md H7)."""
    ptr, idx = synth.csr_uniform(50, 1, 5, seed=4)
    graph_io.write_graph(str(tmp_path), "g", ptr, idx)
    ptr.astype("<i4").tofile(tmp_path / "g.graph.ptrdump")
    nv, ne, p, i = graph_io.load_graph(str(tmp_path), "g")
    assert np.array_equal(i, idx)


def test_errors(tmp_path):
    with pytest.raises(FileNotFoundError):
        graph_io.load_graph(str(tmp_path), "missing")
    ptr, idx = synth.csr_uniform(20, 1, 3, seed=5)
    graph_io.write_graph(str(tmp_path), "bad", ptr, idx)
    with open(tmp_path / "bad.config", "w") as f:
        f.write(f"20 {idx.size + 1}\n")          # num_e disagrees with indptr[num_v]  (data.cu:40-45)
    with pytest.raises(ValueError):
        graph_io.load_graph(str(tmp_path), "bad", write_cache=False)
