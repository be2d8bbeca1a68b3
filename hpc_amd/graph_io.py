"""The course's on-disk graph format -- PA4/workspace/src/data.cu:3-66, src/util.cu:47-69.

  <datadir>/<dset>.config          text: "num_v num_e"
  <datadir>/<dset>.graph           text: num_v+1 row pointers, then num_e column indices
  <datadir>/<dset>.graph.ptrdump   cache: raw little-endian int32[num_v+1]   (data.cu:21-39)
  <datadir>/<dset>.graph.edgedump  cache: raw little-endian int32[num_e]     (data.cu:48-65)

load_graph() keeps the reference's behaviour: read .config; use a dump if present, else parse
the text file and write the dump next to it; assert indptr[num_v] == num_e (data.cu:40-45).
The one reference bug not reproduced: with only the .ptrdump cached it reads the column
indices from an already-closed FILE* (data.cu:15,21-26,54-58) -- here the text
file is re-opened and the pointer section skipped.
"""
import os

import numpy as np


def _paths(datadir, dset):
    base = os.path.join(datadir, dset)
    graph = base + ".graph"
    return base + ".config", graph, graph + ".ptrdump", graph + ".edgedump"


def read_config(datadir, dset):
    config, _, _, _ = _paths(datadir, dset)
    if not os.path.exists(config):
        raise FileNotFoundError(config)          # the reference asserts (util.cu:53, data.cu:11)
    tok = open(config).read().split()
    return int(tok[0]), int(tok[1])


def load_graph(datadir, dset, write_cache=True):
    """-> (num_v, num_e, indptr int32[num_v+1], indices int32[num_e])"""
    config, graph, ptrfile, edgefile = _paths(datadir, dset)
    num_v, num_e = read_config(datadir, dset)
    have_text = os.path.exists(graph)
    if not have_text and not (os.path.exists(ptrfile) and os.path.exists(edgefile)):
        raise FileNotFoundError(f"{graph} (or its .ptrdump/.edgedump caches)")   # util.cu:67-68
    tokens = None

    def text_tokens():
        nonlocal tokens
        if tokens is None:
            tokens = np.array(open(graph).read().split(), dtype=np.int64)
            if tokens.size < num_v + 1 + num_e:
                raise ValueError(f"{graph}: {tokens.size} integers, expected {num_v + 1 + num_e}")
        return tokens

    if os.path.exists(ptrfile):
        indptr = np.fromfile(ptrfile, dtype="<i4", count=num_v + 1)
    else:
        indptr = text_tokens()[: num_v + 1].astype(np.int32)
        if write_cache:
            indptr.astype("<i4").tofile(ptrfile)
    if indptr.size != num_v + 1 or int(indptr[num_v]) != num_e:
        raise ValueError(f"indptr[num_v]={int(indptr[-1]) if indptr.size else None} != num_e={num_e}")  # data.cu:40-45
    if os.path.exists(edgefile):
        indices = np.fromfile(edgefile, dtype="<i4", count=num_e)
    else:
        indices = text_tokens()[num_v + 1: num_v + 1 + num_e].astype(np.int32)
        if write_cache:
            indices.astype("<i4").tofile(edgefile)
    if indices.size != num_e:
        raise ValueError(f"{edgefile}: short read")
    return num_v, num_e, np.ascontiguousarray(indptr, dtype=np.int32), np.ascontiguousarray(indices, dtype=np.int32)


def write_graph(datadir, dset, indptr, indices, text=True, dumps=False):
    """Write a CSR structure in the course format (for tests and for exporting synthetic graphs)."""
    os.makedirs(datadir, exist_ok=True)
    config, graph, ptrfile, edgefile = _paths(datadir, dset)
    indptr = np.asarray(indptr, dtype=np.int32)
    indices = np.asarray(indices, dtype=np.int32)
    with open(config, "w") as f:
        f.write(f"{indptr.size - 1} {indices.size}\n")
    if text:
        with open(graph, "w") as f:
            f.write(" ".join(map(str, indptr.tolist())))
            f.write("\n")
            f.write(" ".join(map(str, indices.tolist())))
            f.write("\n")
    if dumps:
        indptr.astype("<i4").tofile(ptrfile)
        indices.astype("<i4").tofile(edgefile)
