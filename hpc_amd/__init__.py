"""hpc_amd -- MI355X-native CSR SpMM behind the liblaf/hpc PA4 `SpMM` operator interface.

Only what the hot path needs lives here:
  csrc/        hand-written gfx950 HIP kernels + the C ABI (include/mi_spmm.h)
  _lib.py      ctypes binding of that C ABI (fails loudly when the library is missing)
  spmm.py      host-side mirror of the reference operator interface
               (CSR, SpMM, SpMMOpt, valid -- PA4/workspace/include/spmm_base.h, util.h, valid.h)
  synth.py     synthetic CSR generators for the BASELINE.json configurations
  graph_io.py  the course's on-disk graph format (PA4/workspace/src/data.cu)
  timing.py    the reference's timing protocol (PA4/workspace/include/util.h:131-151)
  dist.py      column-sharded multi-GPU driver (RCCL all-gather of C column blocks)

The CPU oracle under oracle/ is test infrastructure and is never imported from here.
"""
from .spmm import CSR, SpMM, SpMMOpt, valid, MiSpmmError  # noqa: F401

__all__ = ["CSR", "SpMM", "SpMMOpt", "valid", "MiSpmmError"]
