"""Generate tests/golden/*.npz ON THE GPU BOX from the reference's own kernel.

    gpurun -- python tests/golden/make_golden.py gpurun_out/golden
    cp gpurun_out/golden/*.npz tests/golden/

Each fixture holds inputs (row_ptr, col_idx, vals, B) and C_ref_kernel = the output of
spmm_kernel_ref (reference PA4/workspace/src/spmm_ref.cu:3-17, compiled for gfx950 by hipcc from
the reference tree by oracle/Makefile `_ref`; the binary oracle/_ref/libspmm_ref_gfx950.so travels
to the GPU box, the reference source does not) launched with SpMMRef's geometry (spmm_ref.cu:20-30).
Fixtures are data only.  Inputs are numpy-Philox seeded (hpc_amd/synth.py) so they are reproducible.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from hpc_amd import synth  # noqa: E402
from oracle import oracle  # noqa: E402


def cases():
    out = {}
    ptr, idx, vals, B, _ = synth.config("C0")                      # BASELINE configs[0]
    out["c0_m1024_n32"] = (ptr, idx, vals, B)

    g = np.random.Generator(np.random.Philox(key=[7, 0]))
    # empty rows everywhere, incl. first and last
    deg = np.array([0, 3, 0, 0, 5, 1, 0, 7, 0], np.int64)
    ptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    idx = np.concatenate([np.sort(g.choice(9, d, replace=False)) for d in deg]).astype(np.int32)
    out["empty_rows_m9_n8"] = (ptr, idx, synth.normal_f32(idx.size, 11), synth.normal_f32(9 * 8, 12).reshape(9, 8))

    # one row longer than the student's kBatchSize = 256 (spmm_opt.cu:6), others short
    K = 700
    deg = np.array([4, 300, 0, 17, 640, 2], np.int64)
    ptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    idx = np.concatenate([np.sort(g.choice(K, d, replace=False)) for d in deg]).astype(np.int32)
    out["long_row_m6_k700_n32"] = (ptr, idx, synth.normal_f32(idx.size, 13), synth.normal_f32(K * 32, 14).reshape(K, 32))

    # duplicate column indices in a row and unsorted columns (order inside a row is unspecified
    # in the reference; both terms of a duplicate count)
    ptr = np.array([0, 5, 9, 12], np.int32)
    idx = np.array([3, 3, 0, 3, 1, 2, 0, 2, 1, 1, 1, 1], np.int32)
    out["dup_unsorted_m3_k4_n16"] = (ptr, idx, synth.normal_f32(12, 15), synth.normal_f32(4 * 16, 16).reshape(4, 16))

    # N not a multiple of 4 (dword path) and N = 256 (whole-wave rows)
    ptr, idx = synth.csr_uniform(64, 0, 20, seed=21)
    out["n5_m64"] = (ptr, idx, synth.normal_f32(idx.size, 22), synth.normal_f32(64 * 5, 23).reshape(64, 5))
    ptr, idx = synth.csr_uniform(96, 0, 40, seed=24)
    out["n256_m96"] = (ptr, idx, synth.normal_f32(idx.size, 25), synth.normal_f32(96 * 256, 26).reshape(96, 256))
    return out


def main(outdir):
    import torch

    assert torch.cuda.is_available(), "needs the MI355X box"
    assert oracle.ref_available(), "oracle/_ref/libspmm_ref_gfx950.so missing (make -C oracle _ref in the container)"
    os.makedirs(outdir, exist_ok=True)
    dev = torch.device("cuda:0")
    for name, (ptr, idx, vals, B) in cases().items():
        M, N = ptr.size - 1, B.shape[1]
        d = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (ptr, idx, vals, B)]
        C = torch.full((M, N), float("nan"), dtype=torch.float32, device=dev)
        oracle.ref_kernel_run(d[0], d[1], d[2], d[3], C, M, N)
        torch.cuda.synchronize()
        Ck = C.cpu().numpy()
        cpu = oracle.spmm_ref(ptr, idx, vals, B)
        same = np.array_equal(cpu.view(np.uint32), Ck.view(np.uint32))
        print(f"{name}: M={M} N={N} nnz={idx.size} reference-kernel vs CPU restatement bitwise equal: {same}")
        np.savez_compressed(os.path.join(outdir, name + ".npz"), row_ptr=ptr, col_idx=idx, vals=vals, B=B, C_ref_kernel=Ck)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "golden"))
