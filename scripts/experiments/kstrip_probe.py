#!/usr/bin/env python3
"""What would L2-resident column strips buy on dense-ish graphs?  A is cut into S sub-matrices by column range (strip s holds the
nonzeros with s*K/S <= col < (s+1)*K/S; columns are sorted inside a row, so a strip is a contiguous piece of every row's chain),
each sub-matrix is run as an operator of its own against the whole B, and the S runs are timed back to back.  That is the work a
strip-ordered step would do minus the carried accumulators (M x N x 4 bytes read per strip after the first): an upper bound on
the gain, measured with no kernel change.  A strip of B is K/S x N x 4 bytes; an XCD's L2 is 4 MiB.

    python scripts/experiments/kstrip_probe.py protein 32 128 256"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import torch
    from hpc_amd import CSR, SpMMOpt, synth

    name = sys.argv[1]
    widths = [int(a) for a in sys.argv[2:]] or [32]
    dev = torch.device("cuda", 0)
    key = name if name in synth.DATASET_SHAPES else name + ".dgl"
    ptr, idx = synth.csr_dataset_shaped(key)
    M = ptr.size - 1
    nnz = idx.size
    rows = np.repeat(np.arange(M, dtype=np.int64), np.diff(ptr))
    vals = synth.make_values(nnz)
    print(f"{key}: M {M}, nnz {nnz}, mean degree {nnz / M:.1f}", flush=True)
    for N in widths:
        d_B = torch.randn(M, N, device=dev) * 0.1
        d_C = torch.empty(M, N, device=dev)
        base = None
        for S in (1, 2, 4, 8, 16, 32, 64):
            strip_bytes = (M / S) * N * 4
            if S > 1 and (nnz / M / S < 4):
                break
            edges = [(M * s) // S for s in range(S + 1)]
            ops, keep = [], []
            for s in range(S):
                m = (idx >= edges[s]) & (idx < edges[s + 1])
                sub_idx = idx[m]
                sub_val = vals[m]
                cnt = np.bincount(rows[m], minlength=M)
                sub_ptr = np.zeros(M + 1, np.int32)
                np.cumsum(cnt, out=sub_ptr[1:])
                t = [torch.from_numpy(a).to(dev) for a in (sub_ptr, sub_idx, sub_val)]
                op = SpMMOpt(CSR(M, int(sub_idx.size), *t), N)
                op.preprocess(d_B, d_C)
                ops.append(op)
                keep.append(t)
            for _ in range(2):
                for op in ops:
                    op.run(d_B, d_C)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            reps = 10
            a.record()
            for _ in range(reps):
                for op in ops:
                    op.run(d_B, d_C)
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b) / reps
            base = base or ms
            carry_ms = (S - 1) * 2 * M * N * 4 / 5e12 * 1e3          # the carried accumulators at 5 TB/s, not in `ms`
            print(f"  N {N:4d}  strips {S:3d}  strip of B {strip_bytes / 2**20:7.2f} MiB  nnz/row/strip {nnz / M / S:7.1f}  "
                  f"{ms:8.3f} ms  ({base / ms:4.2f}x; carried tiles would add ~{carry_ms:.3f} ms)", flush=True)
            del ops, keep
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
