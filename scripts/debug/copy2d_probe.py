#!/usr/bin/env python3
"""Time of one hipMemcpy2DAsync device-to-device of a peer2d panel (512-byte row segments, 131 072 rows = 64 MiB) by pitch and
by the size of the allocations it runs in.  (bench.py's N = 4 rehearsal: peer2d 774 ms per step against 6.5 ms at N = 2.)

    python scripts/debug/copy2d_probe.py            # one process, src and dst both this process's allocations"""
import ctypes as C
import os
import sys


def main():
    import torch

    dev = torch.device("cuda", 0)
    hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    hip.hipMemcpy2DAsync.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p]
    M, n_loc, rows = 1 << 20, 128, 131072
    st = torch.cuda.Stream(device=dev)
    print(f"{'N_total':>8} {'pitch B':>8} {'alloc GiB':>10} {'row0':>8} {'ms':>9} {'GB/s':>8}")
    for world, alloc_gib in [(2, 1), (2, 4), (4, 2), (4, 4), (4, 5), (8, 4), (8, 5)]:
        nt = n_loc * world
        need = 4 * M * nt
        nbytes = max(need, alloc_gib << 30)
        torch.cuda.empty_cache()
        src = torch.zeros(nbytes // 4, dtype=torch.float32, device=dev)
        dst = torch.zeros(nbytes // 4, dtype=torch.float32, device=dev)
        for r0 in (0, M - rows):
            off = (r0 * nt + n_loc) * 4              # the block of rank 1
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            times = []
            for rep in range(3):
                with torch.cuda.stream(st):
                    a.record()
                    e = hip.hipMemcpy2DAsync(C.c_void_p(dst.data_ptr() + off), nt * 4, C.c_void_p(src.data_ptr() + off), nt * 4, n_loc * 4, rows, 3,
                                             C.c_void_p(st.cuda_stream))
                    b.record()
                torch.cuda.synchronize()
                assert e == 0, e
                times.append(a.elapsed_time(b))
            ms = min(times)
            print(f"{nt:>8} {nt * 4:>8} {nbytes / 2**30:>10.2f} {r0:>8} {ms:>9.3f} {rows * n_loc * 4 / ms / 1e6:>8.1f}", flush=True)
        del src, dst


if __name__ == "__main__":
    main()
