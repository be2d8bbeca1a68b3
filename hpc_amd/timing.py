"""The reference's timing protocol -- PA4/workspace/include/util.h:131-151.

getCUDATime:               sync ; t0 ; f() ; sync ; t1                (:131-139)
getAverageTimeWithWarmUp:  10 un-synchronised warm-up calls, then the
                           arithmetic mean of 20 getCUDATime samples   (:141-151)
Returned in seconds, like the reference's `dbg(time)` lines
("time = 0.000382011 (double)", test/test_spmm.cu:52,61).
"""
import time


def _sync():
    import torch

    torch.cuda.synchronize()


def get_device_time(f, sync=_sync):
    sync()
    t0 = time.perf_counter()
    f()
    sync()
    return time.perf_counter() - t0


def get_average_time_with_warmup(f, n_warmup=10, n_run=20, sync=_sync, return_all=False):
    for _ in range(n_warmup):
        f()
    samples = [get_device_time(f, sync) for _ in range(n_run)]
    mean = sum(samples) / len(samples) if samples else 0.0
    return (mean, samples) if return_all else mean


def dbg_time_line(seconds, where="bench.py (TestBody)"):
    """A line the reference's plot.py regex (`time = ([\\d\\.]*) \\(double\\)`, plot.py:13-14) parses."""
    return f"[{where}] time = {seconds:.9g} (double)"


def dbg_dset_line(name, where="bench.py (argParse)"):
    return f'[{where}] dset = "{name}" (std::string)'
