"""scripts/summarize_prof.py condenses rocprofv3 output under gpurun_out/ into profiles/.  gpurun MERGES what each call wrote, so a directory profiled twice
holds both passes' files: only the newest pass may count (round 5: two passes summed doubled the per-step traffic of one commit)."""
import importlib.util
import os
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load():
    spec = importlib.util.spec_from_file_location("summarize_prof", os.path.join(ROOT, "scripts", "summarize_prof.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _pass(d, pid, fetch, mtime):
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, f"{pid}_counter_collection.csv")
    with open(path, "w") as f:
        f.write("Kernel_Name,Counter_Name,Counter_Value\n")
        for _ in range(3):
            f.write(f"void mi::spmm_rows_v2<4>(mi::RowsArgs),FETCH_SIZE,{fetch}\n")
        f.write("other_kernel,FETCH_SIZE,999\n")
    os.utime(path, (mtime, mtime))
    return path


def test_only_the_newest_pass_of_a_profiled_directory_counts(tmp_path):
    sp = _load()
    d = str(tmp_path / "r99_c1_fetch" / "runc")
    now = time.time()
    _pass(d, 9905, 100.0, now - 3600)          # an earlier call's pass, merged into the same directory
    new = _pass(d, 2750, 200.0, now)
    assert sp.newest([os.path.join(d, f) for f in os.listdir(d)]) == [new]
    acc = sp.counters(str(tmp_path / "r99_c1_fetch"), "mi::spmm")
    assert list(acc) == ["void mi::spmm_rows_v2<4>(mi::RowsArgs)"]
    assert acc["void mi::spmm_rows_v2<4>(mi::RowsArgs)"]["FETCH_SIZE"] == [200.0, 200.0, 200.0]      # not six values, not the old pass's
    assert sp.newest([]) == []
