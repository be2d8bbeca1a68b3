"""The C-ABI library loads on a box without a GPU and exports every symbol include/mi_spmm.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "mi_spmm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mi_spmm_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_boundary():
    names = _declared()
    for must in ("mi_spmm_create", "mi_spmm_preprocess", "mi_spmm_run", "mi_spmm_destroy", "mi_spmm_strerror",
                 "mi_spmm_valid_float", "mi_spmm_valid_int"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from hpc_amd import _lib

    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared():
        assert hasattr(lib, name), f"{name} declared in include/mi_spmm.h but not exported"


def test_binding_table_covers_header():
    from hpc_amd import _lib

    assert sorted(_lib.SIGNATURES) == _declared()
    lib = _lib.load()
    assert lib.mi_spmm_abi_version() == 1
    assert b"gfx950" in lib.mi_spmm_build_info()
    assert b"ok" == lib.mi_spmm_strerror(0)
    assert b"CSR" in lib.mi_spmm_strerror(-4)


def test_argument_checks_without_a_gpu():
    """Pure host-side validation paths (no device call is reached)."""
    from hpc_amd import _lib

    lib = _lib.load()
    h = ctypes.c_void_p(None)
    assert lib.mi_spmm_create(ctypes.byref(h), None, None, None, 4, 4, 0, 8) == -1          # NULL row_ptr
    assert lib.mi_spmm_create(None, None, None, None, 4, 4, 0, 8) == -1
    dummy = (ctypes.c_int32 * 5)()
    assert lib.mi_spmm_create(ctypes.byref(h), dummy, None, None, -1, 4, 0, 8) == -1         # negative size
    assert lib.mi_spmm_create(ctypes.byref(h), dummy, None, None, 4, 4, 3, 8) == -1          # nnz>0, NULL idx
    assert lib.mi_spmm_create(ctypes.byref(h), dummy, None, None, 4, 4, 0, 8) == 0
    assert lib.mi_spmm_run(h, None, None, None) == -3                                         # run before preprocess
    assert lib.mi_spmm_set_option(h, b"block_threads", 100) == -1
    assert lib.mi_spmm_set_option(h, b"no_such_key", 1) == -5
    assert lib.mi_spmm_set_option(h, b"block_threads", 128) == 0
    v = ctypes.c_int64(0)
    assert lib.mi_spmm_get_option(h, b"block_threads", ctypes.byref(v)) == 0 and v.value == 128
    assert lib.mi_spmm_destroy(h) == 0
    assert lib.mi_spmm_destroy(None) == 0
    assert lib.mi_spmm_stream_create_concurrent(None, 1, None) == -1                              # no out pointer


def test_product_never_touches_the_oracle():
    """Nothing under hpc_amd/, include/ or bench's product leg may import or link oracle/."""
    for base in ("hpc_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            if "build" in dirpath.split(os.sep):
                continue
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", "Makefile")):
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    for line in text.splitlines():
                        s = line.strip()
                        if s.startswith(("#include", "import ", "from ")) or "CDLL" in s or "-l" in s:
                            assert "oracle" not in s, f"{dirpath}/{f}: {s}"


def test_no_cpu_fallback_in_the_operator():
    """The host mirror refuses host tensors instead of computing on the CPU, and the loader raises when the
    HIP library is missing: a green run can only come from the gfx950 kernels."""
    import numpy as np
    import pytest
    import torch

    from hpc_amd import CSR, SpMMOpt, _lib

    ptr = torch.zeros(3, dtype=torch.int32)
    idx = torch.zeros(0, dtype=torch.int32)
    val = torch.zeros(0, dtype=torch.float32)
    with pytest.raises(TypeError, match="device tensor"):
        CSR(2, 0, ptr, idx, val)
    # loader: a missing library is an error, not a fallback
    real = _lib.LIB_PATH
    try:
        _lib._lib = None
        _lib.LIB_PATH = real + ".missing"
        with pytest.raises(_lib.MiSpmmLibraryMissing):
            _lib.load()
    finally:
        _lib.LIB_PATH = real
        _lib._lib = None
        _lib.load()


# ---- the column-sharded step's own library (include/mi_spmm_dist.h -> hpc_amd/libmi_spmm_dist.so) ----
def _declared_dist():
    text = open(os.path.join(ROOT, "include", "mi_spmm_dist.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mi_spmm_dist_[a-z_0-9]+)\s*\(", text)))


def test_dist_library_exports_every_declared_symbol_and_binding_covers_header():
    from hpc_amd import dist

    names = _declared_dist()
    for must in ("mi_spmm_dist_create", "mi_spmm_dist_run", "mi_spmm_dist_comm_init", "mi_spmm_dist_set_peers", "mi_spmm_dist_destroy"):
        assert must in names
    lib = ctypes.CDLL(dist._DIST_PATH)
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/mi_spmm_dist.h but not exported"
    assert sorted(dist._DIST_SIGNATURES) == names
    # the operator library itself stays free of any communication dependency
    import subprocess
    from hpc_amd import _lib

    needed = subprocess.run(["readelf", "-d", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "rccl" not in needed and "rocsparse" not in needed
    assert "rccl" in subprocess.run(["readelf", "-d", dist._DIST_PATH], capture_output=True, text=True).stdout


def test_dist_argument_checks_without_a_gpu():
    from hpc_amd import _lib, dist

    lib = dist.load_dist()
    d = ctypes.c_void_p(None)
    assert lib.mi_spmm_dist_create(ctypes.byref(d), None, 4, 4, 0, 1, 1) == -1                 # NULL operator
    dummy = (ctypes.c_int32 * 5)()
    h = ctypes.c_void_p(None)
    assert _lib.load().mi_spmm_create(ctypes.byref(h), dummy, None, None, 4, 4, 0, 8) == 0
    assert lib.mi_spmm_dist_create(ctypes.byref(d), h, 4, 8, 0, 1, 1) == -3                    # operator not preprocessed
    assert lib.mi_spmm_dist_create(ctypes.byref(d), h, 4, 8, 2, 2, 1) == -1                    # rank >= world
    assert lib.mi_spmm_dist_destroy(None) == 0
    assert b"NCCL" in lib.mi_spmm_dist_strerror(-1003) or b"internal" in lib.mi_spmm_dist_strerror(-1003)
    assert b"invalid argument" in lib.mi_spmm_dist_strerror(-1)
    _lib.load().mi_spmm_destroy(h)


def test_ipc_exportable_sizes_rule():
    """Allocation sizes HIP IPC can open (include/mi_spmm_dist.h): bit 31 of the size must be clear, sizes that have it set go up
    to the next multiple of 4 GiB (hipIpcOpenMemHandle hangs on 2, 3 and 6 GiB and opens 1, 1.5, 4 and 5 GiB: profiles/r04_ipc_open_sizes.txt)."""
    from hpc_amd import dist

    G = 1 << 30
    f = dist.ipc_exportable_bytes
    assert [f(0), f(1), f(G), f(2 * G - 1), f(3 * G // 2)] == [0, 1, G, 2 * G - 1, 3 * G // 2]
    assert [f(2 * G), f(2 * G + 1), f(3 * G), f(4 * G - 1)] == [4 * G] * 4
    assert [f(4 * G), f(5 * G), f(6 * G - 1)] == [4 * G, 5 * G, 6 * G - 1]
    assert [f(6 * G), f(7 * G + 5), f(8 * G)] == [8 * G, 8 * G, 8 * G]
    assert f(-1) == -1
    # what bench.py allocates as C_full for C1 at 128 columns per GPU: M = 2^20 rows x (128 x G) columns x 4 bytes
    assert [f(4 * (1 << 20) * 128 * g) // G for g in (1, 2, 3, 4, 6, 8)] == [0, 1, 1, 4, 4, 4]
    assert [f(4 * (1 << 20) * 128 * g) for g in (2, 4, 8)] == [G, 4 * G, 4 * G]


def test_profiles_text_files_are_sane():
    """Evidence files under profiles/ are small and not a paragraph repeated thousands of times (round 2 lost one that way:
    an append script iterated over the characters of the old text)."""
    import collections
    import glob

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for path in glob.glob(os.path.join(root, "profiles", "*.txt")) + glob.glob(os.path.join(root, "profiles", "*.md")):
        assert os.path.getsize(path) < 400_000, f"{path}: {os.path.getsize(path)} bytes"
        lines = [l.strip() for l in open(path, errors="replace") if len(l.strip()) > 40]
        if lines:
            line, n = collections.Counter(lines).most_common(1)[0]
            assert n <= 20, f"{path}: a line repeated {n} times: {line[:60]!r}"


def test_generated_hub_chain_assembly_is_in_sync_with_its_generator():
    """hpc_amd/csrc/hub_chain_asm.inc (the hub kernel's chain loop) is committed generator output: an edit to either side
    without the other would ship an assembly loop nobody reviewed."""
    import importlib.util

    path = os.path.join(ROOT, "hpc_amd", "csrc", "gen_hub_chain.py")
    spec = importlib.util.spec_from_file_location("gen_hub_chain", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    with open(os.path.join(ROOT, "hpc_amd", "csrc", "hub_chain_asm.inc")) as f:
        assert f.read() == mod.render()
    # the ring the text was generated for is the ring the kernel declares (HubCfg in spmm_kernels.hpp)
    src = open(os.path.join(ROOT, "hpc_amd", "csrc", "spmm_kernels.hpp")).read()
    for decl in ("static constexpr int ST = 64;", "static constexpr int L = 3;", "static constexpr int CS = ST + 4;", "static constexpr int NB = 2 * L;"):
        assert decl in src, decl
    assert (mod.ST, mod.LOADERS, mod.NB, mod.CS) == (64, 3, 6, 68)


def test_kernel_argument_structs_are_value_initialised():
    """Every kernel-argument struct (`*Args`) handed to a launch is declared `T x{};`.  Round 3's hub_micro experiment
    declared `HubArgs a;`, set every field but `a.po` (added to the struct later) and the kernel's epilogue stored through
    the stack garbage in it: a GPU memory fault under the profiler's preload.  A field added to a struct tomorrow must
    start as zero everywhere the struct is built -- product, experiments and native tests alike."""
    import glob

    pats = [os.path.join(ROOT, "hpc_amd", "csrc", "*.hip"), os.path.join(ROOT, "hpc_amd", "csrc", "*.cpp"),
            os.path.join(ROOT, "scripts", "experiments", "make_*.py"), os.path.join(ROOT, "scripts", "experiments", "*.hip"),
            os.path.join(ROOT, "tests", "native", "*.cpp"), os.path.join(ROOT, "oracle", "*.hip")]
    bare = re.compile(r"\b(\w*Args)\s+(\w+)\s*;")        # `HubArgs a;` -- a declaration with no initialiser
    seen = 0
    for pat in pats:
        for path in glob.glob(pat):
            for no, line in enumerate(open(path), 1):
                if line.lstrip().startswith(("//", "#", "*")):
                    continue
                m = bare.search(line)
                assert not m, f"{path}:{no}: `{m.group(0)}` leaves fields uninitialised: write `{m.group(1)} {m.group(2)}{{}};`"
                seen += len(re.findall(r"\b\w*Args\s+\w+\{\};", line))
    assert seen >= 7, "expected the product's five launch sites and the two experiment drivers"
    # the generated experiment sources are the generators' output as committed
    for gen, out in (("make_hub_micro.py", "hub_micro.hip"), ("make_block_micro.py", "block_micro.hip")):
        src = open(os.path.join(ROOT, "scripts", "experiments", out)).read()
        assert re.search(r"\b\w*Args \w+\{\};", src), f"{out}: regenerate it with scripts/experiments/{gen}"


def test_traffic_entries_carry_a_kernel_source_hash_and_bench_flags_stale_ones():
    """bench.py quotes PMC traffic measured by an earlier profile run (profiles/traffic_latest.json).  Round 3's C4 entry
    described a kernel that had been replaced afterwards and nothing in the line said so: every entry now carries the sha256
    of the kernel sources it was measured on, and the line says `traffic_stale` when the tree's differ."""
    import json
    import sys

    sys.path.insert(0, ROOT)
    import bench
    from hpc_amd._lib import KERNEL_SOURCES, kernel_sources_sha256

    here = kernel_sources_sha256()
    # (round 5: the plan sources too -- they decide strip counts, thresholds and launch sets, i.e. the traffic a profile measures)
    assert here and len(here) == 64 and set(KERNEL_SOURCES) == {"spmm_kernels.hpp", "mi_spmm.hip", "hub_chain_asm.inc", "plan.hpp", "plan_types.hpp",
                                                               "preprocess_gpu.hip"}
    csrc = os.path.join(ROOT, "hpc_amd", "csrc")
    on_disk = {f for f in os.listdir(csrc) if f.endswith((".hpp", ".hip", ".inc"))}
    assert set(KERNEL_SOURCES) == on_disk, on_disk ^ set(KERNEL_SOURCES)       # a new device source must join the hash
    assert bench.traffic_staleness({"kernel_sources_sha256": here})["traffic_stale"] is False
    assert bench.traffic_staleness({"kernel_sources_sha256": "0" * 64})["traffic_stale"] is True
    assert bench.traffic_staleness({})["traffic_stale"] is True                       # entries from before round 4
    st = bench.traffic_staleness({"kernel_sources_sha256": "0" * 64})
    assert st["kernel_sources_sha256"] == {"measured": "0" * 64, "this_run": here}
    tl = json.load(open(os.path.join(ROOT, "profiles", "traffic_latest.json")))
    for key, ent in tl.items():
        for k in ("source", "commit", "hbm_bytes_per_launch", "N"):
            assert k in ent, (key, k)
        assert os.path.exists(os.path.join(ROOT, ent["source"])), ent["source"]
        # an entry names a commit only if that commit's sources are what was profiled; a dirty tree is identified by the hash alone (scripts/summarize_prof.py)
        assert ent["commit"].startswith("dirty-tree") or re.fullmatch(r"[0-9a-f]{7,40}", ent["commit"]), ent["commit"]
        assert "kernel_sources_sha256" in ent
