"""Synthetic CSR inputs for the BASELINE.json configurations.

The reference reads 13 course graphs that are not in its repository
(PA4/workspace/script/run_all.sh:3,10) and fills A's values, B and C with
cuRAND N(0, 0.1) (include/data.h:31, seed 123 test/main.cpp:20).  Here:

  * structure: int32 row_ptr[M+1], col_idx[nnz] sorted ascending and distinct
    inside a row, 0-based, square (K = M) like the reference's adjacency;
  * values: fp32 N(0, 0.1) from numpy's counter-based Philox generator keyed by
    (seed, stream) -- the distribution of the reference, not cuRAND's bit stream
    (XORWOW is not reproducible without cuRAND);
  * seeds: structure 123 (echoing main.cpp:20), vals 124, B 125.

Everything is generated on the host with numpy and is a pure function of its
arguments, so the CPU container, the GPU box and every rank build identical
inputs without shipping files.
"""
import numpy as np

SEED_STRUCT = 123
SEED_VALS = 124
SEED_B = 125


def _rng(seed, stream=0):
    return np.random.Generator(np.random.Philox(key=[int(seed), int(stream)]))


def normal_f32(n, seed, stream=0, mean=0.0, std=0.1, chunk=1 << 24):
    """fp32 N(mean, std) of length n; chunked so peak host memory stays ~2x the output."""
    out = np.empty(int(n), dtype=np.float32)
    g = _rng(seed, stream)
    for s in range(0, int(n), chunk):
        e = min(int(n), s + chunk)
        out[s:e] = g.standard_normal(e - s, dtype=np.float32) * np.float32(std) + np.float32(mean)
    return out


def _finish_rows(M, K, deg, g):
    """Random distinct sorted columns per row for the given degree vector."""
    deg = np.minimum(deg.astype(np.int64), K)
    row_ptr = np.zeros(M + 1, dtype=np.int64)
    np.cumsum(deg, out=row_ptr[1:])
    nnz = int(row_ptr[-1])
    rows = np.repeat(np.arange(M, dtype=np.int64), deg)
    cols = g.integers(0, K, size=nnz, dtype=np.int64)
    key = rows * np.int64(K) + cols
    key.sort()
    # drop duplicate (row, col) pairs: columns stay distinct inside a row
    keep = np.ones(nnz, dtype=bool)
    keep[1:] = key[1:] != key[:-1]
    key = key[keep]
    rows = key // np.int64(K)
    cols = (key - rows * np.int64(K)).astype(np.int32)
    counts = np.bincount(rows, minlength=M).astype(np.int64)
    row_ptr = np.zeros(M + 1, dtype=np.int64)
    np.cumsum(counts, out=row_ptr[1:])
    assert row_ptr[-1] <= np.iinfo(np.int32).max
    return row_ptr.astype(np.int32), cols


def csr_uniform(M, deg_lo, deg_hi, K=None, seed=SEED_STRUCT):
    """Row degree ~ Uniform{deg_lo..deg_hi}, uniform random columns.
    C0: M=1024, 0..32 (mean 16, empty rows included).  C1: M=2^20, 16..48 (mean 32)."""
    K = M if K is None else K
    g = _rng(seed, 0)
    deg = g.integers(deg_lo, deg_hi + 1, size=M, dtype=np.int64)
    return _finish_rows(M, K, deg, g)


def csr_powerlaw(M, mean_deg=32.0, max_deg=4096, alpha=1.5, K=None, seed=SEED_STRUCT, force_max=False):
    """C2: degree_i = min(max_deg, floor(d_min * u^(-1/alpha))), d_min solved (bisection on
    this very sample) so that the mean degree is mean_deg; rows are NOT sorted by degree.
    force_max: give the longest row exactly max_deg nonzeros (dataset-shaped stand-ins whose
    logged max degree a finite sample would not reach)."""
    K = M if K is None else K
    g = _rng(seed, 1)
    u = g.random(M)
    w = u ** (-1.0 / alpha)
    lo, hi = 0.01, float(mean_deg)
    for _ in range(60):
        mid = 0.5 * (lo + hi)
        m = np.minimum(max_deg, np.floor(mid * w)).mean()
        if m < mean_deg:
            lo = mid
        else:
            hi = mid
    deg = np.minimum(max_deg, np.floor(hi * w)).astype(np.int64)
    if force_max and M > 0:
        deg[int(np.argmax(deg))] = min(max_deg, K)
    return _finish_rows(M, K, deg, g)


# The reference's 13 course graphs (W/script/run_all.sh:3) are not in its repository; their longest rows are (W/phase_2.log).
# name: (rows, nonzeros, longest row).  Rows / nonzeros are the public OGB / DGL / CogDL statistics of the datasets of those
# names (approximate; stated here, not taken from the reference).
DATASET_SHAPES = {
    "arxiv": (169_343, 1_166_243, 13_155), "collab": (235_868, 2_358_104, 671), "citation": (2_927_963, 30_387_995, 1_738),
    "ddi": (4_267, 2_135_822, 2_234), "protein": (132_534, 79_122_504, 7_750), "ppa": (576_289, 42_463_862, 3_241),
    "reddit.dgl": (232_965, 114_615_892, 21_657), "products": (2_449_029, 123_718_280, 17_481),
    "youtube": (1_138_499, 5_980_886, 28_754), "amazon_cogdl": (1_569_960, 264_339_468, 75_134),
    "yelp": (716_847, 13_954_819, 4_886), "wikikg2": (2_500_604, 16_109_182, 911), "am": (881_680, 5_668_682, 154_828),
}


def csr_dataset_shaped(name):
    """A stand-in with the SHAPE of one of the reference's datasets: that many rows and nonzeros, a power-law degree profile
    whose longest row is the logged one, uniformly random columns (no community structure: harsher on caches than the real
    graph).  scripts/report_table.py, scripts/hub_bench.py, `bench.py --config am` and the full-size tests use this one."""
    M, nnz_target, max_deg = DATASET_SHAPES[name]
    return csr_powerlaw(M, nnz_target / M, min(max_deg, M), seed=sum(map(ord, name)) % 1000 + 1, force_max=True)


def csr_block_dense(M, rows_per_block=16, run_lo=64, run_hi=128, max_runs=2, K=None, seed=SEED_STRUCT):
    """C4: rows in blocks of `rows_per_block` share 1..max_runs aligned contiguous column runs
    of 64..128 columns (run start and length multiples of 64/..: aligned to 64); every row of a
    block holds every column of the block's runs (>= 64 contiguous nonzeros per row)."""
    K = M if K is None else K
    g = _rng(seed, 2)
    nb = (M + rows_per_block - 1) // rows_per_block
    n_runs = g.integers(1, max_runs + 1, size=nb)
    ptr = np.zeros(M + 1, dtype=np.int64)
    cols_blocks = []
    lens = np.zeros(nb, dtype=np.int64)
    align = 64
    for b in range(nb):
        runs = []
        for _ in range(int(n_runs[b])):
            length = int(g.integers(run_lo // align, run_hi // align + 1)) * align
            start = int(g.integers(0, max(1, (K - length) // align + 1))) * align
            runs.append((start, length))
        runs.sort()
        # merge overlaps so columns stay distinct and ascending
        merged = []
        for s, l in runs:
            if merged and s <= merged[-1][0] + merged[-1][1]:
                ps, pl = merged[-1]
                merged[-1] = (ps, max(pl, s + l - ps))
            else:
                merged.append((s, l))
        c = np.concatenate([np.arange(s, min(K, s + l), dtype=np.int32) for s, l in merged])
        cols_blocks.append(c)
        lens[b] = c.size
    deg = np.repeat(lens, rows_per_block)[:M]
    np.cumsum(deg, out=ptr[1:])
    col_idx = np.empty(int(ptr[-1]), dtype=np.int32)
    for b in range(nb):
        r0 = b * rows_per_block
        r1 = min(M, r0 + rows_per_block)
        c = cols_blocks[b]
        col_idx[ptr[r0]:ptr[r1]] = np.tile(c, r1 - r0)
    assert ptr[-1] <= np.iinfo(np.int32).max
    return ptr.astype(np.int32), col_idx


def _block_dense_fast_params(M, rows_per_block, K, seed):
    """The per-block draws behind csr_block_dense_fast (cheap: one entry per 16-row block)."""
    K = M if K is None else K
    g = _rng(seed, 3)
    nb = (M + rows_per_block - 1) // rows_per_block
    align = 64
    slots = K // align
    n_runs = g.integers(1, 3, size=nb)
    len1 = g.integers(1, 3, size=nb) * align
    len2 = g.integers(1, 3, size=nb) * align
    s1 = g.integers(0, slots - 4, size=nb) * align
    gap = g.integers(0, slots // 2, size=nb) * align
    s2 = np.minimum(s1 + len1 + gap, (slots - 2) * align)   # second run never overlaps the first
    s2 = np.maximum(s2, s1 + len1)
    len2 = np.where(n_runs == 2, len2, 0)
    return len1, len2, s1, s2


def csr_block_dense_fast(M, rows_per_block=16, K=None, seed=SEED_STRUCT):
    """Vectorised C4 generator for M = 2^20: one or two aligned runs of 64 or 128 columns per
    16-row block (same family as csr_block_dense, no Python loop over blocks)."""
    len1, len2, s1, s2 = _block_dense_fast_params(M, rows_per_block, K, seed)
    blk_len = (len1 + len2).astype(np.int64)
    deg = np.repeat(blk_len, rows_per_block)[:M]
    ptr = np.zeros(M + 1, dtype=np.int64)
    np.cumsum(deg, out=ptr[1:])
    nnz = int(ptr[-1])
    assert nnz <= np.iinfo(np.int32).max
    rows = np.repeat(np.arange(M, dtype=np.int64), deg)
    pos = np.arange(nnz, dtype=np.int64) - ptr[:-1][rows]      # position inside the row
    b = rows // rows_per_block
    in_first = pos < len1[b]
    col = np.where(in_first, s1[b] + pos, s2[b] + (pos - len1[b]))
    return ptr.astype(np.int32), col.astype(np.int32)


def csr_block_dense_fast_device(M, device, rows_per_block=16, K=None, seed=SEED_STRUCT):
    """csr_block_dense_fast with the expansion to 151 M nonzeros done on the device (torch): the same per-block draws, the
    same row_ptr / col_idx (test_synth_device_generators_match), seconds instead of half a minute of host time.  Returns
    device int32 tensors (row_ptr, col_idx)."""
    import torch

    len1, len2, s1, s2 = (torch.from_numpy(np.ascontiguousarray(a, dtype=np.int64)).to(device)
                          for a in _block_dense_fast_params(M, rows_per_block, K, seed))
    deg = torch.repeat_interleave(len1 + len2, rows_per_block)[:M]
    ptr = torch.zeros(M + 1, dtype=torch.int64, device=device)
    torch.cumsum(deg, 0, out=ptr[1:])
    nnz = int(ptr[-1].item())
    assert nnz <= np.iinfo(np.int32).max
    rows = torch.repeat_interleave(torch.arange(M, dtype=torch.int32, device=device), deg)
    pos = torch.arange(nnz, dtype=torch.int32, device=device) - ptr[:-1].to(torch.int32)[rows.long()]
    b = (rows // rows_per_block).long()
    del rows
    l1 = len1.to(torch.int32)[b]
    col = torch.where(pos < l1, s1.to(torch.int32)[b] + pos, s2.to(torch.int32)[b] + (pos - l1))
    return ptr.to(torch.int32), col


def csr_long_rows_device(M, device, lo=300, hi=700, K=None, seed=SEED_STRUCT):
    """Long rows over few columns (protein- / reddit- / ddi-like: hundreds of nonzeros in every row), built on the device in well
    under a second: row lengths ~ Uniform{lo..hi} (host draw), row r's j-th column = floor((j + u) * K / len_r) with u ~ U[0, 1)
    drawn on the device -- ascending inside a row (equal neighbours are possible and allowed), spread over all K columns.
    `bench.py`'s `also` entry for the column strips of the segments.  Returns device int32 tensors (row_ptr, col_idx)."""
    import torch

    K = M if K is None else K
    deg = torch.from_numpy(_rng(seed, 7).integers(lo, hi + 1, size=M, dtype=np.int64)).to(device)
    ptr = torch.zeros(M + 1, dtype=torch.int64, device=device)
    torch.cumsum(deg, 0, out=ptr[1:])
    nnz = int(ptr[-1].item())
    assert nnz <= np.iinfo(np.int32).max
    rows = torch.repeat_interleave(torch.arange(M, dtype=torch.int64, device=device), deg)
    pos = torch.arange(nnz, dtype=torch.int64, device=device) - ptr[:-1][rows]
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))
    u = torch.rand(nnz, generator=gen, device=device, dtype=torch.float64)
    col = ((pos.to(torch.float64) + u) * (float(K) / deg[rows].to(torch.float64))).floor().clamp_(0, K - 1)
    return ptr.to(torch.int32), col.to(torch.int32)


def csr_rmat(scale, edge_factor=32, a=0.57, b=0.19, c=0.19, seed=SEED_STRUCT):
    """R-MAT / Kronecker graph (Chakrabarti, Zhan, Faloutsos 2004; Graph500 parameters): M = 2^scale rows,
    about edge_factor * M edges before duplicate removal.  Hubs with 10^4..10^5 nonzeros next to empty rows --
    the shape of the course's real graphs (max degree up to 154 828, W/phase_2.log), harsher than C2."""
    M = 1 << scale
    E = M * edge_factor
    g = _rng(seed, 4)
    rows = np.zeros(E, dtype=np.int64)
    cols = np.zeros(E, dtype=np.int64)
    for _ in range(scale):
        r = g.random(E, dtype=np.float32)
        rows = (rows << 1) | (r >= a + b)
        cols = (cols << 1) | (((r >= a) & (r < a + b)) | (r >= a + b + c))
    key = rows * np.int64(M) + cols
    key = np.unique(key)
    rows = key // np.int64(M)
    cols = (key - rows * np.int64(M)).astype(np.int32)
    ptr = np.zeros(M + 1, dtype=np.int64)
    np.cumsum(np.bincount(rows, minlength=M), out=ptr[1:])
    assert ptr[-1] <= np.iinfo(np.int32).max
    return ptr.astype(np.int32), cols


def csr_banded(M, deg_lo=16, deg_hi=48, width=2048, seed=SEED_STRUCT):
    """Locality-rich structure: row r's columns are drawn within +-width of r (community / mesh-like
    adjacency): neighbouring rows share B rows, so L2 / Infinity Cache reuse exists to be had."""
    g = _rng(seed, 5)
    deg = g.integers(deg_lo, deg_hi + 1, size=M, dtype=np.int64)
    rows = np.repeat(np.arange(M, dtype=np.int64), deg)
    off = g.integers(-width, width + 1, size=rows.size, dtype=np.int64)
    cols = np.clip(rows + off, 0, M - 1)
    key = np.unique(rows * np.int64(M) + cols)
    rows = key // np.int64(M)
    cols = (key - rows * np.int64(M)).astype(np.int32)
    ptr = np.zeros(M + 1, dtype=np.int64)
    np.cumsum(np.bincount(rows, minlength=M), out=ptr[1:])
    return ptr.astype(np.int32), cols


# ---- structured graphs (round 5): nothing below draws its columns uniformly at random ---------------------------------
# The reference's 13 course graphs are community-structured, symmetric adjacency matrices with correlated degrees: a hub ROW is also a hub COLUMN
# (its B row is gathered by thousands of rows), neighbours share neighbours, and a BFS / RCM / partitioner ordering puts a community's vertices next to
# each other.  The dataset-shaped stand-ins above have none of that.  These generators do, in well under a second on the device (torch), so that the
# auto rules (tile width, column strips, hub threshold, side streams, hub slices) meet such inputs: scripts/regret.py.


def _powerlaw_weights(M, mean_deg, max_deg, alpha, seed, stream):
    """Expected degrees with csr_powerlaw's profile (host, M numbers): min(max_deg, d_min * u^(-1/alpha)), mean = mean_deg; the longest = max_deg."""
    u = _rng(seed, stream).random(M)
    w = u ** (-1.0 / alpha)
    lo, hi = 1e-3, float(mean_deg)
    for _ in range(60):
        mid = 0.5 * (lo + hi)
        if np.minimum(max_deg, mid * w).mean() < mean_deg:
            lo = mid
        else:
            hi = mid
    w = np.minimum(float(max_deg), hi * w)
    if M > 0:
        w[int(np.argmax(w))] = float(min(max_deg, M))
    return w


def _csr_from_pairs_device(rows, cols, M, K, sort_cols=True, seed=0):
    """(row, col) pairs on the device -> int32 (row_ptr, col_idx): duplicates dropped; columns ascending inside a row, or -- sort_cols = False --
    in a random order (the reference leaves the order unspecified, util.h:120-129: switches the column strips off, `segments_unsorted`)."""
    import torch

    key = torch.unique(rows.to(torch.int64) * int(K) + cols.to(torch.int64))      # sorted
    rows = torch.div(key, int(K), rounding_mode="floor")
    cols = (key - rows * int(K)).to(torch.int32)
    del key
    ptr = torch.zeros(M + 1, dtype=torch.int64, device=rows.device)
    torch.cumsum(torch.bincount(rows, minlength=M), 0, out=ptr[1:])
    assert int(ptr[-1].item()) <= np.iinfo(np.int32).max
    if not sort_cols:
        gen = torch.Generator(device=rows.device)
        gen.manual_seed(int(seed) + 77)
        r = torch.rand(cols.numel(), generator=gen, device=rows.device, dtype=torch.float64)
        order = torch.argsort(rows.to(torch.float64) + r * 0.999)               # rows stay grouped, the order inside a row is random
        cols = cols[order]
    return ptr.to(torch.int32), cols


def csr_dcsbm_device(M, nnz_target, max_deg, device, alpha=1.5, mean_comm=2048, p_in=0.8, order="community", symmetric=True, sort_cols=True,
                     seed=SEED_STRUCT):
    """Degree-corrected stochastic block model: expected degrees with a power-law profile (longest row ~ max_deg), communities of random sizes
    (mean `mean_comm`), every edge's second endpoint drawn -- in proportion to the expected degrees -- inside the first endpoint's community with
    probability p_in and from the whole graph otherwise; symmetric = True adds the transposed edges (A = A^T: hub rows are hub columns).
    order: "community" -- a community's vertices have consecutive ids (what a BFS / RCM / partitioner ordering of such a graph yields);
           "shuffled"  -- ids permuted at random (no locality; the degree correlation stays);
           "degree"    -- vertices sorted by expected degree, hubs first (crawl order of many public datasets; un-permuted R-MAT looks like this).
    alpha = 0: all expected degrees equal (a plain SBM: dense diagonal blocks + sparse off-diagonal).
    Returns device int32 (row_ptr, col_idx)."""
    import torch

    dev = torch.device(device)
    mean_deg = nnz_target / M
    if alpha > 0:
        w = _powerlaw_weights(M, mean_deg, max_deg, alpha, seed, 11)
    else:
        w = np.full(M, mean_deg)
    g = _rng(seed, 12)
    n_comm = max(1, int(round(M / mean_comm)))
    cuts = np.sort(g.choice(np.arange(1, M), size=n_comm - 1, replace=False)) if n_comm > 1 else np.empty(0, dtype=np.int64)
    comm_beg = np.concatenate([[0], cuts]).astype(np.int64)
    comm_end = np.concatenate([cuts, [M]]).astype(np.int64)
    w_d = torch.from_numpy(w).to(dev)
    sizes = torch.from_numpy(comm_end - comm_beg).to(dev)
    comm_of = torch.repeat_interleave(torch.arange(n_comm, device=dev), sizes)
    # every vertex splits its expected degree into an intra-community share and a global share.  A vertex whose expected degree is a large part of its
    # community cannot keep p_in of its edges inside (they would be duplicates): its share shrinks -- hubs connect everywhere, like the real ones.
    q = p_in * torch.clamp(0.25 * sizes[comm_of].to(torch.float64) / w_d, max=1.0)
    wa, wb = w_d * q, w_d * (1.0 - q)
    del q
    E = int(nnz_target // 2) if symmetric else int(nnz_target)
    E_in = int(round(E * float(wa.sum().item()) / float(w_d.sum().item())))
    gen = torch.Generator(device=dev)
    gen.manual_seed(int(seed))

    def draw(cw, lo, hi, n):                            # n vertices in proportion to the weights behind the cumulative sums cw, inside [lo, hi)
        u = torch.rand(n, generator=gen, device=dev, dtype=torch.float64)
        return torch.searchsorted(cw, lo + u * (hi - lo)).clamp_(max=M - 1)

    zero = torch.zeros((), dtype=torch.float64, device=dev)
    # intra-community edges: first endpoint in proportion to wa (picks the community too), second one in proportion to wa inside that community
    ca = torch.cumsum(wa, 0)
    ca0 = torch.cat([zero.reshape(1), ca])              # ca0[v] = weight in front of vertex v
    i_in = draw(ca, zero, ca[-1], E_in)
    c = comm_of[i_in]
    j_in = draw(ca, ca0[torch.from_numpy(comm_beg).to(dev)[c]], ca0[torch.from_numpy(comm_end).to(dev)[c]], E_in)
    del c, ca, ca0
    # global edges: both endpoints in proportion to wb
    cb_ = torch.cumsum(wb, 0)
    i_gl = draw(cb_, zero, cb_[-1], E - E_in)
    j_gl = draw(cb_, zero, cb_[-1], E - E_in)
    del cb_
    i, j = torch.cat([i_in, i_gl]), torch.cat([j_in, j_gl])
    del i_in, j_in, i_gl, j_gl
    if order == "shuffled":
        perm = torch.from_numpy(g.permutation(M)).to(dev)
        i, j = perm[i], perm[j]
    elif order == "degree":
        rank = torch.empty(M, dtype=torch.int64, device=dev)
        rank[torch.argsort(w_d, descending=True, stable=True)] = torch.arange(M, device=dev)
        i, j = rank[i], rank[j]
    elif order != "community":
        raise ValueError(order)
    if symmetric:
        i, j = torch.cat([i, j]), torch.cat([j, i])
    return _csr_from_pairs_device(i, j, M, M, sort_cols=sort_cols, seed=seed)


def csr_dataset_structured_device(name, device, order="community", sort_cols=True, p_in=0.8):
    """The SHAPE of one of the reference's datasets (rows, nonzeros, longest row: DATASET_SHAPES) with the STRUCTURE csr_dcsbm_device gives it."""
    M, nnz_target, max_deg = DATASET_SHAPES[name]
    mean_comm = 256 if M < 10_000 else 2048
    return csr_dcsbm_device(M, nnz_target, min(max_deg, M), device, mean_comm=mean_comm, p_in=p_in, order=order, sort_cols=sort_cols,
                            seed=sum(map(ord, name)) % 1000 + 1)


def csr_rmat_device(scale, device, edge_factor=32, a=0.57, b=0.19, c=0.19, seed=SEED_STRUCT, sort_cols=True):
    """csr_rmat's model drawn on the device (its own random stream: not the same edges), vertex ids NOT permuted: vertex 0 is the largest hub, ids with few
    set bits are hubs, and hub rows are hub columns -- the structure a Graph500 generator has before its final relabelling."""
    import torch

    dev = torch.device(device)
    M = 1 << scale
    E = M * edge_factor
    gen = torch.Generator(device=dev)
    gen.manual_seed(int(seed))
    rows = torch.zeros(E, dtype=torch.int64, device=dev)
    cols = torch.zeros(E, dtype=torch.int64, device=dev)
    for _ in range(scale):
        r = torch.rand(E, generator=gen, device=dev)
        rows = (rows << 1) | (r >= a + b)
        cols = (cols << 1) | (((r >= a) & (r < a + b)) | (r >= a + b + c))
    return _csr_from_pairs_device(rows, cols, M, M, sort_cols=sort_cols, seed=seed)


def csr_banded_long_rows_device(M, device, width=2048, lo=300, hi=700, seed=7):
    """Every column within +-width of the row's own index (mesh / banded), lo..hi nonzeros per row: long rows whose columns sit inside one or two column
    strips -- the structure round 4's strip rule excluded by its locality gate (scripts/regret.py: strips and row-ordered segments halve its steps)."""
    import torch

    dev = torch.device(device)
    deg = torch.from_numpy(_rng(seed, 7).integers(lo, hi + 1, size=M)).to(dev)
    rows = torch.repeat_interleave(torch.arange(M, device=dev), deg)
    gen = torch.Generator(device=dev)
    gen.manual_seed(int(seed))
    off = torch.randint(-width, width + 1, (rows.numel(),), generator=gen, device=dev)
    return _csr_from_pairs_device(rows, (rows + off).clamp_(0, M - 1), M, M)


def csr_reorder_rcm(ptr, idx):
    """A true reverse Cuthill-McKee ordering (scipy, host) of a graph given as numpy CSR: rows and columns relabelled by the same permutation, columns
    re-sorted.  For the small dataset shapes (seconds up to a few million nonzeros); the larger ones use order = "community" instead."""
    import scipy.sparse as sp
    from scipy.sparse.csgraph import reverse_cuthill_mckee

    M = ptr.size - 1
    A = sp.csr_matrix((np.ones(idx.size, dtype=np.int8), idx, ptr), shape=(M, M))
    perm = reverse_cuthill_mckee(A, symmetric_mode=False)
    A = A[perm][:, perm].tocsr()
    A.sort_indices()
    return A.indptr.astype(np.int32), A.indices.astype(np.int32)


def make_values(nnz, seed=SEED_VALS):
    return normal_f32(nnz, seed, 0)


def make_dense(K, N, seed=SEED_B):
    return normal_f32(int(K) * int(N), seed, 0).reshape(int(K), int(N))


def config(name, N=None, M=None):
    """(row_ptr, col_idx, vals, B, meta) for BASELINE.json configs C0..C4 (optionally down-sized)."""
    name = name.upper()
    if name == "C0":
        M = 1024 if M is None else M
        N = 32 if N is None else N
        ptr, idx = csr_uniform(M, 0, 32)
    elif name == "C1":
        M = (1 << 20) if M is None else M
        N = 128 if N is None else N
        ptr, idx = csr_uniform(M, 16, 48)
    elif name == "C2":
        M = (1 << 20) if M is None else M
        N = 128 if N is None else N
        ptr, idx = csr_powerlaw(M, 32.0, 4096)
    elif name == "C4":
        M = (1 << 20) if M is None else M
        N = 256 if N is None else N
        ptr, idx = csr_block_dense_fast(M)
    else:
        raise ValueError(name)
    vals = make_values(idx.size)
    B = make_dense(M, N)
    deg = np.diff(ptr)
    meta = {
        "config": name, "M": int(M), "K": int(M), "N": int(N), "nnz": int(idx.size),
        "deg_mean": float(deg.mean()) if M else 0.0, "deg_max": int(deg.max()) if M else 0,
        "deg_p50": float(np.percentile(deg, 50)) if M else 0.0, "deg_p99": float(np.percentile(deg, 99)) if M else 0.0,
    }
    return ptr, idx, vals, B, meta


def bytes_model(M, K, N, nnz):
    """algorithmic (gather-model) bytes, compulsory lower bound, flops."""
    alg = 8 * nnz + 4 * (M + 1) + 4 * N * nnz + 4 * M * N
    low = 8 * nnz + 4 * (M + 1) + 4 * K * N + 4 * M * N
    return {"bytes_alg": int(alg), "bytes_min": int(low), "flops": int(2 * nnz * N)}
