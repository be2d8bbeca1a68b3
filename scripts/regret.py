#!/usr/bin/env python3
"""Regret of every auto rule on STRUCTURED graphs: `auto` against forced settings, per graph x kLen, in one process.

    python scripts/regret.py [--only NAME] [--lens 32,128,256] [--quick] > gpurun_out/regret.jsonl      (GPU box)
    python scripts/regret.py --summarize gpurun_out/regret.jsonl > profiles/r05_regret.md            (anywhere)

The rules under test (hpc_amd/csrc/plan.hpp, mi_spmm.hip): `tile_cols` (resolve_tile_cols), `col_strips` (resolve_col_strips + the hub fold),
`long_row_threshold` (resolve_hub_threshold), `medium_row_threshold`, `segment_overlap` / `hub_overlap` (ensure_side_streams), `hub_slice`.
All of them are scheduling only, so every forced setting must give the SAME BITS: each C is compared with auto's C (count_bitdiff) and auto's C with the
reference's own spmm_kernel_ref compiled for this GPU (oracle/_ref) -- a setting that differs is reported as an error, never timed.

Search: small graphs (< 8 M nonzeros) get the full grid  tile_cols {0,64,128,256} x col_strips {1,auto,S/2,2S} x long_row_threshold {256 .. 8192, none}
x segment_overlap {0,1} x hub_slice {16,32}  (dimensions that cannot change the launch set are collapsed: no hub at a threshold -> one hub_slice ...).
Larger graphs get the cross  long_row_threshold x col_strips  (the two rules that interact: the hub fold) at auto's other settings, then coordinate sweeps
of the remaining options at the best point, then one more pass over threshold and strips.  Timings: min over 3 batches of the batch mean (device events);
auto and the best forced setting are then re-timed INTERLEAVED (5 alternating batches each) and the regret is taken from that pair.

Test / measurement infrastructure: loads oracle/ as the checker (like scripts/suite.py), never as the thing measured.
"""
import argparse
import itertools
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NONE_THR = 1 << 30          # "no hub": every row one exact segment


def graph_list(quick):
    """name -> builder(dev) returning device int32 (row_ptr, col_idx).  None of these draws its columns uniformly at random."""
    from hpc_amd import synth

    def ds(name, **kw):
        return lambda dev: synth.csr_dataset_structured_device(name, dev, **kw)

    def rcm(name):
        def f(dev):
            import torch

            p, i = synth.csr_dataset_structured_device(name, "cpu", order="shuffled")
            p, i = synth.csr_reorder_rcm(p.numpy(), i.numpy())
            return torch.from_numpy(p).to(dev), torch.from_numpy(i).to(dev)
        return f

    g = {
        "rmat20-unpermuted": lambda dev: synth.csr_rmat_device(20, dev),
        "sbm-1M-deg32": lambda dev: synth.csr_dcsbm_device(1 << 20, 32 << 20, 64, dev, alpha=0, mean_comm=2048, p_in=0.9),
        "arxiv-community": ds("arxiv"), "arxiv-rcm": rcm("arxiv"), "arxiv-degree": ds("arxiv", order="degree"), "arxiv-unsorted": ds("arxiv", sort_cols=False),
        "collab-community": ds("collab"), "collab-rcm": rcm("collab"),
        "ddi-community": ds("ddi"),
        "youtube-community": ds("youtube"), "youtube-shuffled": ds("youtube", order="shuffled"),
        "am-community": ds("am"), "am-degree": ds("am", order="degree"),
        "yelp-community": ds("yelp"),
        "wikikg2-community": ds("wikikg2"),
        "citation-community": ds("citation"),
        "ppa-community": ds("ppa"),
        "protein-community": ds("protein"), "protein-shuffled": ds("protein", order="shuffled"), "protein-unsorted": ds("protein", sort_cols=False),
        "reddit-community": ds("reddit.dgl"), "reddit-degree": ds("reddit.dgl", order="degree"),
        "products-community": ds("products"),
        # every column within +-2048 of the row (mesh-like): 300-700 nonzeros per row -> segments whose rows sit inside one or two column strips
        "banded-long-rows": lambda dev: synth.csr_banded_long_rows_device(1 << 17, dev),
        "banded-deg32": lambda dev: tuple(__import__("torch").from_numpy(a).to(dev) for a in synth.csr_banded(1 << 20)),
    }
    if quick:
        g = {k: g[k] for k in ("arxiv-community", "ddi-community", "youtube-community", "protein-community")}
    return g


def holdout_list():
    """Graphs NO rule was fitted on (--holdout): other orders of the same dataset shapes, the largest shape, a weaker block model, a smaller R-MAT, a wider
    band -- the check that the rules of plan.hpp were fitted to structure, not to the 25 graphs of graph_list()."""
    from hpc_amd import synth

    def ds(name, **kw):
        return lambda dev: synth.csr_dataset_structured_device(name, dev, **kw)

    return {
        "citation-shuffled": ds("citation", order="shuffled"), "citation-degree": ds("citation", order="degree"),
        "ppa-shuffled": ds("ppa", order="shuffled"), "ppa-degree": ds("ppa", order="degree"),
        "products-degree": ds("products", order="degree"),
        "yelp-degree": ds("yelp", order="degree"), "yelp-shuffled": ds("yelp", order="shuffled"),
        "wikikg2-shuffled": ds("wikikg2", order="shuffled"),
        "collab-degree": ds("collab", order="degree"),
        "reddit-shuffled": ds("reddit.dgl", order="shuffled"),
        "protein-degree": ds("protein", order="degree"),
        "ddi-shuffled": ds("ddi", order="shuffled"),
        "youtube-degree": ds("youtube", order="degree"),
        "amazon-community": ds("amazon_cogdl"),
        "weak-sbm-1M-deg32": lambda dev: synth.csr_dcsbm_device(1 << 20, 32 << 20, 64, dev, alpha=0, mean_comm=4096, p_in=0.6, seed=77),
        "rmat18-unpermuted": lambda dev: synth.csr_rmat_device(18, dev, seed=5),
        "banded-wide-long-rows": lambda dev: synth.csr_banded_long_rows_device(1 << 17, dev, width=16384, seed=9),
        "community-yelp-p05": ds("yelp", p_in=0.5),
    }


class Timer:
    def __init__(self):
        import torch

        self.torch = torch
        self.a, self.b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def batch(self, f, reps):
        self.torch.cuda.synchronize()
        self.a.record()
        for _ in range(reps):
            f()
        self.b.record()
        self.torch.cuda.synchronize()
        return self.a.elapsed_time(self.b) / reps


def summarize(path, before=None):
    # several files, comma separated: a later file's entry for the same (graph, kLen) replaces an earlier one (a re-run of the graphs a last rule touched)
    merged, n_files = {}, 0
    for one in path.split(","):
        n_files += 1
        for l in open(one):
            if l.startswith("{"):
                r = json.loads(l)
                if "auto_ms" in r:
                    r["_file"] = os.path.basename(one)
                    merged[(r["graph"], r["N"])] = r
    rows = list(merged.values())
    old = {}
    if before:
        old = {(r["graph"], r["N"]): r for r in (json.loads(l) for l in open(before) if l.startswith("{")) if "auto_ms" in r}
    print("# Regret of the auto rules on structured graphs (scripts/regret.py; every C bit-identical to `spmm_kernel_ref`)\n")
    print("Graphs: `hpc_amd/synth.py` `csr_dcsbm_device` (degree-corrected block model with the rows / nonzeros / longest row of the reference's datasets, symmetric: hub rows are hub columns), "
          "order = community (a BFS / RCM / partitioner order), shuffled, degree (hubs first); `-rcm`: a true reverse Cuthill-McKee order (scipy); `-unsorted`: columns in random order inside a row; "
          "`rmat20-unpermuted`; `sbm`: equal degrees, dense diagonal blocks.  `auto` and `best forced` re-timed interleaved; regret = auto / best - 1.\n")
    if old:
        print(f"`auto ms, round-4 rules`: the same graph and width measured by the same script before any rule was touched (`{os.path.basename(before)}`; another box: +-3 %).\n")
    print("| graph | rows | nnz | longest row | locality % | kLen | auto ms, round-4 rules | auto ms | best forced ms | regret % | best forced setting | single options that beat auto by > 4 % (time ratio) | auto's resolved setting | settings tried | bit-different settings | auto vs spmm_kernel_ref (differing elements) |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    worst = []
    short = {"long_row_threshold": "thr", "col_strips": "strips", "tile_cols": "tile", "medium_row_threshold": "mthr", "segment_overlap": "seg_ov", "hub_overlap": "hub_ov",
             "hub_slice": "slice", "fused_step": "fused", "fused_order": "f_order", "segment_order": "seg_order", "rows_unroll": "unroll"}

    def fmt(cfg):
        return ", ".join(f"{short.get(k, k)}={'none' if v == NONE_THR else v}" for k, v in cfg.items()) or "(auto)"

    for r in rows:
        a = r["auto_cfg"]
        singles = []
        for k, d in (r.get("one_at_a_time_vs_auto") or {}).items():
            best_v = min(((v, kk) for kk, v in d.items() if v), default=None)
            if best_v and best_v[0] < 0.96:
                singles.append(f"{short.get(k, k)}={best_v[1]}: {best_v[0]:.2f}")
        auto_s = (f"thr {a['thr'] if a['thr'] != NONE_THR else 'none'}, strips {a['S']}, tile {a['tile']}, mthr {a['mthr']}, {a['hubs']} hubs, {a['segments']} segments ({a['seg_nnz_pct']} % of nnz), "
                  f"{a['launches']} launch{'es' if a['launches'] != 1 else ''}{', small-step kernel' if a.get('fused') else ''}, front {a.get('front_pct')} %, preprocess {a['preprocess_us']} us")
        o = old.get((r["graph"], r["N"]))
        print(f"| {r['graph']} | {r['M']} | {r['nnz']} | {r['max_row']} | {a.get('locality_pct', '')} | {r['N']} | {(str(round(o['auto_ms'], 4)) if o else '-')} | {r['auto_ms']:.4f} | {r['best_ms']:.4f} | {r['regret_pct']:.1f} | "
              f"{fmt(r['best_cfg'])} | {'; '.join(singles) or '-'} | {auto_s} | {r['n_tried']} | {r['n_bitdiff']} | {r.get('auto_bitdiff_vs_spmm_kernel_ref')} |")
        worst.append((r["regret_pct"], r["graph"], r["N"]))
    worst.sort(reverse=True)
    if n_files > 1:
        later = sorted({(r["graph"]) for r in rows if r["_file"] != os.path.basename(path.split(",")[0])})
        print(f"\nEntries from a later run (the rules' last touches re-measured on the graphs they concern: {', '.join(later)}): {', '.join(os.path.basename(x) for x in path.split(',')[1:])}.")
    print(f"\nmax regret {worst[0][0]:.1f} % ({worst[0][1]}, kLen {worst[0][2]}); entries above 10 %: {sum(1 for w in worst if w[0] > 10)} of {len(worst)}; "
          f"median {sorted(w[0] for w in worst)[len(worst) // 2]:.1f} %")
    print("\nworst ten: " + "; ".join(f"{g} kLen {n}: {p:.1f} %" for p, g, n in worst[:10]))
    if old:
        both = [(o["auto_ms"] / r["auto_ms"]) for r in rows for o in [old.get((r["graph"], r["N"]))] if o]
        if both:
            print(f"\nauto, round-4 rules -> now, over the {len(both)} entries measured both times: geometric mean speed-up {float(np.exp(np.mean(np.log(both)))):.3f}, "
                  f"best {max(both):.2f}x, worst {min(both):.2f}x")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None, help="comma list of substrings")
    ap.add_argument("--lens", default="32,128,256")
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--autotune", action="store_true", help="the handle under test has \"autotune\" = 1 (preprocess measures instead of guessing): its regret against the same forced grid")
    ap.add_argument("--holdout", action="store_true", help="the graphs no rule was fitted on (holdout_list) instead of graph_list")
    ap.add_argument("--no-ref", action="store_true", help="skip the comparison with spmm_kernel_ref (bits are still compared with auto's)")
    ap.add_argument("--summarize", default=None)
    ap.add_argument("--before", default=None, help="with --summarize: the JSONL of the same script from before the rule fixes (adds a column)")
    ap.add_argument("--full-grid-below", type=int, default=8_000_000)
    args = ap.parse_args()
    if args.summarize:
        summarize(args.summarize, args.before)
        return
    import torch
    from hpc_amd import CSR, SpMMOpt
    from hpc_amd.spmm import count_bitdiff
    from oracle import oracle

    dev = torch.device("cuda:0")
    tm = Timer()
    lens = [int(x) for x in args.lens.split(",")]
    only = args.only.split(",") if args.only else None
    for gname, build in (holdout_list() if args.holdout else graph_list(args.quick)).items():
        if only and not any(o in gname for o in only):
            continue
        t0 = time.time()
        d_ptr, d_idx = build(dev)
        torch.cuda.synchronize()
        M, nnz = d_ptr.numel() - 1, int(d_idx.numel())
        d_val = torch.randn(nnz, device=dev) * 0.1
        max_row = int(torch.diff(d_ptr).max().item())
        g = CSR(M, nnz, d_ptr, d_idx, d_val)
        print(f"# {gname}: M={M} nnz={nnz} longest {max_row} built in {time.time() - t0:.1f}s", file=sys.stderr, flush=True)
        for N in lens:
            t1 = time.time()
            d_B = torch.randn(M, N, device=dev) * 0.1
            d_auto = torch.full((M, N), float("nan"), device=dev)
            d_C = torch.full((M, N), float("nan"), device=dev)
            est_ms = max(0.02, (nnz * (4.0 * N + 8) + 4.0 * M * N) / 8e9)
            reps = int(min(20, max(3, math.ceil(2.0 / est_ms))))

            def make(cfg):
                op = SpMMOpt(g, N)
                for k, v in cfg.items():
                    op.set_option(k, v)
                op.preprocess(d_B, d_C)
                return op

            def measure(op, out):
                for _ in range(2):
                    op.run(d_B, out)
                return min(tm.batch(lambda: op.run(d_B, out), reps) for _ in range(3))

            auto = make({"autotune": 1} if args.autotune else {})
            auto_ms = measure(auto, d_auto)
            acfg = {"thr": auto.get_option("long_row_threshold"), "S": auto.get_option("n_col_strips"), "tile": auto.get_option("lanes_per_row") * 4,
                    "mthr": auto.get_option("medium_row_threshold"), "hubs": auto.get_option("n_hub_rows"), "segments": auto.get_option("n_chunks"),
                    "seg_nnz_pct": round(100.0 * auto.get_option("segment_nnz") / max(1, nnz), 1), "locality_pct": auto.get_option("column_locality_pct"),
                    "unsorted": auto.get_option("segments_unsorted"), "launches": auto.get_option("n_launches"), "preprocess_us": auto.get_option("preprocess_us"),
                    "fused": auto.get_option("fused_step_in_force"), "front_pct": auto.get_option("column_front_pct")}
            if args.autotune:
                acfg.update({"autotune_evals": auto.get_option("autotune_evals"), "autotune_mask": auto.get_option("autotune_mask"),
                             "autotune_auto_us": auto.get_option("autotune_auto_us"), "autotune_best_us": auto.get_option("autotune_best_us"),
                             "tuned": {k: auto.get_option(k) for k in ("tile_cols", "col_strips", "medium_row_threshold", "fused_step")}})
            ref_diff = None
            if not args.no_ref and oracle.ref_available():
                d_R = torch.zeros((M, N), device=dev)
                oracle.ref_kernel_run(d_ptr, d_idx, d_val, d_B, d_R, M, N)
                torch.cuda.synchronize()
                ref_diff = count_bitdiff(d_auto, d_R)[0]
                del d_R
            tried, n_bitdiff = {}, 0

            def evaluate(cfg):
                nonlocal n_bitdiff
                key = tuple(sorted(cfg.items()))
                if key in tried:
                    return tried[key]
                try:
                    op = make(cfg)
                    d_C.fill_(float("nan"))
                    ms = measure(op, d_C)
                    nd = count_bitdiff(d_C, d_auto)[0]
                    sig = (op.get_option("n_hub_rows"), op.get_option("n_chunks"), op.get_option("n_col_strips"))
                    del op
                except Exception as e:       # a refused option is a finding, not a crash
                    print(json.dumps({"graph": gname, "N": N, "cfg": cfg, "error": repr(e)[:160]}), flush=True)
                    tried[key] = (float("inf"), None)
                    return tried[key]
                if nd:
                    n_bitdiff += 1
                    print(json.dumps({"graph": gname, "N": N, "cfg": cfg, "bitdiff_vs_auto": nd}), flush=True)
                    ms = float("inf")
                tried[key] = (ms, sig)
                return tried[key]

            S_auto = acfg["S"]
            strips = sorted({1, S_auto, max(2, S_auto // 2), min(64, 2 * S_auto)}) if S_auto > 1 else [1, 4, 12]
            thrs = [t for t in (256, 512, 1024, 2048, 4096, 8192) if t < max_row] + [NONE_THR]
            tiles = sorted({0} | {t for t in (64, 128, 256) if t < N or t == 256})
            best = ({}, auto_ms)

            def consider(cfg):
                nonlocal best
                ms, _ = evaluate(cfg)
                if ms < best[1]:
                    best = (dict(cfg), ms)

            def hub_dims(cfg):
                """hub_slice / segment_overlap variants that can matter for this plan (probe once without them)."""
                _, sig = evaluate(cfg)
                if sig is None:
                    return [{}]
                hubs, segs, _ = sig
                hs = [{"hub_slice": 16}, {"hub_slice": 32}] if hubs > 0 else [{}]
                so = [{"segment_overlap": 0}, {"segment_overlap": 1}] if segs > 0 else [{}]
                return [dict(**a, **b) for a in hs for b in so]

            # one option at a time from auto: which RULE is at fault when the best forced setting changes several at once
            marg = {}
            for key_, vals_ in (("tile_cols", [t for t in tiles if t]), ("col_strips", strips), ("long_row_threshold", thrs), ("medium_row_threshold", [32, 64, 128, 256, 512, 1024]), ("segment_order", [1, 2]),
                                ("segment_overlap", [0, 1]), ("hub_overlap", [0, 2]), ("hub_slice", [16, 32] if acfg["hubs"] > 0 else []), ("fused_step", [0, 1]), ("fused_order", [1, 2]), ("rows_unroll", [16])):
                for v_ in vals_:
                    consider({key_: v_})
                    ms_, _ = evaluate({key_: v_})
                    marg.setdefault(key_, {})[str(v_) if v_ != NONE_THR else "none"] = round(ms_ / auto_ms, 3) if ms_ < float("inf") else None
            if nnz < args.full_grid_below:
                for T, S, tile in itertools.product(thrs, strips, tiles):
                    base = {"long_row_threshold": T, "col_strips": S}
                    if tile:
                        base["tile_cols"] = tile
                    consider(base)                       # (fused_step auto: the small-step kernel where eligible)
                    if S == strips[0]:
                        for fo in (1, 2):                # the small-step kernel's role order interacts with the hub threshold (which role's chain is the step)
                            consider(dict(base, fused_order=fo))
                    sep = dict(base, fused_step=0)       # the separate kernels, with the options that only they have
                    consider(sep)
                    for extra in hub_dims(sep):
                        if extra:
                            consider(dict(sep, **extra))
            else:
                for T, S in itertools.product(thrs, strips):
                    consider({"long_row_threshold": T, "col_strips": S})
                for _pass in range(2):
                    cur = dict(best[0])
                    for tile in tiles:
                        c = dict(cur)
                        c.pop("tile_cols", None)
                        if tile:
                            c["tile_cols"] = tile
                        consider(c)
                    cur = dict(best[0])
                    for extra in hub_dims(cur):
                        consider(dict(cur, **extra))
                    cur = dict(best[0])
                    for m in (32, 64, 128, 256, 512, 1024):
                        consider(dict(cur, medium_row_threshold=m))
                    cur = dict(best[0])
                    for so in (1, 2):
                        consider(dict(cur, segment_order=so))
                    cur = dict(best[0])
                    for ho in (0, 2):
                        consider(dict(cur, hub_overlap=ho))
                    cur = dict(best[0])
                    for fs in (0, 1):
                        consider(dict(cur, fused_step=fs))
                    cur = dict(best[0])
                    for T in thrs:
                        consider(dict(cur, long_row_threshold=T))
                    cur = dict(best[0])
                    for S in strips:
                        consider(dict(cur, col_strips=S))
                    consider(dict(best[0], rows_unroll=16))
            # the medium threshold and hub_overlap for the small graphs too (coordinate sweeps at the best point)
            if nnz < args.full_grid_below:
                cur = dict(best[0])
                for m in (32, 64, 128, 256, 512, 1024):
                    consider(dict(cur, medium_row_threshold=m))
                cur = dict(best[0])
                for so in (1, 2):
                    consider(dict(cur, segment_order=so))
                cur = dict(best[0])
                for ho in (0, 2):
                    consider(dict(cur, hub_overlap=ho))
                consider(dict(best[0], rows_unroll=16))
            # auto against the best forced setting, interleaved
            best_cfg, best_ms = best
            a_ms, b_ms = auto_ms, best_ms
            if best_cfg:
                bop = make(best_cfg)
                for _ in range(2):
                    bop.run(d_B, d_C)
                    auto.run(d_B, d_auto)
                ta, tb = [], []
                for _ in range(5):
                    ta.append(tm.batch(lambda: auto.run(d_B, d_auto), reps))
                    tb.append(tm.batch(lambda: bop.run(d_B, d_C), reps))
                a_ms, b_ms = min(ta), min(tb)
                del bop
            if b_ms > a_ms:
                b_ms, best_cfg = a_ms, {}
            row = {"graph": gname, "M": M, "nnz": nnz, "max_row": max_row, "N": N, "auto_ms": round(a_ms, 5), "best_ms": round(b_ms, 5),
                   "regret_pct": round(100.0 * (a_ms / b_ms - 1.0), 2), "best_cfg": best_cfg, "auto_cfg": acfg, "n_tried": len(tried), "n_bitdiff": n_bitdiff,
                   "auto_bitdiff_vs_spmm_kernel_ref": ref_diff, "one_at_a_time_vs_auto": marg, "first_pass_auto_ms": round(auto_ms, 5), "first_pass_best_ms": round(best_ms, 5), "seconds": round(time.time() - t1, 1)}
            print(json.dumps(row), flush=True)
            del auto, d_B, d_auto, d_C
            torch.cuda.empty_cache()
        del d_ptr, d_idx, d_val, g
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
