#!/usr/bin/env python3
"""A/B of ONE option of the handle on a few graphs, interleaved in one process, bits compared:  python scripts/debug/option_ab.py KEY A B [--graphs a,b] [--lens 32,128]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from hpc_amd import CSR, SpMMOpt, synth
from hpc_amd.spmm import count_bitdiff

ap = argparse.ArgumentParser()
ap.add_argument("key"); ap.add_argument("a", type=int); ap.add_argument("b", type=int)
ap.add_argument("--graphs", default="C1,C2,rmat20,ppa-shuffled,ppa-degree,yelp-degree,youtube-degree,citation-shuffled")
ap.add_argument("--lens", default="32,128,256")
args = ap.parse_args()
dev = torch.device("cuda:0")
def host(t): return tuple(torch.from_numpy(x).to(dev) for x in t)
def ds(name, order): return lambda: synth.csr_dataset_structured_device(name, dev, order=order)
graphs = {"C1": lambda: host(synth.csr_uniform(1 << 20, 16, 48)), "C2": lambda: host(synth.csr_powerlaw(1 << 20, 32.0, 4096)), "rmat20": lambda: host(synth.csr_rmat(20, 32)),
          "denseish": lambda: host(synth.csr_uniform(1 << 18, 300, 700))}
for n in ("ppa", "yelp", "youtube", "citation", "products", "protein", "reddit", "wikikg2", "collab", "arxiv", "am", "ddi"):
    for o in ("shuffled", "degree", "community"):
        graphs[f"{n}-{o}"] = ds(n, o)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
def batch(f, reps):
    torch.cuda.synchronize(); ev[0].record()
    for _ in range(reps): f()
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps
for name in args.graphs.split(","):
    p, i = graphs[name]()
    M, nnz = p.numel() - 1, int(i.numel())
    v = torch.randn(nnz, device=dev) * 0.1
    for N in [int(x) for x in args.lens.split(",")]:
        B = torch.randn(M, N, device=dev) * 0.1
        ops, Cs = [], []
        for u in (args.a, args.b):
            op = SpMMOpt(CSR(M, nnz, p, i, v), N); op.set_option(args.key, u); C = torch.full((M, N), float("nan"), device=dev); op.preprocess(B, C)
            for _ in range(2): op.run(B, C)
            ops.append(op); Cs.append(C)
        reps = 5 if nnz * N > 1 << 30 else 20
        t = [[], []]
        for _ in range(4):
            for k in (0, 1): t[k].append(batch(lambda: ops[k].run(B, Cs[k]), reps))
        print(json.dumps({"graph": name, "N": N, "key": args.key, f"ms_{args.a}": round(min(t[0]), 4), f"ms_{args.b}": round(min(t[1]), 4), "ratio_b_over_a": round(min(t[1]) / min(t[0]), 3),
                          "bitdiff": count_bitdiff(Cs[0], Cs[1])[0], "local": ops[0].get_option("column_locality_pct"), "mthr": ops[0].get_option("medium_row_threshold"),
                          "hubs": ops[0].get_option("n_hub_rows"), "segments": ops[0].get_option("n_chunks"), "launches": [o.get_option("n_launches") for o in ops]}), flush=True)
        del ops, Cs, B
