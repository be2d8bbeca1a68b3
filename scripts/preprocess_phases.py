import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from hpc_amd import CSR, SpMMOpt, synth
dev = torch.device('cuda:0')
for name, gen in (("C1", lambda: synth.csr_uniform(1<<20,16,48)), ("C2", lambda: synth.csr_powerlaw(1<<20)), ("RMAT", lambda: synth.csr_rmat(20,32)), ("C4", lambda: synth.csr_block_dense_fast(1<<20))):
    ptr, idx = gen(); vals = synth.make_values(idx.size)
    d = [torch.from_numpy(a).to(dev) for a in (ptr, idx, vals)]
    B = torch.zeros(1<<20, 128, device=dev); C = torch.empty_like(B)
    op = SpMMOpt(CSR(ptr.size-1, idx.size, *d), 128)
    for rep in range(3):
        torch.cuda.synchronize(); t=time.perf_counter(); op.preprocess(B, C); torch.cuda.synchronize(); dt=(time.perf_counter()-t)*1e3
    print(name, "total ms %.2f"%dt, {k: op.get_option(k) for k in ("pre_d2h_us","pre_colcheck_us","pre_detect_us","pre_table_us","pre_upload_us","n_chunks","n_block_groups")})
