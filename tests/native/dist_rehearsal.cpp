// tests/native/dist_rehearsal.cpp -- include/mi_spmm_dist.h driven from plain C++ (what a C++ host of the reference
// would do with the `ncclComm_t` its util.h:30 only hints at): a world of ONE over RCCL with "rehearse" on, so that
// the communicator bootstrap, the panel pipeline, the in-place all-gather / grouped send-recv + re-layout kernel and
// the IPC + strided-copy exchange all execute on a one-GPU box.  Checked against the plain operator on the device
// (bit for bit); the CPU oracle is not involved.
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "mi_spmm_dist.h"

#define CK(x) do { int c_ = (int)(x); if (c_ != 0) { std::fprintf(stderr, "%s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, #x, c_, mi_spmm_dist_strerror(c_)); return 1; } } while (0)

int main()
{
    const int M = 20000, K = 20000, N = 128;
    // a small random CSR: 0..31 nonzeros per row (LCG, fixed seed)
    std::vector<int> ptr(M + 1, 0), idx;
    std::vector<float> val;
    unsigned long long s = 12345;
    auto rnd = [&]() { s = s * 6364136223846793005ULL + 1442695040888963407ULL; return (unsigned)(s >> 33); };
    for (int r = 0; r < M; ++r) {
        const int d = (int)(rnd() % 32);
        for (int k = 0; k < d; ++k) { idx.push_back((int)(rnd() % K)); val.push_back((float)((int)(rnd() % 2001) - 1000) * 1e-3f); }
        ptr[r + 1] = (int)idx.size();
    }
    const long long nnz = (long long)idx.size();
    int *d_ptr, *d_idx;
    float *d_val, *d_B, *d_C, *d_ref;
    CK(hipMalloc((void **)&d_ptr, sizeof(int) * (M + 1)));
    CK(hipMalloc((void **)&d_idx, sizeof(int) * nnz));
    CK(hipMalloc((void **)&d_val, sizeof(float) * nnz));
    CK(hipMalloc((void **)&d_B, sizeof(float) * (size_t)K * N));
    CK(hipMalloc((void **)&d_C, sizeof(float) * (size_t)M * N));
    CK(hipMalloc((void **)&d_ref, sizeof(float) * (size_t)M * N));
    CK(hipMemcpy(d_ptr, ptr.data(), sizeof(int) * (M + 1), hipMemcpyHostToDevice));
    CK(hipMemcpy(d_idx, idx.data(), sizeof(int) * nnz, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_val, val.data(), sizeof(float) * nnz, hipMemcpyHostToDevice));
    CK(mi_spmm_fill_normal(d_B, (int64_t)K * N, 123, 0, 0.f, 0.1f, nullptr));
    mi_spmm_handle *h = nullptr;
    CK(mi_spmm_create(&h, d_ptr, d_idx, d_val, M, K, nnz, N));
    CK(mi_spmm_preprocess(h, d_B, d_ref));
    CK(mi_spmm_run(h, d_B, d_ref, nullptr));
    CK(hipDeviceSynchronize());

    hipStream_t stream;
    CK(hipStreamCreate(&stream));
    const char *names[4] = {"allgather", "direct", "peer2d", "peer_store"};
    for (int ex = 0; ex < 4; ++ex) {
        mi_spmm_dist *d = nullptr;
        CK(mi_spmm_dist_create(&d, h, M, N, /*rank*/ 0, /*world*/ 1, /*panels*/ 4));
        char id[MI_SPMM_DIST_UNIQUE_ID_BYTES];
        CK(mi_spmm_dist_unique_id(id));              // a multi-rank host broadcasts these 128 bytes from rank 0
        CK(mi_spmm_dist_comm_init(d, id));
        CK(mi_spmm_dist_set_option(d, "exchange", ex));
        CK(mi_spmm_dist_set_option(d, "rehearse", 1));
        if (ex >= 2) {
            char handle[MI_SPMM_DIST_IPC_HANDLE_BYTES];
            int64_t off = 0;
            CK(mi_spmm_dist_export_c(d, d_C, handle, &off));     // a multi-rank host all-gathers handles and offsets
            CK(mi_spmm_dist_set_peers(d, d_C, handle, &off));
            // the link probe a multi-rank host runs once after set_peers; a world of one has no peer: every rate 0, nothing copied
            double per_peer[1] = {-1.0}, all_peers = -1.0;
            int32_t link_type[1] = {0}, hops[1] = {0};
            CK(mi_spmm_dist_link_probe(d, d_C, 0, nullptr, per_peer, &all_peers, link_type, hops));
            if (per_peer[0] != 0.0 || all_peers != 0.0 || link_type[0] != -1 || hops[0] != -1) { std::fprintf(stderr, "link probe at world 1: unexpected result\n"); return 3; }
        }
        CK(hipMemsetAsync(d_C, 0xff, sizeof(float) * (size_t)M * N, stream));   // NaN pattern
        for (int it = 0; it < 3; ++it) CK(mi_spmm_dist_run(d, d_B, d_C, (void *)stream));
        CK(hipStreamSynchronize(stream));
        int64_t ndiff = -1;
        float maxabs = 0.f;
        CK(mi_spmm_count_bitdiff(d_C, d_ref, (int64_t)M * N, &ndiff, &maxabs, nullptr));
        int64_t staging = 0;
        CK(mi_spmm_dist_get_option(d, "staging_bytes", &staging));
        std::printf("exchange %s: %s (bitdiff %lld, staging %lld bytes)\n", names[ex], ndiff == 0 ? "bit-identical" : "DIFFERENT",
                    (long long)ndiff, (long long)staging);
        CK(mi_spmm_dist_destroy(d));
        if (ndiff != 0) return 2;
    }
    CK(mi_spmm_destroy(h));
    return 0;
}
