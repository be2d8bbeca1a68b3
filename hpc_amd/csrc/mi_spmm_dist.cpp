// mi_spmm_dist.cpp -- the column-sharded multi-GPU step behind a C ABI (include/mi_spmm_dist.h).
//
// Host code only: streams, events, RCCL calls, strided peer copies.  The kernels are libmi_spmm.so's
// (mi_spmm_run_rows, mi_spmm_unpack_gathered).  One process per GPU; the reference has no counterpart
// (PA4/workspace/include/util.h:30 is a commented-out `extern ncclComm_t comm;`).
//
//   compute stream (the caller's) :  rows(p0) | rows(p1) | rows(p2) | ...
//   exchange stream(s)            :           | gather(p0) | gather(p1) | ...          (back to back: link-bound)
//   re-layout stream              :                        | unpack(p0) | unpack(p1) | ...     (not in peer2d mode)
#include "../../include/mi_spmm_dist.h"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <utility>
#include <vector>

namespace {

enum { kAllGather = 0, kDirect = 1, kPeer2D = 2, kPeerStore = 3, kIpcPull = 4 };

#define HIP_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return (int)e_; } while (0)
#define NCCL_TRY(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) return MI_SPMM_DIST_ENCCL_BASE - (int)r_; } while (0)
#define MI_TRY(x) do { int c_ = (x); if (c_ != 0) return c_; } while (0)

}  // namespace

struct mi_spmm_dist {
    uint32_t magic = 0x4d494453u;  // "MIDS"
    mi_spmm_handle *h = nullptr;
    int32_t M = 0, n_loc = 0, rank = 0, world = 1, n_panels = 1;
    int64_t N_total = 0;
    int exchange = kAllGather;
    bool rehearse = false;                     // run the exchange machinery even at world == 1 (one-GPU rehearsal)
    int comm_stream_overlaps = -1, post_stream_overlaps = -1;   // mi_spmm_stream_create_concurrent's verdicts (-1: streams not made yet)
    bool external_barrier = false;             // peer2d without a communicator: the CALLER brackets every step with a cross-rank barrier
    bool ipc_any_size = false;                 // export allocations of any size (see mi_spmm_dist_ipc_exportable_bytes)
    std::vector<std::pair<int32_t, int32_t>> panels;
    int32_t rows_max = 0;
    // streams / events (created on first use, on the device current at that time)
    bool streams_ready = false;
    hipStream_t s_comm = nullptr, s_post = nullptr;
    std::vector<hipStream_t> s_push;           // peer2d: one per peer, so every link runs at the same time
    hipEvent_t ev_start = nullptr;
    std::vector<hipEvent_t> ev_computed, ev_gathered;
    hipEvent_t ev_unpacked[2] = {nullptr, nullptr};
    bool unpacked_valid[2] = {false, false};
    std::vector<hipEvent_t> ev_pushed;         // per peer stream
    // staging (allgather / direct)
    float *staging[2] = {nullptr, nullptr};
    size_t staging_elems = 0;
    // communicator
    ncclComm_t comm = nullptr;
    bool own_comm = false;
    float *d_token = nullptr;                  // one float: the all-reduce that serves as a device-side barrier
    // peers (peer2d)
    std::vector<void *> peer_base;             // what hipIpcOpenMemHandle returned (to close)
    std::vector<float *> peer_C;               // the peers' C_full
    float *exported_C = nullptr;
    // ipc_pull: the peers' two staging buffers, mapped through HIP IPC
    std::vector<void *> peer_stage_base;       // [world * 2], to close
    std::vector<float *> peer_stage;           // [world * 2]: peer q's staging[b] at q * 2 + b
    // a cross-rank barrier supplied by the host, used where a step needs one and no communicator exists
    mi_spmm_dist_barrier_fn host_barrier = nullptr;
    void *host_barrier_ctx = nullptr;
};

namespace {

bool good(const mi_spmm_dist *d) { return d && d->magic == 0x4d494453u; }

// hipIpcOpenMemHandle never returns for an allocation whose SIZE has bit 31 set (2 GiB <= size mod 4 GiB): measured with the HIP
// runtime of the torch 2.10+rocm7.0 wheel in dmabuf IPC mode (HSA_ENABLE_IPC_MODE_LEGACY=0), two processes on one MI355X --
// 1, 1.5, 4 and 5 GiB open in under a millisecond and read back right to their last byte; 2, 2.002, 3 and 6 GiB hang in the
// importer (scripts/debug/ipc_open_probe.py, profiles/r04_ipc_open_sizes.txt).  An exporter cannot un-hang its peers, so it refuses.
bool ipc_size_ok(size_t size) { return ((size >> 31) & 1u) == 0; }

int refuse_ipc_size(const char *where, size_t size)
{
    if (const char *dbg = std::getenv("MI_SPMM_DEBUG"))
        if (dbg[0] == '1')
            std::fprintf(stderr, "%s: the allocation is %zu bytes; hipIpcOpenMemHandle hangs on sizes with bit 31 set -- allocate %lld bytes "
                                 "(mi_spmm_dist_ipc_exportable_bytes) or set \"ipc_any_size\"\n", where, size,
                         (long long)mi_spmm_dist_ipc_exportable_bytes((int64_t)size));
    return MI_SPMM_EUNSUPPORTED;
}

void make_panels(mi_spmm_dist *d)
{
    d->panels.clear();
    d->rows_max = 0;
    if (d->M <= 0) return;
    const int64_t np = d->n_panels < 1 ? 1 : d->n_panels;
    int64_t per = (d->M + np - 1) / np;
    per = (per + 255) / 256 * 256;
    for (int64_t r = 0; r < d->M; r += per) {
        const int64_t e = r + per < d->M ? r + per : d->M;
        d->panels.emplace_back((int32_t)r, (int32_t)e);
        if (e - r > d->rows_max) d->rows_max = (int32_t)(e - r);
    }
}

void close_peer_staging(mi_spmm_dist *d);

void free_staging(mi_spmm_dist *d)
{
    close_peer_staging(d);       // the peers' view of OUR buffers dies with them too: ipc_pull needs a new export / set round
    for (int i = 0; i < 2; ++i) {
        if (d->staging[i]) (void)hipFree(d->staging[i]);
        d->staging[i] = nullptr;
    }
    d->staging_elems = 0;
}

int ensure_streams(mi_spmm_dist *d)
{
    if (d->streams_ready) return 0;
    // exchange and re-layout at high priority: they must not queue behind a compute kernel that fills every CU -- and they must run BESIDE it,
    // which a stream does or does not depending on the hardware queue the runtime hands it: tested candidates (mi_spmm.h)
    {
        void *sc = nullptr, *sp = nullptr;
        MI_TRY(mi_spmm_stream_create_concurrent(&sc, 1, &d->comm_stream_overlaps));
        d->s_comm = (hipStream_t)sc;
        MI_TRY(mi_spmm_stream_create_concurrent(&sp, 1, &d->post_stream_overlaps));
        d->s_post = (hipStream_t)sp;
    }
    HIP_TRY(hipEventCreateWithFlags(&d->ev_start, hipEventDisableTiming));
    for (int i = 0; i < 2; ++i) HIP_TRY(hipEventCreateWithFlags(&d->ev_unpacked[i], hipEventDisableTiming));
    d->streams_ready = true;
    return 0;
}

int ensure_panel_events(mi_spmm_dist *d)
{
    while (d->ev_computed.size() < d->panels.size()) {
        hipEvent_t a = nullptr, b = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&a, hipEventDisableTiming));
        d->ev_computed.push_back(a);
        HIP_TRY(hipEventCreateWithFlags(&b, hipEventDisableTiming));
        d->ev_gathered.push_back(b);
    }
    return 0;
}

int ensure_push_streams(mi_spmm_dist *d)
{
    int lo = 0, hi = 0;
    HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
    while ((int)d->s_push.size() < d->world - 1) {
        hipStream_t s = nullptr;
        HIP_TRY(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, hi));
        d->s_push.push_back(s);
        hipEvent_t e = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        d->ev_pushed.push_back(e);
    }
    return 0;
}

int ensure_staging(mi_spmm_dist *d)
{
    const size_t need = (size_t)d->world * (size_t)d->rows_max * (size_t)d->n_loc;
    if (d->staging[0] && d->staging_elems >= need) return 0;
    free_staging(d);
    // allocated at a size the runtime can export (ipc_pull maps the peers' staging buffers; mi_spmm_dist_ipc_exportable_bytes)
    const size_t bytes = (size_t)mi_spmm_dist_ipc_exportable_bytes((int64_t)((need ? need : 1) * sizeof(float)));
    for (int i = 0; i < 2; ++i)
        if (hipMalloc((void **)&d->staging[i], bytes) != hipSuccess) { free_staging(d); return MI_SPMM_ENOMEM; }
    d->staging_elems = need;
    d->unpacked_valid[0] = d->unpacked_valid[1] = false;
    return 0;
}

int ensure_token(mi_spmm_dist *d)
{
    if (d->d_token) return 0;
    HIP_TRY(hipMalloc((void **)&d->d_token, 256));
    HIP_TRY(hipMemset(d->d_token, 0, 256));
    return 0;
}

// every rank has finished everything it enqueued on its exchange stream before this point
int device_barrier(mi_spmm_dist *d, hipStream_t s)
{
    if (!d->comm) return 0;
    MI_TRY(ensure_token(d));
    NCCL_TRY(ncclAllReduce(d->d_token, d->d_token, 1, ncclFloat, ncclSum, d->comm, s));
    return 0;
}

void close_peer_staging(mi_spmm_dist *d)
{
    for (void *b : d->peer_stage_base)
        if (b) (void)hipIpcCloseMemHandle(b);
    d->peer_stage_base.clear();
    d->peer_stage.clear();
}

// Every rank has reached this point of the step: the communicator's one-element all-reduce on stream s, or -- without a
// communicator -- the host's barrier after this rank's two streams have drained (rehearsal-grade: it serialises the step).
int step_barrier(mi_spmm_dist *d, hipStream_t s, hipStream_t also)
{
    if (d->comm) return device_barrier(d, s);
    if (!d->host_barrier) return MI_SPMM_ESTATE;
    HIP_TRY(hipStreamSynchronize(also));
    HIP_TRY(hipStreamSynchronize(s));
    d->host_barrier(d->host_barrier_ctx);
    return 0;
}

// The barriers of a peer_store / peer2d step ("nobody still reads the C I am about to store into", "every store has landed"):
// the communicator's all-reduce; without one, the host's barrier callback when one is registered (so an in-process or
// shared-GPU driver synchronises itself); only with neither does "external_barrier" apply -- the caller has promised to
// bracket every step -- and without that promise step() has already refused.
int peer_barrier(mi_spmm_dist *d, hipStream_t s, hipStream_t also)
{
    if (d->comm) return device_barrier(d, s);
    if (d->host_barrier) return step_barrier(d, s, also);
    return d->external_barrier ? 0 : MI_SPMM_ESTATE;
}

void close_peers(mi_spmm_dist *d)
{
    for (void *b : d->peer_base)
        if (b) (void)hipIpcCloseMemHandle(b);
    d->peer_base.clear();
    d->peer_C.clear();
}

// fill stage[world][rows][n_loc] with every rank's block; the rank's own block already sits in its slot
int exchange_panel(mi_spmm_dist *d, float *stage, int32_t rows, hipStream_t s)
{
    const size_t blk = (size_t)rows * (size_t)d->n_loc;
    float *own = stage + (size_t)d->rank * blk;
    if (d->exchange == kAllGather) {
        NCCL_TRY(ncclAllGather(own, stage, blk, ncclFloat, d->comm, s));   // in place: sendbuff == recvbuff + rank * count
        return 0;
    }
    // all-pairs: rank r sends to r+k while it receives from r-k, every k a perfect matching
    NCCL_TRY(ncclGroupStart());
    for (int k = 1; k < d->world; ++k) {
        const int to = (d->rank + k) % d->world, from = (d->rank - k + d->world) % d->world;
        ncclResult_t r = ncclSend(own, blk, ncclFloat, to, d->comm, s);
        if (r == ncclSuccess) r = ncclRecv(stage + (size_t)from * blk, blk, ncclFloat, from, d->comm, s);
        if (r != ncclSuccess) { (void)ncclGroupEnd(); return MI_SPMM_DIST_ENCCL_BASE - (int)r; }
    }
    NCCL_TRY(ncclGroupEnd());
    return 0;
}

int step(mi_spmm_dist *d, const float *d_B_loc, float *d_C_full, hipStream_t main, bool do_compute, bool do_exchange)
{
    if (!good(d)) return MI_SPMM_ESTATE;
    if (d->M == 0 || d->n_loc == 0) return 0;
    if (!d_C_full || (do_compute && !d_B_loc)) return MI_SPMM_EINVAL;
    const int64_t n_loc = d->n_loc, NT = d->N_total;
    if (d->world == 1 && !d->rehearse) {   // nothing to exchange: the local block IS C
        if (do_compute) return mi_spmm_run_rows(d->h, d_B_loc, n_loc, d_C_full, NT, 0, d->M, (void *)main);
        return 0;
    }
    if (d->exchange == kPeer2D || d->exchange == kPeerStore) {
        if ((int)d->peer_C.size() != d->world || d->exported_C != d_C_full) return MI_SPMM_ESTATE;
        if (d->exchange == kPeerStore && d->world - 1 > 7) return MI_SPMM_EUNSUPPORTED;   // mi_spmm_run_rows_multi: at most 7 extra destinations
        // the two device-side barriers of a peer2d step are all-reduces on the communicator: without one, ranks would push
        // into C_full buffers that may still be read, and nobody would know when the pushes have landed -- refuse, unless
        // the caller has said that it brackets every step with a cross-rank barrier of its own ("external_barrier")
        if (do_exchange && d->world > 1 && !d->comm && !d->host_barrier && !d->external_barrier) return MI_SPMM_ESTATE;
    } else if (d->exchange == kIpcPull) {
        if ((int)d->peer_stage.size() != 2 * d->world) return MI_SPMM_ESTATE;
        if (d->world > 1 && !d->comm && !d->host_barrier) return MI_SPMM_ESTATE;
    } else if (!d->comm) return MI_SPMM_ESTATE;
    MI_TRY(ensure_streams(d));
    MI_TRY(ensure_panel_events(d));
    HIP_TRY(hipEventRecord(d->ev_start, main));
    HIP_TRY(hipStreamWaitEvent(d->s_comm, d->ev_start, 0));   // earlier work on the caller's stream (readers of C_full, staging)

    if (d->exchange == kPeerStore) {
        // The kernels' epilogues store every finished row segment into the local C_full AND into every peer's (IPC-mapped):
        // no copy, no staging, no re-layout, nothing to pipeline -- one launch set over all rows between two barriers.
        float *own = d_C_full + (size_t)d->rank * (size_t)n_loc;
        float *extra[8];
        int n_extra = 0;
        if (do_exchange)
            for (int q = 0; q < d->world; ++q)
                if (q != d->rank) extra[n_extra++] = d->peer_C[(size_t)q] + (size_t)d->rank * (size_t)n_loc;
        if (do_exchange) {
            MI_TRY(peer_barrier(d, d->s_comm, main));               // nobody still reads the C_full we are about to store into
            HIP_TRY(hipEventRecord(d->ev_gathered[0], d->s_comm));
            HIP_TRY(hipStreamWaitEvent(main, d->ev_gathered[0], 0));
        }
        if (do_compute) MI_TRY(mi_spmm_run_rows_multi(d->h, d_B_loc, n_loc, own, NT, 0, d->M, n_extra, extra, (void *)main));
        if (do_exchange) {
            HIP_TRY(hipEventRecord(d->ev_computed[0], main));
            HIP_TRY(hipStreamWaitEvent(d->s_comm, d->ev_computed[0], 0));
            MI_TRY(peer_barrier(d, d->s_comm, main));               // every rank's stores have landed (a finished kernel's stores are visible)
            HIP_TRY(hipEventRecord(d->ev_gathered[0], d->s_comm));
            HIP_TRY(hipStreamWaitEvent(main, d->ev_gathered[0], 0));
        }
        return 0;
    }
    if (d->exchange == kPeer2D) {
        MI_TRY(ensure_push_streams(d));
        // nobody may still be consuming the C_full we are about to overwrite remotely
        if (do_exchange) MI_TRY(peer_barrier(d, d->s_comm, main));
        HIP_TRY(hipEventRecord(d->ev_gathered[0], d->s_comm));     // reused as "step may start pushing"
        for (hipStream_t s : d->s_push) HIP_TRY(hipStreamWaitEvent(s, d->ev_gathered[0], 0));
        float *own = d_C_full + (size_t)d->rank * (size_t)n_loc;
        for (size_t p = 0; p < d->panels.size(); ++p) {
            const int32_t r0 = d->panels[p].first, r1 = d->panels[p].second;
            if (do_compute) MI_TRY(mi_spmm_run_rows(d->h, d_B_loc, n_loc, own, NT, r0, r1, (void *)main));
            HIP_TRY(hipEventRecord(d->ev_computed[p], main));
            if (!do_exchange) continue;
            int si = 0;
            for (int k = 1; k < d->world; ++k, ++si) {
                const int to = (d->rank + k) % d->world;
                HIP_TRY(hipStreamWaitEvent(d->s_push[si], d->ev_computed[p], 0));
                const size_t off = (size_t)r0 * (size_t)NT + (size_t)d->rank * (size_t)n_loc;
                HIP_TRY(hipMemcpy2DAsync(d->peer_C[to] + off, (size_t)NT * 4, d_C_full + off, (size_t)NT * 4, (size_t)n_loc * 4,
                                         (size_t)(r1 - r0), hipMemcpyDeviceToDevice, d->s_push[si]));
            }
        }
        if (do_exchange) {
            for (size_t i = 0; i < d->s_push.size(); ++i) {
                HIP_TRY(hipEventRecord(d->ev_pushed[i], d->s_push[i]));
                HIP_TRY(hipStreamWaitEvent(d->s_comm, d->ev_pushed[i], 0));
            }
            MI_TRY(peer_barrier(d, d->s_comm, main));               // every rank's pushes have landed
            HIP_TRY(hipEventRecord(d->ev_gathered[0], d->s_comm));
            HIP_TRY(hipStreamWaitEvent(main, d->ev_gathered[0], 0));
        }
        return 0;
    }

    MI_TRY(ensure_staging(d));
    HIP_TRY(hipStreamWaitEvent(d->s_post, d->ev_start, 0));
    for (size_t p = 0; p < d->panels.size(); ++p) {
        const int32_t r0 = d->panels[p].first, r1 = d->panels[p].second, rows = r1 - r0;
        float *stage = d->staging[p & 1];
        float *own = stage + (size_t)d->rank * (size_t)rows * (size_t)n_loc;
        // the staging buffer's previous contents (panel p-2, or the previous step) have been consumed
        if (d->unpacked_valid[p & 1]) HIP_TRY(hipStreamWaitEvent(main, d->ev_unpacked[p & 1], 0));
        // the rank's block goes straight into its slot of the staging buffer: the all-gather is in place
        if (do_compute) MI_TRY(mi_spmm_run_rows(d->h, d_B_loc, n_loc, own - (size_t)r0 * (size_t)n_loc, n_loc, r0, r1, (void *)main));
        HIP_TRY(hipEventRecord(d->ev_computed[p], main));
        if (!do_exchange) continue;
        HIP_TRY(hipStreamWaitEvent(d->s_comm, d->ev_computed[p], 0));
        if (d->exchange == kIpcPull) {
            // every rank's block of this panel sits in its own staging buffer: pull the peers' blocks out of THEIR buffers
            // (same rank-major layout, same slot) into ours, then tell them we are done with their buffer
            MI_TRY(step_barrier(d, d->s_comm, main));
            const size_t blk = (size_t)rows * (size_t)n_loc;
            for (int k = 1; k < d->world; ++k) {
                const int from = (d->rank + k) % d->world;
                HIP_TRY(hipMemcpyAsync(stage + (size_t)from * blk, d->peer_stage[(size_t)from * 2 + (p & 1)] + (size_t)from * blk, blk * sizeof(float),
                                       hipMemcpyDeviceToDevice, d->s_comm));
            }
            MI_TRY(step_barrier(d, d->s_comm, main));
        } else
        MI_TRY(exchange_panel(d, stage, rows, d->s_comm));
        HIP_TRY(hipEventRecord(d->ev_gathered[p], d->s_comm));
        HIP_TRY(hipStreamWaitEvent(d->s_post, d->ev_gathered[p], 0));
        MI_TRY(mi_spmm_unpack_gathered(stage, d_C_full + (size_t)r0 * (size_t)NT, rows, d->world, (int32_t)n_loc, NT, (void *)d->s_post));
        HIP_TRY(hipEventRecord(d->ev_unpacked[p & 1], d->s_post));
        d->unpacked_valid[p & 1] = true;
    }
    if (do_exchange) {
        // the re-layout stream runs the panels in order: its last event covers the step
        const size_t last = d->panels.size() - 1;
        HIP_TRY(hipStreamWaitEvent(main, d->ev_unpacked[last & 1], 0));
        if (d->panels.size() > 1) HIP_TRY(hipStreamWaitEvent(main, d->ev_unpacked[(last - 1) & 1], 0));
    }
    return 0;
}

}  // namespace

extern "C" {

int mi_spmm_dist_create(mi_spmm_dist **out, mi_spmm_handle *h, int32_t num_v, int32_t n_loc, int32_t rank, int32_t world,
                        int32_t n_panels)
{
    if (!out) return MI_SPMM_EINVAL;
    *out = nullptr;
    if (!h || num_v < 0 || n_loc < 0 || world < 1 || rank < 0 || rank >= world || n_panels < 1) return MI_SPMM_EINVAL;
    int64_t prepared = 0, feat = 0, rows = 0;
    MI_TRY(mi_spmm_get_option(h, "prepared", &prepared));
    if (!prepared) return MI_SPMM_ESTATE;
    // the operator must be THIS rank's: n_loc columns, num_v rows (a wider n_loc would exchange columns nobody computed,
    // a smaller num_v would drop rows silently)
    MI_TRY(mi_spmm_get_option(h, "feat", &feat));
    MI_TRY(mi_spmm_get_option(h, "num_v", &rows));
    if (feat != n_loc || rows != num_v) return MI_SPMM_EINVAL;
    mi_spmm_dist *d = new (std::nothrow) mi_spmm_dist();
    if (!d) return MI_SPMM_ENOMEM;
    d->h = h;
    d->M = num_v;
    d->n_loc = n_loc;
    d->rank = rank;
    d->world = world;
    d->n_panels = n_panels;
    d->N_total = (int64_t)world * n_loc;
    make_panels(d);
    *out = d;
    return 0;
}

int mi_spmm_dist_destroy(mi_spmm_dist *d)
{
    if (!d) return 0;
    if (!good(d)) return MI_SPMM_ESTATE;
    (void)hipDeviceSynchronize();
    close_peers(d);
    close_peer_staging(d);
    free_staging(d);
    if (d->d_token) (void)hipFree(d->d_token);
    if (d->comm && d->own_comm) (void)ncclCommDestroy(d->comm);
    for (hipEvent_t e : d->ev_computed) (void)hipEventDestroy(e);
    for (hipEvent_t e : d->ev_gathered) (void)hipEventDestroy(e);
    for (hipEvent_t e : d->ev_pushed) (void)hipEventDestroy(e);
    for (int i = 0; i < 2; ++i) if (d->ev_unpacked[i]) (void)hipEventDestroy(d->ev_unpacked[i]);
    if (d->ev_start) (void)hipEventDestroy(d->ev_start);
    for (hipStream_t s : d->s_push) (void)hipStreamDestroy(s);
    if (d->s_comm) (void)hipStreamDestroy(d->s_comm);
    if (d->s_post) (void)hipStreamDestroy(d->s_post);
    d->magic = 0;
    delete d;
    return 0;
}

int mi_spmm_dist_unique_id(void *id_out)
{
    static_assert(sizeof(ncclUniqueId) == MI_SPMM_DIST_UNIQUE_ID_BYTES, "ncclUniqueId size");
    if (!id_out) return MI_SPMM_EINVAL;
    ncclUniqueId id;
    NCCL_TRY(ncclGetUniqueId(&id));
    std::memcpy(id_out, &id, sizeof(id));
    return 0;
}

int mi_spmm_dist_comm_init(mi_spmm_dist *d, const void *id)
{
    if (!good(d) || !id) return MI_SPMM_EINVAL;
    if (d->comm && d->own_comm) (void)ncclCommDestroy(d->comm);
    d->comm = nullptr;
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    NCCL_TRY(ncclCommInitRank(&d->comm, d->world, uid, d->rank));
    d->own_comm = true;
    return 0;
}

int mi_spmm_dist_set_comm(mi_spmm_dist *d, void *nccl_comm)
{
    if (!good(d)) return MI_SPMM_EINVAL;
    if (d->comm && d->own_comm) (void)ncclCommDestroy(d->comm);
    d->comm = (ncclComm_t)nccl_comm;
    d->own_comm = false;
    return 0;
}

int mi_spmm_dist_export_c(mi_spmm_dist *d, float *d_C_full, void *handle_out, int64_t *offset_out)
{
    static_assert(sizeof(hipIpcMemHandle_t) == MI_SPMM_DIST_IPC_HANDLE_BYTES, "hipIpcMemHandle_t size");
    if (!good(d) || !d_C_full || !handle_out || !offset_out) return MI_SPMM_EINVAL;
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    HIP_TRY(hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)d_C_full));   // the caller's allocator may sub-allocate
    if (!d->ipc_any_size && !ipc_size_ok(size)) return refuse_ipc_size("mi_spmm_dist_export_c", size);
    hipIpcMemHandle_t hnd;
    HIP_TRY(hipIpcGetMemHandle(&hnd, base));
    std::memcpy(handle_out, &hnd, sizeof(hnd));
    *offset_out = (int64_t)((char *)d_C_full - (char *)base);
    return 0;
}

int mi_spmm_dist_set_peers(mi_spmm_dist *d, float *d_C_full, const void *handles, const int64_t *offsets)
{
    if (!good(d) || !d_C_full || !handles || !offsets) return MI_SPMM_EINVAL;
    close_peers(d);
    d->peer_base.assign((size_t)d->world, nullptr);
    d->peer_C.assign((size_t)d->world, nullptr);
    for (int q = 0; q < d->world; ++q) {
        if (q == d->rank) { d->peer_C[(size_t)q] = d_C_full; continue; }
        hipIpcMemHandle_t hnd;
        std::memcpy(&hnd, (const char *)handles + (size_t)q * sizeof(hnd), sizeof(hnd));
        void *base = nullptr;
        hipError_t e = hipIpcOpenMemHandle(&base, hnd, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) { close_peers(d); return (int)e; }
        d->peer_base[(size_t)q] = base;
        d->peer_C[(size_t)q] = (float *)((char *)base + offsets[q]);
    }
    d->exported_C = d_C_full;
    return 0;
}

int mi_spmm_dist_set_host_barrier(mi_spmm_dist *d, mi_spmm_dist_barrier_fn fn, void *ctx)
{
    if (!good(d)) return MI_SPMM_EINVAL;
    d->host_barrier = fn;
    d->host_barrier_ctx = ctx;
    return 0;
}

int mi_spmm_dist_export_staging(mi_spmm_dist *d, void *handles_out, int64_t *offsets_out)
{
    if (!good(d) || !handles_out || !offsets_out) return MI_SPMM_EINVAL;
    MI_TRY(ensure_staging(d));
    for (int b = 0; b < 2; ++b) {
        hipDeviceptr_t base = nullptr;
        size_t size = 0;
        HIP_TRY(hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)d->staging[b]));
        if (!d->ipc_any_size && !ipc_size_ok(size)) return refuse_ipc_size("mi_spmm_dist_export_staging", size);
        hipIpcMemHandle_t hnd;
        HIP_TRY(hipIpcGetMemHandle(&hnd, base));
        std::memcpy((char *)handles_out + (size_t)b * sizeof(hnd), &hnd, sizeof(hnd));
        offsets_out[b] = (int64_t)((char *)d->staging[b] - (char *)base);
    }
    return 0;
}

int mi_spmm_dist_set_peer_staging(mi_spmm_dist *d, const void *handles, const int64_t *offsets)
{
    if (!good(d) || !handles || !offsets) return MI_SPMM_EINVAL;
    MI_TRY(ensure_staging(d));
    close_peer_staging(d);
    d->peer_stage_base.assign((size_t)d->world * 2, nullptr);
    d->peer_stage.assign((size_t)d->world * 2, nullptr);
    for (int q = 0; q < d->world; ++q)
        for (int b = 0; b < 2; ++b) {
            const size_t i = (size_t)q * 2 + b;
            if (q == d->rank) { d->peer_stage[i] = d->staging[b]; continue; }
            hipIpcMemHandle_t hnd;
            std::memcpy(&hnd, (const char *)handles + i * sizeof(hnd), sizeof(hnd));
            void *base = nullptr;
            const hipError_t e = hipIpcOpenMemHandle(&base, hnd, hipIpcMemLazyEnablePeerAccess);
            if (e != hipSuccess) { close_peer_staging(d); return (int)e; }
            d->peer_stage_base[i] = base;
            d->peer_stage[i] = (float *)((char *)base + offsets[i]);
        }
    return 0;
}

int mi_spmm_dist_set_peer_pointers(mi_spmm_dist *d, float *d_C_full, float *const *peer_C_full)
{
    if (!good(d) || !d_C_full || !peer_C_full) return MI_SPMM_EINVAL;
    close_peers(d);
    d->peer_base.assign((size_t)d->world, nullptr);          // nothing of ours to close: the pointers are the caller's
    d->peer_C.assign((size_t)d->world, nullptr);
    for (int q = 0; q < d->world; ++q) {
        float *p = q == d->rank ? d_C_full : peer_C_full[q];
        if (!p) { close_peers(d); return MI_SPMM_EINVAL; }
        d->peer_C[(size_t)q] = p;
    }
    d->exported_C = d_C_full;
    return 0;
}

int mi_spmm_dist_set_option(mi_spmm_dist *d, const char *key, int64_t v)
{
    if (!good(d) || !key) return MI_SPMM_EINVAL;
    const std::string k(key);
    if (k == "exchange") { if (v < 0 || v > 4) return MI_SPMM_EINVAL; d->exchange = (int)v; }
    else if (k == "rehearse") d->rehearse = v != 0;
    else if (k == "external_barrier") d->external_barrier = v != 0;
    else if (k == "ipc_any_size") d->ipc_any_size = v != 0;
    else if (k == "n_panels") {
        if (v < 1 || v > (1 << 20)) return MI_SPMM_EINVAL;
        (void)hipDeviceSynchronize();   // staging buffers of a step in flight
        d->n_panels = (int32_t)v;
        make_panels(d);
        free_staging(d);
        d->unpacked_valid[0] = d->unpacked_valid[1] = false;
    } else return MI_SPMM_EUNSUPPORTED;
    return 0;
}

int mi_spmm_dist_get_option(const mi_spmm_dist *d, const char *key, int64_t *value)
{
    if (!good(d) || !key || !value) return MI_SPMM_EINVAL;
    const std::string k(key);
    const int64_t moved = (int64_t)(d->world - 1) * d->M * (int64_t)d->n_loc * 4;
    if (k == "exchange") *value = d->exchange;
    else if (k == "n_panels") *value = (int64_t)d->panels.size();
    else if (k == "world") *value = d->world;
    else if (k == "rank") *value = d->rank;
    else if (k == "has_comm") *value = d->comm ? 1 : 0;
    else if (k == "external_barrier") *value = d->external_barrier ? 1 : 0;
    else if (k == "ipc_any_size") *value = d->ipc_any_size ? 1 : 0;
    else if (k == "comm_stream_overlaps") *value = d->comm_stream_overlaps;
    else if (k == "post_stream_overlaps") *value = d->post_stream_overlaps;
    else if (k == "has_peers") *value = (int)d->peer_C.size() == d->world ? 1 : 0;
    else if (k == "staging_bytes") *value = (d->exchange == kPeer2D || d->exchange == kPeerStore) ? 0 : (int64_t)(2 * d->staging_elems * sizeof(float));
    else if (k == "bytes_sent_per_step") *value = moved;
    else if (k == "bytes_received_per_step") *value = moved;
    else return MI_SPMM_EUNSUPPORTED;
    return 0;
}

int mi_spmm_dist_run(mi_spmm_dist *d, const float *d_B_loc, float *d_C_full, void *stream)
{
    return step(d, d_B_loc, d_C_full, (hipStream_t)stream, true, true);
}

int mi_spmm_dist_run_compute_only(mi_spmm_dist *d, const float *d_B_loc, float *d_C_full, void *stream)
{
    return step(d, d_B_loc, d_C_full, (hipStream_t)stream, true, false);
}

int mi_spmm_dist_run_exchange_only(mi_spmm_dist *d, float *d_C_full, void *stream)
{
    return step(d, nullptr, d_C_full, (hipStream_t)stream, false, true);
}

// "measure link bandwidth first".  What one link delivers, what all of a GPU's links deliver together, and what kind of links they are --
// the one unknown DESIGN.md's ceiling for the 8-GPU step swings on (76.8 vs 153 GB/s per direction).  Plain device-to-device copies from the front of the
// rank's C_full into the front of each peer's (mapped by set_peers / set_peer_pointers): set-up time, outside any timed region; C_full is scratch afterwards.
int mi_spmm_dist_link_probe(mi_spmm_dist *d, float *d_C_full, int64_t nbytes, const int32_t *peer_device, double *per_peer_gbs,
                            double *all_peers_gbs, int32_t *link_type, int32_t *hops)
{
    if (!good(d) || !d_C_full || !per_peer_gbs || !all_peers_gbs || !link_type || !hops) return MI_SPMM_EINVAL;
    if ((int)d->peer_C.size() != d->world || d->exported_C != d_C_full) return MI_SPMM_ESTATE;      // set_peers first
    const int64_t have = (int64_t)d->M * d->N_total * 4;
    if (nbytes <= 0 || nbytes > have) nbytes = have < ((int64_t)256 << 20) ? have : ((int64_t)256 << 20);
    *all_peers_gbs = 0.0;
    for (int q = 0; q < d->world; ++q) { per_peer_gbs[q] = 0.0; link_type[q] = -1; hops[q] = -1; }
    if (d->world == 1 || nbytes <= 0) return 0;
    MI_TRY(ensure_streams(d));
    MI_TRY(ensure_push_streams(d));
    // topology as the runtime reports it (needs the peers' device ordinals as THIS process sees them; -1 / null: unknown, e.g. one visible device per process)
    int me = -1, ndev = 0;
    HIP_TRY(hipGetDevice(&me));
    HIP_TRY(hipGetDeviceCount(&ndev));
    for (int q = 0; q < d->world && peer_device; ++q) {
        const int pd = peer_device[q];
        if (q == d->rank || pd < 0 || pd >= ndev || pd == me) continue;
        uint32_t lt = 0, hc = 0;
        if (hipExtGetLinkTypeAndHopCount(me, pd, &lt, &hc) == hipSuccess) { link_type[q] = (int32_t)lt; hops[q] = (int32_t)hc; }
        else (void)hipGetLastError();
    }
    // every rank at the same point before each shift: the step's own barrier (communicator all-reduce, else the host's callback), then the host waits for it
    auto align = [&]() -> int {
        if (d->comm) { MI_TRY(device_barrier(d, d->s_comm)); HIP_TRY(hipStreamSynchronize(d->s_comm)); }
        else if (d->host_barrier) d->host_barrier(d->host_barrier_ctx);
        return 0;
    };
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_TRY(hipEventCreate(&e0));
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return MI_SPMM_ENOMEM; }
    int rc = 0;
    auto fail = [&](int code) { rc = code; };
    // one peer at a time: shift k, every rank sends to rank + k while it receives from rank - k (what one link carries in one direction while its reverse
    // direction and every other GPU's links are busy too).  Best of three: the first copy also pays the mapping's first touch.
    for (int k = 1; k < d->world && rc == 0; ++k) {
        const int to = (d->rank + k) % d->world;
        if ((rc = align()) != 0) break;
        float best = 1e30f;
        for (int rep = 0; rep < 3 && rc == 0; ++rep) {
            hipError_t e = hipEventRecord(e0, d->s_push[0]);
            if (e == hipSuccess) e = hipMemcpyAsync(d->peer_C[(size_t)to], d_C_full, (size_t)nbytes, hipMemcpyDeviceToDevice, d->s_push[0]);
            if (e == hipSuccess) e = hipEventRecord(e1, d->s_push[0]);
            if (e == hipSuccess) e = hipEventSynchronize(e1);
            float ms = 0.f;
            if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
            if (e != hipSuccess) { fail((int)e); break; }
            if (ms < best) best = ms;
        }
        if (rc == 0 && best > 0.f) per_peer_gbs[to] = (double)nbytes / ((double)best * 1e-3) / 1e9;
    }
    // all peers at once, one stream per peer (the peer2d step's streams): what the GPU's links deliver together, outbound
    if (rc == 0 && (rc = align()) == 0) {
        float best = 1e30f;
        for (int rep = 0; rep < 3 && rc == 0; ++rep) {
            hipError_t e = hipEventRecord(e0, d->s_comm);
            for (int k = 1; k < d->world && e == hipSuccess; ++k) {
                const int to = (d->rank + k) % d->world;
                hipStream_t s = d->s_push[(size_t)(k - 1)];
                e = hipStreamWaitEvent(s, e0, 0);
                if (e == hipSuccess) e = hipMemcpyAsync(d->peer_C[(size_t)to], d_C_full, (size_t)nbytes, hipMemcpyDeviceToDevice, s);
                if (e == hipSuccess) e = hipEventRecord(d->ev_pushed[(size_t)(k - 1)], s);
                if (e == hipSuccess) e = hipStreamWaitEvent(d->s_comm, d->ev_pushed[(size_t)(k - 1)], 0);
            }
            if (e == hipSuccess) e = hipEventRecord(e1, d->s_comm);
            if (e == hipSuccess) e = hipEventSynchronize(e1);
            float ms = 0.f;
            if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
            if (e != hipSuccess) { fail((int)e); break; }
            if (ms < best) best = ms;
        }
        if (rc == 0 && best > 0.f) *all_peers_gbs = (double)(d->world - 1) * (double)nbytes / ((double)best * 1e-3) / 1e9;
    }
    if (rc == 0) rc = align();
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

int64_t mi_spmm_dist_ipc_exportable_bytes(int64_t nbytes)
{
    if (nbytes < 0) return MI_SPMM_EINVAL;
    const uint64_t n = (uint64_t)nbytes;
    return ((n >> 31) & 1u) ? (int64_t)(((n >> 32) + 1) << 32) : nbytes;      // bit 31 set: up to the next multiple of 4 GiB
}

const char *mi_spmm_dist_strerror(int code)
{
    if (code <= MI_SPMM_DIST_ENCCL_BASE) return ncclGetErrorString((ncclResult_t)(MI_SPMM_DIST_ENCCL_BASE - code));
    return mi_spmm_strerror(code);
}

}  // extern "C"
