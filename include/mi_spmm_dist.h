/*
 * mi_spmm_dist.h -- C ABI of the column-sharded multi-GPU step (hpc_amd/libmi_spmm_dist.so).
 *
 *     C[:, g*n : (g+1)*n] = A * B[:, g*n : (g+1)*n]   on GPU g of G,   then every rank holds row-major C[M][G*n].
 *
 * The reference has no multi-GPU code: its only trace is the commented-out `extern ncclComm_t comm;` of
 * PA4/workspace/include/util.h:30.  This library is what that line would have grown into -- the exchange of
 * the C column blocks over RCCL / xGMI (BASELINE.json north_star) behind plain C entry points, so
 * that the reference's own C++ harness (one process per GPU, MPI or any launcher) reaches the N > 1 path:
 * no torch, no Python in the signatures.  It is a separate library: libmi_spmm.so (the operator) stays free
 * of any communication dependency; this one links librccl.
 *
 * One process per GPU.  A (the CSR arrays of the mi_spmm handle) is replicated, the rank's B slice is a
 * contiguous K x n_loc array, C_full is the caller's row-major M x (world*n_loc) array on every rank.
 * A step is pipelined by row panels over three streams (compute / exchange / re-layout).
 *
 * Exchange schedules ("exchange" option):
 *   0  allgather  ncclAllGather of the panel's column blocks into rank-major staging, then a re-layout
 *                 kernel into C_full (an all-gather concatenates contiguous per-rank buffers).
 *                 The rank computes its own block straight into its staging slot (in-place all-gather).
 *   1  direct     the same staging layout filled by ONE grouped launch of world-1 ncclSend of the rank's block and
 *                 world-1 ncclRecv straight into staging[peer]: on a fully connected node every pair has its own
 *                 xGMI link, so all links of a GPU carry one block each at the same time.
 *   2  peer2d     no staging, no re-layout: the rank computes its block straight into ITS C_full (row pitch
 *                 world*n_loc) and pushes the panel into every peer's C_full with a strided 2-D device-to-device
 *                 copy (512-byte row segments at pitch 4*N_total; the peers' C_full are mapped through HIP IPC).
 *                 Per step and GPU this moves (world-1)/world of C once out and once in, and nothing else.
 *   3  peer_store the kernels' epilogues store every finished row segment into the local C_full AND into every peer's
 *                 (mi_spmm_run_rows_multi; the peers' C_full mapped through HIP IPC as for peer2d): no copy engine, no
 *                 staging, no re-layout -- the remote stores never touch local memory, so per step and GPU the local
 *                 memory system carries the gather (18.0 GB at C3) plus what the peers store INTO it (3.76 GB) and nothing
 *                 else: the only schedule whose byte budget allows the north star's >= 6x at 8 GPUs.
 *                 One launch set over all rows between two barriers; world <= 8.
 *   4  ipc_pull   the allgather schedule's staging layout, double buffering and re-layout kernel with a different
 *                 transport: every rank computes its block into its own staging buffer and PULLS the peers' blocks out of
 *                 their staging buffers (mapped through HIP IPC: mi_spmm_dist_export_staging / _set_peer_staging) with
 *                 device-to-device copies; two cross-rank barriers per panel (blocks computed / blocks pulled) -- the
 *                 communicator's all-reduce, or mi_spmm_dist_set_host_barrier's callback.  Exists so that the staging path
 *                 of a step runs with several real ranks where RCCL cannot (ranks sharing one GPU); not a fast path.
 * 0 and 1 need a communicator (mi_spmm_dist_comm_init); 2 and 3 need the peers' C_full (mi_spmm_dist_set_peers) and,
 * to be self-synchronising, a communicator too (a one-element all-reduce is the end-of-step barrier).  Without a
 * communicator the two barriers go through the host callback of mi_spmm_dist_set_host_barrier when one is registered
 * (in-process and shared-GPU drivers: the step still synchronises itself); with neither, a peer2d / peer_store step at
 * world > 1 is refused (MI_SPMM_ESTATE) unless "external_barrier" = 1 says that the caller brackets every step with a
 * cross-rank barrier of its own (after the previous step's consumers, and after the step's stream work has completed).
 *
 * Errors: 0 = ok; negative MI_SPMM_E* codes of mi_spmm.h; positive hipError_t; ncclResult_t r is returned as
 * MI_SPMM_DIST_ENCCL_BASE - r.  Never aborts.
 */
#ifndef MI_SPMM_DIST_H
#define MI_SPMM_DIST_H

#include <stdint.h>

#include "mi_spmm.h"

#ifdef __cplusplus
extern "C" {
#endif

#define MI_SPMM_DIST_ENCCL_BASE (-1000)
#define MI_SPMM_DIST_UNIQUE_ID_BYTES 128   /* sizeof(ncclUniqueId) */
#define MI_SPMM_DIST_IPC_HANDLE_BYTES 64   /* sizeof(hipIpcMemHandle_t) */

typedef struct mi_spmm_dist mi_spmm_dist;

/* h: a PREPROCESSED operator for this rank's n_loc = feat_in columns and num_v rows (mi_spmm_preprocess done; a handle
 * of another shape is MI_SPMM_EINVAL); it stays owned by the caller and must outlive the object.  n_panels: row panels per step (>= 1; panels are multiples of 256 rows). */
int mi_spmm_dist_create(mi_spmm_dist **out, mi_spmm_handle *h, int32_t num_v, int32_t n_loc, int32_t rank,
                        int32_t world, int32_t n_panels);
int mi_spmm_dist_destroy(mi_spmm_dist *d);

/* Communicator bootstrap, the usual RCCL way: rank 0 fills 128 bytes (ncclGetUniqueId), the host launcher
 * broadcasts them (MPI_Bcast, a file, torch.distributed ...), every rank calls comm_init (collective).
 * mi_spmm_dist_set_comm adopts a communicator the host already has instead (ncclComm_t; not destroyed by us). */
int mi_spmm_dist_unique_id(void *id_out);
int mi_spmm_dist_comm_init(mi_spmm_dist *d, const void *id);
int mi_spmm_dist_set_comm(mi_spmm_dist *d, void *nccl_comm);

/* peer2d: export this rank's C_full (64-byte HIP IPC handle + the byte offset of d_C_full inside its
 * allocation), all-gather handles and offsets on the host side, hand the tables in.  handles: world x 64 bytes,
 * offsets: world x int64; the rank's own entry is ignored.  Re-call when C_full changes. */
int mi_spmm_dist_export_c(mi_spmm_dist *d, float *d_C_full, void *handle_out, int64_t *offset_out);

/* Allocation sizes and HIP IPC.  hipIpcOpenMemHandle does not return for an allocation whose size has bit 31 set (2 GiB <= size
 * mod 4 GiB; measured in dmabuf IPC mode, round 4: 2 and 3 GiB hang in the importer, 1, 1.5, 4 and 5 GiB open at once), and an
 * exporter cannot un-hang its peers: mi_spmm_dist_export_c / _export_staging return MI_SPMM_EUNSUPPORTED for such an allocation
 * (the one d_C_full lies in: hipMemGetAddressRange) unless "ipc_any_size" is set.  A host that wants peer2d / peer_store with
 * M x N_total x 4 in that range (C1 on four GPUs: exactly 2 GiB) allocates C_full with this many bytes instead:
 * nbytes itself when it can be exported, else the next multiple of 4 GiB.  The library's own staging buffers are sized that way. */
int64_t mi_spmm_dist_ipc_exportable_bytes(int64_t nbytes);
int mi_spmm_dist_set_peers(mi_spmm_dist *d, float *d_C_full, const void *handles, const int64_t *offsets);

/* Link probe ("measure link bandwidth first"; no reference counterpart).  After set_peers / set_peer_pointers, at set-up time, COLLECTIVE
 * (every rank calls it; the shifts are separated by the step's own barrier: the communicator's all-reduce, else the host barrier callback).  Copies nbytes
 * (<= 0 or too large: min(256 MiB, the size of C_full)) device-to-device from the front of this rank's C_full into the front of a peer's:
 *   per_peer_gbs[world]  GB/s to peer q, measured while every rank sends to rank + k and receives from rank - k (one link, both directions busy); own entry 0
 *   *all_peers_gbs       GB/s out of this GPU with all world - 1 copies in flight at once (one stream per peer)
 *   link_type[world], hops[world]   hipExtGetLinkTypeAndHopCount(this device, peer_device[q]) (HSA_AMD_LINK_INFO_TYPE_*: 2 = PCIe, 4 = xGMI); -1 = unknown.
 *                        peer_device (world entries, may be NULL): rank q's device ordinal as THIS process sees it, -1 if it does not.
 * C_full holds junk afterwards (it is the step's output buffer: run a step before reading it).  A probe that hangs is the caller's watchdog's to catch. */
int mi_spmm_dist_link_probe(mi_spmm_dist *d, float *d_C_full, int64_t nbytes, const int32_t *peer_device, double *per_peer_gbs,
                            double *all_peers_gbs, int32_t *link_type, int32_t *hops);

/* The same table for a host that drives several ranks from ONE process (one thread or object per GPU, peer access enabled by
 * the host): the peers' C_full as plain device pointers, no IPC.  peer_C_full: world pointers, the rank's own entry ignored. */
int mi_spmm_dist_set_peer_pointers(mi_spmm_dist *d, float *d_C_full, float *const *peer_C_full);

/* ipc_pull: export this rank's two staging buffers (2 x 64-byte handles, 2 offsets), all-gather them on the host side, hand
 * the tables in (world x 2 handles, world x 2 offsets, rank-major).  Re-do both after "n_panels" changes. */
int mi_spmm_dist_export_staging(mi_spmm_dist *d, void *handles_out, int64_t *offsets_out);
int mi_spmm_dist_set_peer_staging(mi_spmm_dist *d, const void *handles, const int64_t *offsets);

/* A cross-rank barrier supplied by the host (MPI_Barrier, a gloo barrier ...).  Used where a step needs every rank at the
 * same point and there is no communicator: the library drains its streams, then calls fn(ctx).  Serialises the step on
 * the host: for rehearsals and hosts without RCCL, not for speed.  fn == NULL removes it. */
typedef void (*mi_spmm_dist_barrier_fn)(void *ctx);
int mi_spmm_dist_set_host_barrier(mi_spmm_dist *d, mi_spmm_dist_barrier_fn fn, void *ctx);

/* keys: "exchange" (0 allgather, 1 direct, 2 peer2d, 3 peer_store, 4 ipc_pull), "n_panels", "rehearse" (1: run the staging / collective /
 * re-layout machinery even at world == 1 -- the one-GPU rehearsal of the N > 1 path), "external_barrier" (see above), "ipc_any_size" (1: export allocations whatever their size, see mi_spmm_dist_ipc_exportable_bytes); read-only: "world", "rank", "has_comm",
 * "has_peers", "staging_bytes", "bytes_sent_per_step", "bytes_received_per_step", "comm_stream_overlaps" / "post_stream_overlaps" (1: the exchange /
 * re-layout stream was tested to run beside the compute stream -- mi_spmm_stream_create_concurrent, mi_spmm.h; -1 before the first step) */
int mi_spmm_dist_set_option(mi_spmm_dist *d, const char *key, int64_t value);
int mi_spmm_dist_get_option(const mi_spmm_dist *d, const char *key, int64_t *value);

/* One step.  d_B_loc: K x n_loc (pitch n_loc); d_C_full: M x (world*n_loc) (pitch world*n_loc), the same
 * pointer that was exported for peer2d.  Asynchronous: everything is ordered after the work already on `stream`,
 * and `stream` waits for the step's exchange at the end.  world == 1: the block IS C. */
int mi_spmm_dist_run(mi_spmm_dist *d, const float *d_B_loc, float *d_C_full, void *stream);

/* The two legs on their own, for the reported breakdown (bench.py): compute-only writes the rank's block into its
 * staging slot / C_full; exchange-only moves whatever the block holds. */
int mi_spmm_dist_run_compute_only(mi_spmm_dist *d, const float *d_B_loc, float *d_C_full, void *stream);
int mi_spmm_dist_run_exchange_only(mi_spmm_dist *d, float *d_C_full, void *stream);

const char *mi_spmm_dist_strerror(int code);

#ifdef __cplusplus
}
#endif
#endif /* MI_SPMM_DIST_H */
