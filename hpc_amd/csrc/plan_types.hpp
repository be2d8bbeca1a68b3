// plan_types.hpp -- plain structs shared by the kernels and the two plan builders.
#pragma once
#include <stdint.h>

namespace mi {

constexpr int kBlockThreads = 256;  // 4 waves; 8 such blocks fill a CU's 32 wave slots

struct Chunk {              // one segment of one row, handled by one lane group
    int32_t beg;            // first nonzero (index into col_idx/vals)
    int32_t end;            // one past the last
    int32_t slot;           // >= 0: row of the partial-sum workspace it writes (piece of a split row)
                            //  -1: the segment starts the row's chain (from +0) -> result goes straight to C[row]
                            //  -2 (kSlotContinue): a column strip's sub-segment that continues the chain from C[row]
    int32_t row;            // CSR row it belongs to
};
constexpr int32_t kSlotContinue = -2;

struct LongRow {            // a row longer than the split threshold (a hub)
    int32_t row;
    int32_t first_slot;     // split mode: its pieces' partial sums are slots [first_slot, first_slot + n_chunks); hub mode: -1
    int32_t n_chunks;       // split mode: number of pieces; hub mode (exact order, spmm_hub): 0
    int32_t len;            // nonzeros in the row
};

// ---- extra destinations of every finished C row segment (multi-GPU "peer_store" exchange: the peers' C, mapped through HIP IPC)
constexpr int kMaxPeerOut = 7;
struct PeerOut {
    float *p[kMaxPeerOut];      // same layout as the local C (same pitch, same column offset already applied)
    int32_t n;                  // 0 (every single-GPU call) .. kMaxPeerOut
    int32_t pad;
};

// ---- block (MFMA) path ---------------------------------------------------------------------------
constexpr int kMaxPieces = 4;     // most pieces (= passes) a group's column list is cut into
constexpr int kShareLenUnit = 32; // a run piece is shared only if its length is a multiple of this (two of the kernels' largest k batches)

struct GroupPieces {              // analyze_group_runs output, one per qualifying group (64 bytes)
    int32_t n;                    // pieces: 1..kMaxPieces
    int32_t p0;                   // row_ptr[16 g]
    int32_t row_len;              // nonzeros per row of the group
    int32_t pad;
    int32_t k0[kMaxPieces];       // position of the piece inside the group's column list
    int32_t c0[kMaxPieces];       // >= 0: first column of a run of consecutive columns; < 0: -1 - first column of a plain list piece
    int32_t len[kMaxPieces];
};

enum : int32_t { kPieceCarryIn = 1, kPieceCarryOut = 2 };

constexpr int kMaxShare = 2;      // most pieces per item = widest block kernel instantiated (spmm_block_items<.., G, ..>)

struct BlockPiece {               // one piece of one group, as the kernel sees it (24 bytes)
    int32_t group;                // 16-row group index (rows 16*group .. 16*group+15)
    int32_t k0;                   // first position of the piece inside each row
    int32_t len;                  // nonzeros per row in the piece
    int32_t flags;                // kPieceCarryIn: the chain continues from C; kPieceCarryOut: a later pass continues it
    int32_t p0;                   // row_ptr[16*group]: the group's 16 rows are p0 + i*row_len (equal lengths: they qualified)
    int32_t row_len;              // nonzeros per row of the group (the whole list, not the piece)
};

struct BlockItem {                // what one wave sweeps: up to kMaxShare pieces sharing their B rows, longest first.
    int32_t m;                    // pieces                       One 64-byte record = one memory round trip before
    int32_t c0;                   // run items: B rows c0, c0+1, ...; list items: < 0        the first B row is requested.
    BlockPiece p[kMaxShare];
    int32_t pad[2];
};
static_assert(sizeof(BlockItem) == 64, "one item = one 64-byte record");

}  // namespace mi
