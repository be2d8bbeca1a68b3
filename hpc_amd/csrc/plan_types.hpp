// plan_types.hpp -- plain structs shared by the kernels and the two plan builders.
#pragma once
#include <stdint.h>

namespace mi {

constexpr int kBlockThreads = 256;  // 4 waves; 8 such blocks fill a CU's 32 wave slots

struct Chunk {              // one segment of one row, handled by one lane group
    int32_t beg;            // first nonzero (index into col_idx/vals)
    int32_t end;            // one past the last
    int32_t slot;           // >= 0: row of the partial-sum workspace it writes (piece of a split row)
                            //  < 0: the segment is the WHOLE row -> result goes straight to C[row]
    int32_t row;            // CSR row it belongs to
};

struct LongRow {
    int32_t row;
    int32_t first_slot;
    int32_t n_chunks;
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

// ---- block path, B-stationary sweeps (spmm_block_sweep) ----------------------------------------------
// The host lays the run pieces on tracks and writes out, trip by trip, what every track of every workgroup does: the
// kernel is an interpreter of that table (no piece bookkeeping on the device).
constexpr int kSweepTracks = 6;   // tracks per workgroup = accumulator sets per wave (each wave: all tracks, one 64-column quarter of the slab)
constexpr int kSweepAhead = 3;    // trips the B loads run ahead of the MFMAs (register sets)
constexpr int kSweepTrip = 16;    // columns (B rows) per trip; sweepable pieces start and end on multiples of it

constexpr int kSweepFirst = 1;    // the piece's first trip: accumulators start from +0 or from the carried tile
constexpr int kSweepLast = 2;     // its last trip: the tile is stored (final rows, or the chain a later pass continues)
constexpr int kSweepCarryIn = 4;
constexpr int kSweepCarryOut = 8;

struct SweepEnt {                 // one track during one trip (16 bytes; lane l of a wave loads track l's entry)
    int32_t group;                // block group (rows 16*group ..) of the piece under this trip's columns; -1: none
    int32_t flags;                // kSweep*
    int32_t a_off;                // vals index of (first row of the group, this trip's first column)
    int32_t row_len;              // nonzeros per row of the group (row i of the group: a_off + i * row_len)
};
static_assert(sizeof(SweepEnt) == 16, "sweep entry");

struct SweepWG { int32_t trip_begin, n_trips; };   // a workgroup's trips in the trip tables (cols[], ents[][kSweepTracks])

}  // namespace mi
