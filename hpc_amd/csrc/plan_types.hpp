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

struct GroupPieces {              // analyze_group_runs output, one per qualifying group (64 bytes)
    int32_t n;                    // pieces: 1..kMaxPieces
    int32_t pad[3];
    int32_t k0[kMaxPieces];       // position of the piece inside the group's column list
    int32_t c0[kMaxPieces];       // >= 0: first column of a run of consecutive columns; < 0: -1 - first column of a plain list piece
    int32_t len[kMaxPieces];
};

enum : int32_t { kPieceCarryIn = 1, kPieceCarryOut = 2 };

struct BlockPiece {               // one piece of one group, as the kernel sees it
    int32_t group;                // 16-row group index (rows 16*group .. 16*group+15)
    int32_t k0;                   // first position of the piece inside each row
    int32_t len;                  // nonzeros per row in the piece
    int32_t flags;                // kPieceCarryIn: the chain continues from C; kPieceCarryOut: a later pass continues it
};

struct BlockItem {                // what one wave sweeps: up to G pieces sharing their B rows (longest first)
    int32_t first;                // index of its first piece in the pieces array
    int32_t m;                    // pieces
    int32_t c0;                   // >= 0: B rows c0, c0+1, ...; < 0: the columns are read from col_idx (m == 1)
    int32_t len_max;              // length of the longest (= first) piece
};

}  // namespace mi
