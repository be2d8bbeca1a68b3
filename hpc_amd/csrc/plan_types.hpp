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

}  // namespace mi
