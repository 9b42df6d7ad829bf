// kernels_fast_gen_cols.hip -- instantiates the generated cols configurations
// (tools/gen_fast_table.py -> fast_table_gen_cols.inc).
#include "fast_table.h"

namespace mifft {

static const FastEntry kGenColsTable[] = {
#include "fast_table_gen_cols.inc"
};

const FastEntry* gen_cols_table(int* count) {
    *count = (int)(sizeof(kGenColsTable) / sizeof(kGenColsTable[0]));
    return kGenColsTable;
}

}  // namespace mifft
