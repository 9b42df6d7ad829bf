// kernels_fast_gen_rows.hip -- instantiates the generated rows configurations
// (tools/gen_fast_table.py -> fast_table_gen_rows.inc).
#include "fast_table.h"

namespace mifft {

static const FastEntry kGenRowsTable[] = {
#include "fast_table_gen_rows.inc"
};

const FastEntry* gen_rows_table(int* count) {
    *count = (int)(sizeof(kGenRowsTable) / sizeof(kGenRowsTable[0]));
    return kGenRowsTable;
}

}  // namespace mifft
