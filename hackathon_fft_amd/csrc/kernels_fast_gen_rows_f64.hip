// kernels_fast_gen_rows_f64.hip -- instantiates the generated fp64 configurations
// (tools/gen_fast_table.py -> fast_table_gen_rows_f64.inc).
#include "fast_table.h"

namespace mifft {

static const FastEntry kGenRowsF64Table[] = {
#include "fast_table_gen_rows_f64.inc"
};

const FastEntry* gen_rows_f64_table(int* count) {
    *count = (int)(sizeof(kGenRowsF64Table) / sizeof(kGenRowsF64Table[0]));
    return kGenRowsF64Table;
}

}  // namespace mifft
