// kernels_fast_gen_cols_f64.hip -- instantiates the generated fp64 configurations
// (tools/gen_fast_table.py -> fast_table_gen_cols_f64.inc).
#include "fast_table.h"

namespace mifft {

static const FastEntry kGenColsF64Table[] = {
#include "fast_table_gen_cols_f64.inc"
};

const FastEntry* gen_cols_f64_table(int* count) {
    *count = (int)(sizeof(kGenColsF64Table) / sizeof(kGenColsF64Table[0]));
    return kGenColsF64Table;
}

}  // namespace mifft
