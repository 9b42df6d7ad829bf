// kernels_fast.hip -- specialised register-butterfly kernel families (gfx950).
#include "mifft_internal.h"

namespace mifft {

bool select_fast(const Plan& plan, DimPass& pass) {
    (void)plan;
    (void)pass;
    return false;
}

}  // namespace mifft
