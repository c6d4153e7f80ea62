// Large-graph form of the guessing decoders' kernels (scratch region of the layout -- messages, sort keys -- in HBM): kind 8 = the
// serial tree walk (kind 1), kind 9 = the threaded ensemble (kind 7) on window-major tickets.  A functional path for graphs whose
// messages do not fit a CU's LDS next to the guessing decoders' state (the reference's [[288,12,18]] (4,1) windows under
// bpgdg_decoder, `Sliding Window GDG.ipynb` cell 8), one launcher per variant of SWD_BIG_VARIANTS.
#include "swd_plan.h"
#include "swd_variants.h"

namespace swd {
#define X(nt, vf, dm, kg) SWD_DEFINE_BIG_GDG_LAUNCHER(8, nt, vf, dm, kg) SWD_DEFINE_BIG_GDG_LAUNCHER(9, nt, vf, dm, kg)
SWD_BIG_VARIANTS(X)
#undef X
} // namespace swd
