// Large-graph form of the osd_window kernels (scratch region of the layout in HBM): explicit instantiations, one launcher per
// variant of SWD_BIG_VARIANTS.  Kind 5 keeps the posterior history in its HBM ring, kind 6 accumulates its sum in registers.
#define SWD_OSDW_TUNED 1
#include "swd_plan.h"
#include "swd_variants.h"

namespace swd {
#define X(nt, vf, dm, kg) SWD_DEFINE_BIG_LAUNCHER(5, nt, vf, dm, kg) SWD_DEFINE_BIG_LAUNCHER(6, nt, vf, dm, kg)
SWD_BIG_VARIANTS(X)
#undef X
} // namespace swd
