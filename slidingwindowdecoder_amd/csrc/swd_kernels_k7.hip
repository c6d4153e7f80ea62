// Instantiations of swd::pipeline_kernel for kind 7 (bpgdg_decoder(multi_thread=True): the reference's threaded ensemble,
// swd_gdg_kernel.h: gdg_ensemble_tree) and their launchers; its own translation unit so that the serial and the parallel
// tree-walk kernels (kinds 1, 2) keep their register allocation.
#include "swd_plan.h"
#include "swd_variants.h"

namespace swd {
#define SWD_IF_0(...)
#define SWD_IF_1(...) __VA_ARGS__
#define SWD_IF(c, ...) SWD_IF_##c(__VA_ARGS__)
#define X(nt, vf, dm, kg, sf, k1, k2, k3) SWD_IF(k1, SWD_DEFINE_GDG_LAUNCHER(7, nt, vf, dm, kg, sf))
SWD_VARIANTS(X)
#undef X
} // namespace swd
