// Instantiations of swd::pipeline_kernel for kind 1 (guessing decoders, serial tree walk) and
// their launchers (swd_plan.h); one translation unit per kind so that the kernels compile in parallel.
#include "swd_plan.h"
#include "swd_variants.h"

namespace swd {
#define SWD_IF_0(...)
#define SWD_IF_1(...) __VA_ARGS__
#define SWD_IF(c, ...) SWD_IF_##c(__VA_ARGS__)
#define X(nt, vf, dm, kg, sf, k1, k2, k3) SWD_IF(k1, SWD_DEFINE_GDG_LAUNCHER(1, nt, vf, dm, kg, sf))
SWD_VARIANTS(X)
#undef X
} // namespace swd
