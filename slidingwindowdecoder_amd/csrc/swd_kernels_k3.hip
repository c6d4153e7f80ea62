// Instantiations of swd::pipeline_kernel for kind 3 (osd_window windows, posterior history accumulated in registers) and
// their launchers (swd_plan.h); one translation unit per kind so that the kernels compile in parallel.
#ifndef SWD_POST_DEPTH2
#define SWD_POST_DEPTH2 3 // depth-2 register cache for the shortened graph (swd_osdw_kernel.h): bit 0 the 1024-thread kernels, bit 1 those of up to 256 threads
#endif
#define SWD_OSDW_TUNED 1 // packed register caches, three waves per SIMD where they fit (swd_osdw_kernel.h)
#include "swd_plan.h"
#include "swd_variants.h"

namespace swd {
#define SWD_IF_0(...)
#define SWD_IF_1(...) __VA_ARGS__
#define SWD_IF(c, ...) SWD_IF_##c(__VA_ARGS__)
#define X(nt, vf, dm, kg, sf, k1, k2, k3) SWD_IF(k3, SWD_DEFINE_LAUNCHER(3, nt, vf, dm, kg, sf))
SWD_VARIANTS(X)
#undef X
} // namespace swd
