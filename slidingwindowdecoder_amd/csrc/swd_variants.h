// Kernel variants: threads per shot, variable nodes per thread, column-degree bound, groups of four row positions,
// sf (the full-graph phase shares heavy checks among threads: needs the host-built map, 4 * kg may be < K),
// then which further kernels exist for the variant: guessing decoders in serial form (kind 1) and parallel form
// (kind 2), osd_window with the posterior history accumulated in registers (kind 3).
// A plan uses the first variant with NT >= m, NT * VF >= n, DM >= D, 4 * KG >= K over all its windows.
// X(nt, vf, dm, kg, sf, has_k1, has_k2, has_k3)
#pragma once
#if defined(SWD_HEADLINE_ONLY) && defined(SWD_EXP512) // experiment: the [[144,12,12]] windows on 512 threads, heavy checks shared by two lanes in both phases
#define SWD_VARIANTS(X) X(512, 4, 6, 6, 1, 0, 0, 1) X(256, 7, 6, 9, 0, 1, 1, 1)
#elif defined(SWD_HEADLINE_ONLY) // development builds: only the [[144,12,12]] kernels
#define SWD_VARIANTS(X) X(256, 7, 6, 9, 0, 1, 1, 1)
#elif defined(SWD_V7816_ONLY) // development builds: the <256, 7, 8, 16> kernels
#define SWD_VARIANTS(X) X(256, 7, 8, 16, 0, 1, 1, 0)
#elif defined(SWD_EXP_288_512) // experiment (round 5): [[288,12,18]] windows of up to 512 checks ((3,1): 432 x 3456) on 512 threads / 256 VGPRs
#define SWD_VARIANTS(X) X(512, 8, 6, 9, 0, 0, 0, 1) X(1024, 5, 6, 6, 1, 0, 0, 1) X(1024, 5, 6, 9, 0, 1, 0, 1)
#elif defined(SWD_EXP_288_640) // experiment (round 5): the [[288,12,18]] (4,1) windows (576 checks) on 640 threads = ten waves, one check per thread, 168 VGPRs
#define SWD_VARIANTS(X) X(640, 8, 6, 9, 0, 0, 0, 1) X(1024, 5, 6, 6, 1, 0, 0, 1) X(1024, 5, 6, 9, 0, 1, 0, 1)
#elif defined(SWD_BB288_ONLY) // development builds: only the [[288,12,18]] kernels
#define SWD_VARIANTS(X) X(1024, 5, 6, 6, 1, 0, 0, 1) X(1024, 5, 6, 9, 0, 1, 0, 1)
#else
#define SWD_VARIANTS(X)                                                                                              \
    X(64, 4, 4, 2, 0, 1, 1, 0)      /* small codes, e.g. [[72,12,6]] hx (n=72, D=3, K=6) */                             \
    X(64, 4, 8, 16, 0, 1, 1, 0)                                                                                         \
    X(256, 2, 8, 16, 0, 1, 1, 0)                                                                                        \
    X(256, 4, 8, 16, 0, 1, 1, 0)                                                                                        \
    X(256, 2, 10, 12, 0, 1, 1, 1)   /* SHYPS r=3 circuit-level windows (63 x 476, column weight <= 9) */                \
    X(256, 7, 6, 9, 0, 1, 1, 1)     /* [[144,12,12]] circuit-level windows */                                           \
    X(256, 7, 8, 16, 0, 1, 1, 0)                                                                                        \
    X(1024, 5, 6, 6, 1, 0, 0, 1)    /* [[288,12,18]] circuit-level windows, osd_window */                               \
    X(1024, 5, 6, 9, 0, 1, 0, 1)    /* [[288,12,18]] circuit-level windows */                                           \
    X(1024, 3, 10, 12, 0, 1, 0, 0)  /* SHYPS r=3 twelve-round windows (252 x 2240, column weight 9, row weight 44) */   \
    X(1024, 8, 8, 16, 0, 1, 0, 0)
#endif

// Large graphs (row J of the scope table: osd_window on matrices whose fp64 messages do not fit a CU's LDS, e.g. the un-windowed
// 936 x 8784 detector error model of /root/reference/IBM.ipynb:119): one 1024-thread workgroup per shot, the scratch region of
// the layout in HBM (pipeline_kernel<..., BIG>).  Chosen when no variant above fits.  X(nt, vf, dm, kg)
#define SWD_BIG_VARIANTS(X)                                                                                           \
    X(1024, 9, 6, 9)     /* [[144,12,12]] global DEM: 936 x 8784, column weight 6, row weight 35 */                      \
    X(1024, 9, 10, 16)   /* up to 1024 checks, 9216 columns, column weight 10, row weight 64 */
