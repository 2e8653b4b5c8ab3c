// What pinning host memory costs (run on the GPU box): hipHostMalloc and hipHostRegister of the sizes the command line handles
// (a decoded slide region, the feature rows), against the time the same bytes take over the link pageable.
//   hipcc --offload-arch=gfx950 -O2 scripts/pin_probe.hip -o /tmp/pin_probe && /tmp/pin_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipFree(0);
    for (size_t mb : {64, 256, 1024, 4096}) {
        const size_t n = mb << 20;
        void *p = nullptr, *d = nullptr;
        hipMalloc(&d, n);
        double t = now();
        hipHostMalloc(&p, n, hipHostMallocDefault);
        const double t_alloc = now() - t;
        t = now(); memset(p, 1, n); const double t_touch = now() - t;
        t = now(); hipMemcpy(d, p, n, hipMemcpyHostToDevice); const double t_h2d_pinned = now() - t;
        t = now(); hipHostFree(p); const double t_free = now() - t;
        char *q = (char *)malloc(n);
        memset(q, 2, n);
        t = now(); hipMemcpy(d, q, n, hipMemcpyHostToDevice); const double t_h2d_pageable = now() - t;
        t = now(); hipHostRegister(q, n, hipHostRegisterDefault); const double t_reg = now() - t;
        t = now(); hipMemcpy(d, q, n, hipMemcpyHostToDevice); const double t_h2d_reg = now() - t;
        t = now(); hipHostUnregister(q); const double t_unreg = now() - t;
        printf("%5zu MB: hipHostMalloc %7.1f ms (%5.1f GB/s), first touch %6.1f ms, free %6.1f ms | hipHostRegister %7.1f ms (%5.1f GB/s), unregister %6.1f ms | "
               "H2D pinned %6.1f ms, registered %6.1f ms, pageable %6.1f ms\n", mb, 1e3 * t_alloc, n / t_alloc / 1e9, 1e3 * t_touch, 1e3 * t_free, 1e3 * t_reg,
               n / t_reg / 1e9, 1e3 * t_unreg, 1e3 * t_h2d_pinned, 1e3 * t_h2d_reg, 1e3 * t_h2d_pageable);
        free(q);
        hipFree(d);
    }
    return 0;
}
