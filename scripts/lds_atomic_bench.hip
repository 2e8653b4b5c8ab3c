// LDS atomic throughput on gfx950: cycles per wave-instruction per CU for reads, non-returning adds, returning adds, cmpswap,
// conflict-free and with same-address multiplicity.  One block per CU, W waves per block, all issuing back to back.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE, int MULT>
__global__ void k(unsigned *out, long long *cyc, int iters)
{
    __shared__ unsigned s[8192];
    const int t = threadIdx.x, lane = t & 63;
    for (int i = t; i < 8192; i += blockDim.x) s[i] = 0;
    __syncthreads();
    // address pattern: within a wave, lanes map to distinct banks/addresses; MULT lanes share one address
    unsigned base = ((t >> 6) * 64) & 8191;
    unsigned a = (base + (lane / MULT) * 1) & 8191;
    unsigned acc = 0;
    long long t0 = clock64();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned addr = (a + 64 * u + it) & 8191;
            if (MODE == 0) acc += s[addr];
            else if (MODE == 1) __hip_atomic_fetch_add(&s[addr], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), acc += 0;
            else if (MODE == 2) acc += __hip_atomic_fetch_add(&s[addr], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else if (MODE == 3) acc += atomicCAS(&s[addr], 0xFFFFFFFFu, 5u);
        }
    }
    long long t1 = clock64();
    out[blockIdx.x * blockDim.x + t] = acc;
    if (t == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE, int MULT> void run(const char *name, int waves)
{
    unsigned *out; long long *cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 2000;
    hipLaunchKernelGGL((k<MODE, MULT>), dim3(256), dim3(64 * waves), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= 256;
    // clock64 ticks at 100 MHz on gfx9?  report raw ticks per wave-instruction per CU
    printf("%-28s mult %2d waves %2d: %8.1f ticks per wave-instr per CU (raw %lld)\n", name, MULT, waves, avg / (double)(iters * 8 * waves), h[0]);
    hipFree(out); hipFree(cyc);
}
int main()
{
    for (int w : {4, 8, 16}) {
        run<0, 1>("ds_read_b32", w);
        run<1, 1>("ds_add (no return)", w);
        run<2, 1>("ds_add_rtn", w);
        run<3, 1>("ds_cmpst_rtn", w);
        run<1, 4>("ds_add (no return)", w);
        run<2, 4>("ds_add_rtn", w);
        run<1, 16>("ds_add (no return)", w);
        run<2, 16>("ds_add_rtn", w);
    }
    return 0;
}
