// Host link rates as libnfx's copies see them (run on the GPU box): H2D, D2H and both at once, out of host memory pinned three ways.
//   hipcc --offload-arch=gfx950 -O2 scripts/h2d_probe.hip -o /tmp/h2d_probe && /tmp/h2d_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    const size_t N = (size_t)2 << 30, PIECE = (size_t)100 << 20;
    void *d_in, *d_out;
    CK(hipMalloc(&d_in, N)); CK(hipMalloc(&d_out, N));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    struct { const char *name; unsigned flags; int mode; } kinds[] = {
        {"hipHostMalloc default", hipHostMallocDefault, 0}, {"hipHostMalloc non-coherent", hipHostMallocNonCoherent, 0},
        {"hipHostMalloc coherent", hipHostMallocCoherent, 0}, {"malloc + hipHostRegister", 0, 1}, {"pageable malloc", 0, 2}};
    for (auto &k : kinds) {
        void *h_in = nullptr, *h_out = nullptr;
        if (k.mode == 0) { CK(hipHostMalloc(&h_in, N, k.flags)); CK(hipHostMalloc(&h_out, N, k.flags)); }
        else {
            h_in = aligned_alloc(4096, N); h_out = aligned_alloc(4096, N);
            if (k.mode == 1) { CK(hipHostRegister(h_in, N, hipHostRegisterDefault)); CK(hipHostRegister(h_out, N, hipHostRegisterDefault)); }
        }
        memset(h_in, 1, N); memset(h_out, 2, N);
        for (int rep = 0; rep < 2; ++rep) {
            double t0 = now();
            for (size_t o = 0; o < N; o += PIECE) CK(hipMemcpyAsync((char *)d_in + o, (char *)h_in + o, PIECE < N - o ? PIECE : N - o, hipMemcpyHostToDevice, s1));
            CK(hipStreamSynchronize(s1));
            double h2d = now() - t0;
            t0 = now();
            for (size_t o = 0; o < N; o += PIECE) CK(hipMemcpyAsync((char *)h_out + o, (char *)d_out + o, PIECE < N - o ? PIECE : N - o, hipMemcpyDeviceToHost, s2));
            CK(hipStreamSynchronize(s2));
            double d2h = now() - t0;
            t0 = now();
            for (size_t o = 0; o < N; o += PIECE) {
                CK(hipMemcpyAsync((char *)d_in + o, (char *)h_in + o, PIECE < N - o ? PIECE : N - o, hipMemcpyHostToDevice, s1));
                CK(hipMemcpyAsync((char *)h_out + o, (char *)d_out + o, PIECE < N - o ? PIECE : N - o, hipMemcpyDeviceToHost, s2));
            }
            CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
            double both = now() - t0;
            if (rep) printf("%-28s H2D %5.1f GB/s   D2H %5.1f GB/s   both at once %5.1f + %5.1f = %5.1f GB/s\n", k.name, N / h2d / 1e9, N / d2h / 1e9, N / both / 1e9, N / both / 1e9, 2 * N / both / 1e9);
        }
        if (k.mode == 0) { hipHostFree(h_in); hipHostFree(h_out); }
        else { if (k.mode == 1) { hipHostUnregister(h_in); hipHostUnregister(h_out); } free(h_in); free(h_out); }
    }
    return 0;
}
