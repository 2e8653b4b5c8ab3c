// Which engine carries a device-to-pinned-host copy, by copy kind (run on the GPU box under the tracer: a copy made by the runtime's copy
// kernel shows up as __amd_rocclr_copyBuffer in the kernel trace, one made by the DMA engines in the memory-copy trace only), and
// what a busy chip does to each: a spinning kernel that fills every CU runs beside the second round of copies; in the third round the
// copy's stream waits for that kernel through an event.
//   hipcc --offload-arch=gfx950 -O2 scripts/d2h_probe.hip -o /tmp/d2h_probe && rocprofv3 --kernel-trace --memory-copy-trace --stats -d out -- /tmp/d2h_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void spin(float *x, long iters) // 256 VGPR-light threads per block; holds its CU slot for ~iters FMAs
{
    float a = x[threadIdx.x], b = 1.0001f;
    for (long i = 0; i < iters; ++i) a = a * b + 0.5f;
    if (a == 12345.f) x[0] = a;
}
__global__ void fill(unsigned *x, size_t n, unsigned seed) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] = (unsigned)i * 2654435761u + seed; }
int main(int argc, char **argv)
{
    const size_t N = (size_t)320 << 20;
    unsigned *d, *h; float *dx;
    // (argument 1: the source lies that many GiB into one large allocation, as a workspace arena's buffers do)
    const size_t big_gib = argc > 1 ? (size_t)atol(argv[1]) : 0;
    if (big_gib) { char *big; CK(hipMalloc(&big, (big_gib << 30) + N)); d = (unsigned *)(big + (big_gib << 30)); printf("source %zu GiB into one allocation\n", big_gib); }
    else CK(hipMalloc(&d, N)); CK(hipMalloc(&dx, 4096)); CK(hipMemset(dx, 0, 4096));
    CK(hipHostMalloc(&h, N, hipHostMallocDefault));
    unsigned *d2, *h2;
    CK(hipMalloc(&d2, N)); CK(hipHostMalloc(&h2, N, hipHostMallocDefault)); memset(h2, 3, N);
    hipStream_t sc, sk;
    CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking));
    struct { const char *name; hipMemcpyKind kind; } kinds[] = {{"hipMemcpyDeviceToHost", hipMemcpyDeviceToHost}, {"hipMemcpyDefault", hipMemcpyDefault},
                                                               {"hipMemcpyDeviceToDeviceNoCU", hipMemcpyDeviceToDeviceNoCU}};
    hipEvent_t ev;
    CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    for (int busy = 0; busy < 4; ++busy) // 3: H2D copies of another stream are in flight (the DMA engines are not idle); 2: the copy's stream WAITS for the kernel's end through an event (as a pipeline stage does)
        for (auto &k : kinds) {
            hipLaunchKernelGGL(fill, dim3(1024), dim3(256), 0, sc, d, N / 4, (unsigned)(busy * 16 + (int)k.kind));
            CK(hipStreamSynchronize(sc));
            memset(h, 0, N);
            double tk0 = now();
            if (busy == 3) for (int r = 0; r < 4; ++r) CK(hipMemcpyAsync(d2, h2, N, hipMemcpyHostToDevice, sk));
            else if (busy) hipLaunchKernelGGL(spin, dim3(256 * 8), dim3(1024), 0, sk, dx, 300000L); // every CU full of waves for a while
            if (busy == 2) { CK(hipEventRecord(ev, sk)); CK(hipStreamWaitEvent(sc, ev, 0)); }
            double t0 = now();
            hipError_t e = hipMemcpyAsync(h, d, N, k.kind, sc);
            if (e != hipSuccess) { printf("%-30s busy %d: refused (%s)\n", k.name, busy, hipGetErrorString(e)); (void)hipGetLastError(); CK(hipDeviceSynchronize()); continue; }
            CK(hipStreamSynchronize(sc));
            double t1 = now();
            CK(hipStreamSynchronize(sk));
            double tk1 = now();
            size_t bad = 0;
            for (size_t i = 0; i < N / 4; i += 4099) bad += h[i] != (unsigned)i * 2654435761u + (unsigned)(busy * 16 + (int)k.kind);
            printf("%-30s busy %d: %6.1f ms = %5.1f GB/s, %zu wrong words; spin kernel %6.1f ms\n", k.name, busy, 1e3 * (t1 - t0), N / (t1 - t0) / 1e9, bad, busy ? 1e3 * (tk1 - tk0) : 0.0);
        }
    return 0;
}
