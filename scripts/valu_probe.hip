// valu_probe.hip -- issue cost of packed / plain FP32 FMAs on gfx950, by waves per SIMD (evidence for DESIGN.md "Gabor").
// Build: hipcc -O3 --offload-arch=gfx950 scripts/valu_probe.hip -o scratch_exp/valu_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void probe(float *out, long long *cyc, int iters)
{
    f2 acc[16];
    f2 w = {1.0f + threadIdx.x * 1e-9f, 0.5f}, x[8];
    const f2 sw0 = {__builtin_amdgcn_readfirstlane(iters) * 1e-9f + 1.0f, 0.5f}, sw1 = {0.25f, __builtin_amdgcn_readfirstlane(iters) * 1e-9f + 0.75f}; // wave-uniform: SGPR pairs
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (f2){(float)i, (float)threadIdx.x};
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = (f2){0.25f * i, 0.125f * threadIdx.x};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(w), "v"(x[(i + rep) & 7]));
                if (MODE == 1) { // two plain FMAs
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].x) : "v"(w.x), "v"(x[(i + rep) & 7].x));
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].y) : "v"(w.y), "v"(x[(i + rep) & 7].y));
                }
                if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[i]) : "v"(w), "v"(x[(i + rep) & 7]));
                if (MODE == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(w), "v"(acc[(i + 5) & 15])); // all operands distinct pairs
                if (MODE == 4) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(acc[i]) : "v"(w), "v"(x[(i + rep) & 7]));
                if (MODE == 5) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(acc[i]) : "v"(w), "v"(x[(i + rep) & 7]));
                // the fold of k_gabor's fused row pass (round 6): (a + b, a - b) from halves of two register pairs
                if (MODE == 6) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(acc[i]) : "v"(w), "v"(x[(i + rep) & 7]));
                if (MODE == 7) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(acc[i]) : "v"(w), "v"(x[(i + rep) & 7]));
                if (MODE == 8) asm volatile("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(acc[i]) : "v"(w), "v"(x[(i + rep) & 7]));
                if (MODE == 9) { // a fold and the two FMAs it feeds (tap pairs in SGPRs), as taps2_fold2 issues them per output and tap
                    f2 sd;
                    asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(sd) : "v"(x[(i + rep) & 7]), "v"(x[(i + rep + 3) & 7]));
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "s"(sw0), "v"(sd));
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[(i + 8) & 15]) : "s"(sw1), "v"(sd));
                }
                if (MODE == 10) { // the same work unfolded: four FMAs with scalar tap pairs (two taps x two groups)
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[i]) : "s"(sw0), "v"(x[(i + rep) & 7]));
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[i]) : "s"(sw1), "v"(x[(i + rep + 3) & 7]));
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[(i + 8) & 15]) : "s"(sw0), "v"(x[(i + rep) & 7]));
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[(i + 8) & 15]) : "s"(sw1), "v"(x[(i + rep + 3) & 7]));
                }
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int MODE>
void run(const char *name, int waves_per_simd)
{
    const int threads = 64 * 4 * waves_per_simd, blocks = 256, iters = 2000;
    float *out; long long *cyc;
    hipMalloc(&out, sizeof(float) * threads * blocks);
    hipMalloc(&cyc, sizeof(long long) * blocks * threads / 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MODE><<<blocks, threads>>>(out, cyc, 10);
    hipEventRecord(e0);
    probe<MODE><<<blocks, threads>>>(out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks * threads / 64);
    hipMemcpy(h.data(), cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += v; mean /= h.size();
    const double n_inst = (double)iters * 64 * (MODE == 1 ? 2 : (MODE == 9 ? 3 : (MODE == 10 ? 4 : 1))); // per wave
    const double flops = (double)iters * 64 * 64 * 4.0 * (threads / 64) * blocks * (MODE == 4 ? 0.5 : 1.0);
    printf("%-34s waves/SIMD %d: %.2f counter-ticks per instruction per wave, %.1f TFLOP/s (%.3f ms)\n", name, waves_per_simd, mean / n_inst, flops / ms * 1e-9, ms);
    hipFree(out); hipFree(cyc);
}

int main()
{
    for (int w = 1; w <= 4; w *= 2) {
        run<0>("v_pk_fma_f32 (w, x[8], acc)", w);
        run<2>("v_pk_fma_f32 x broadcast (op_sel)", w);
        run<3>("v_pk_fma_f32 src1 = other acc", w);
        run<1>("2 x v_fma_f32", w);
        run<4>("v_pk_mul_f32", w);
        run<5>("v_pk_add_f32", w);
        run<6>("v_pk_add_f32 op_sel [0,1] neg_hi", w);
        run<7>("v_pk_add_f32 op_sel [1,0] neg_hi", w);
        run<8>("v_pk_add_f32 neg_hi only", w);
        run<9>("fold + 2 pk_fma (3 instr)", w);
        run<10>("4 pk_fma, scalar taps (4 instr)", w);
    }
    return 0;
}
