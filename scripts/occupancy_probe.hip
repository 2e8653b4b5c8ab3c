// occupancy_probe.hip -- how many single-wave k_glcm workgroups a CU really holds per size class (LDS granularity included),
// asked of the runtime.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I nuclei-feature-extraction_amd/csrc
//        scripts/occupancy_probe.hip -o gpurun_out/occupancy_probe ; run on the GPU box.
#include <cstdio>
#include "../nuclei-feature-extraction_amd/csrc/k_glcm.hip"
namespace nfx {
bool stream_capturing(nfx_ctx *) { return false; }
void timing_begin(nfx_ctx *, int) {}
void timing_end(nfx_ctx *) {}
}
template <typename K> static void probe(const char *name, K kernel)
{
    int blocks = 0;
    hipFuncAttributes a{};
    (void)hipFuncGetAttributes(&a, reinterpret_cast<const void *>(kernel));
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, kernel, 64, 0);
    printf("%-28s static LDS %6zu B, %3d VGPRs -> %2d workgroups (waves) per CU\n", name, a.sharedSizeBytes, a.numRegs, blocks);
}
int main()
{
    using namespace nfx;
    probe("k_glcm<u8,10,3> (ps 32)", k_glcm<uint8_t, 10, 3>);
    probe("k_glcm<u8,10,7> (ps 48)", k_glcm<uint8_t, 10, 7>);
    probe("k_glcm<u8,11,7> (ps 48)", k_glcm<uint8_t, 11, 7>);
    probe("k_glcm<u8,12,7> (ps 48)", k_glcm<uint8_t, 12, 7>);
    probe("k_glcm<u8,10,12> (ps 64)", k_glcm<uint8_t, 10, 12>);
    probe("k_glcm<u8,11,12> (ps 64)", k_glcm<uint8_t, 11, 12>);
    probe("k_glcm<u8,12,12> (ps 64)", k_glcm<uint8_t, 12, 12>);
    probe("k_glcm<u8,13,12> (ps 64)", k_glcm<uint8_t, 13, 12>);
    return 0;
}
