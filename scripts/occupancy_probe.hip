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
    probe("k_glcm<u8, 512,3> (ps 32)", k_glcm<uint8_t, 512, 3>);
    probe("k_glcm<u8, 768,3> (ps 32)", k_glcm<uint8_t, 768, 3>);
    probe("k_glcm<u8, 512,7> (ps 48)", k_glcm<uint8_t, 512, 7>);
    probe("k_glcm<u8, 768,7> (ps 48)", k_glcm<uint8_t, 768, 7>);
    probe("k_glcm<u8,1024,7> (ps 48)", k_glcm<uint8_t, 1024, 7>);
    probe("k_glcm<u8,1536,7> (ps 48)", k_glcm<uint8_t, 1536, 7>);
    probe("k_glcm<u8,2048,7> (ps 48)", k_glcm<uint8_t, 2048, 7>);
    probe("k_glcm<u8, 512,12> (ps 64)", k_glcm<uint8_t, 512, 12>);
    probe("k_glcm<u8, 768,12> (ps 64)", k_glcm<uint8_t, 768, 12>);
    probe("k_glcm<u8,1024,12> (ps 64)", k_glcm<uint8_t, 1024, 12>);
    probe("k_glcm<u8,1536,12> (ps 64)", k_glcm<uint8_t, 1536, 12>);
    probe("k_glcm<u8,2048,12> (ps 64)", k_glcm<uint8_t, 2048, 12>);
    probe("k_glcm<u8,4096,12> (ps 64)", k_glcm<uint8_t, 4096, 12>);
    return 0;
}
