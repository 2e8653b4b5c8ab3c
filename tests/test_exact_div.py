"""exact_div.h: the 3-instruction constant divisions equal IEEE division on every input the u8 pixel path produces."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#include <cstdio>
#include "exact_div.h"
int main() {
    long bad = 0;
    float t[256];
    for (int i = 0; i < 256; ++i) { t[i] = (float)i / 255.0f; bad += nfx::u8_unit((unsigned)i) != t[i]; }
    for (int a = 0; a < 256; ++a)
        for (int b = 0; b < 256; ++b) {
            const float ab = t[a] + t[b];
            for (int c = 0; c < 256; ++c) { const float s = ab + t[c]; bad += nfx::third_of_u8_sum(s) != s / 3.0f; }
        }
    printf("%ld\n", bad);
    return bad != 0;
}
"""


def test_constant_divisions_are_exact_on_the_whole_u8_domain(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text(SRC)
    exe = tmp_path / "t"
    inc = os.path.join(ROOT, "nuclei-feature-extraction_amd", "csrc")
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-I", inc, "-o", str(exe), str(src)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.strip() == "0", r.stdout


def test_window_addressing_reciprocals_are_exact():
    """The kernels that read a nucleus's window in place (PatchRef, round 5) turn a pixel or piece index into (row, column) with one
    multiplication by a rounded-up reciprocal instead of a division; every index the kernels can form must come out exactly.
      k_color:            r = (p * ceil(2^20 / ps)) >> 20          for p < ps^2          (16 bits are wrong at ps 48, p 2303 -- a GPU fault in round 5)
      k_glrlm / k_gabor:  r = (q * ceil(2^16 / (ps/16))) >> 16     for q < ps^2 / 16      (pieces of 16 pixels of one channel)
      k_glcm:             ch = (x * ceil(2^20 / (ps^2/16))) >> 20, r = (xx * ceil(2^20 / (ps/16))) >> 20   for x < 3 ps^2 / 16"""
    for ps in (16, 32, 48, 64):
        inv = ((1 << 20) + ps - 1) // ps
        assert all(((p * inv) >> 20) == p // ps for p in range(ps * ps)), ps
        assert any(((p * ((65536 + ps - 1) // ps)) >> 16) != p // ps for p in range(ps * ps)) == (ps == 48)   # why 16 bits were not enough
        ppr, pc = ps // 16, ps * ps // 16
        inv16 = (65536 + ppr - 1) // ppr
        assert all(((q * inv16) >> 16) == q // ppr for q in range(pc)), ps
        inv_pc, inv_ppr = ((1 << 20) + pc - 1) // pc, ((1 << 20) + ppr - 1) // ppr
        for x in range(3 * pc):
            ch = (x * inv_pc) >> 20
            xx = x - ch * pc
            assert ch == x // pc and ((xx * inv_ppr) >> 20) == xx // ppr, (ps, x)
