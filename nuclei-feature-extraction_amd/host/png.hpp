// png.hpp -- minimal PNG decoder (zlib inflate + unfiltering) to the reference's image tensor layout.
//
// The reference loads .png/.jpg with tch::vision::image::load (main.rs:25-27), which yields a u8 tensor
// [3,H,W] (planar RGB; alpha dropped, gray replicated).  Supported here: 8-bit, non-interlaced, colour types
// 0 (gray), 2 (RGB), 3 (palette), 4 (gray+alpha), 6 (RGBA).  JPEG and OpenSlide formats need decoders that
// are not in this image; the CLI reports them as unsupported instead of guessing.
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include <zlib.h>

namespace nfxhost {

struct ImageU8 { // planar [3][H][W]
    int64_t H = 0, W = 0;
    std::vector<uint8_t> data;
};

inline uint32_t be32(const uint8_t *p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }

inline ImageU8 decode_png(const uint8_t *buf, size_t len)
{
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (len < 8 || memcmp(buf, sig, 8) != 0) throw std::runtime_error("png: bad signature");
    size_t pos = 8;
    uint32_t W = 0, H = 0;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte;
    bool have_ihdr = false;
    while (pos + 12 <= len) {
        const uint32_t clen = be32(buf + pos);
        const char *type = reinterpret_cast<const char *>(buf + pos + 4);
        const uint8_t *body = buf + pos + 8;
        if (pos + 12 + clen > len) throw std::runtime_error("png: truncated chunk");
        if (!memcmp(type, "IHDR", 4)) {
            if (clen < 13) throw std::runtime_error("png: bad IHDR");
            W = be32(body); H = be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12];
            have_ihdr = true;
        } else if (!memcmp(type, "PLTE", 4)) {
            plte.assign(body, body + clen);
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + clen);
        } else if (!memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + clen;
    }
    if (!have_ihdr || W == 0 || H == 0) throw std::runtime_error("png: missing IHDR");
    if (depth != 8) throw std::runtime_error("png: only 8-bit samples are supported");
    if (interlace != 0) throw std::runtime_error("png: interlaced images are not supported");
    int ch;
    switch (ctype) {
    case 0: ch = 1; break;
    case 2: ch = 3; break;
    case 3: ch = 1; break;
    case 4: ch = 2; break;
    case 6: ch = 4; break;
    default: throw std::runtime_error("png: unknown colour type");
    }
    const size_t stride = (size_t)W * ch;
    std::vector<uint8_t> raw((stride + 1) * H);
    uLongf rawlen = (uLongf)raw.size();
    if (uncompress(raw.data(), &rawlen, idat.data(), (uLong)idat.size()) != Z_OK || rawlen != raw.size())
        throw std::runtime_error("png: inflate failed");
    // unfilter in place (PNG spec 9.2)
    std::vector<uint8_t> prev(stride, 0);
    ImageU8 img;
    img.H = H; img.W = W;
    img.data.resize((size_t)3 * H * W);
    for (uint32_t y = 0; y < H; ++y) {
        uint8_t *line = raw.data() + (stride + 1) * y;
        const int ft = line[0];
        uint8_t *cur = line + 1;
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= (size_t)ch ? cur[i - ch] : 0, b = prev[i], c = i >= (size_t)ch ? prev[i - ch] : 0;
            int pred = 0;
            switch (ft) {
            case 0: pred = 0; break;
            case 1: pred = a; break;
            case 2: pred = b; break;
            case 3: pred = (a + b) >> 1; break;
            case 4: {
                const int pa = abs(b - c), pb = abs(a - c), pc = abs(a + b - 2 * c);
                pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                break;
            }
            default: throw std::runtime_error("png: bad filter type");
            }
            cur[i] = (uint8_t)(cur[i] + pred);
        }
        memcpy(prev.data(), cur, stride);
        uint8_t *R = img.data.data() + (size_t)y * W, *G = R + (size_t)H * W, *B = G + (size_t)H * W;
        for (uint32_t x = 0; x < W; ++x) {
            const uint8_t *px = cur + (size_t)x * ch;
            switch (ctype) {
            case 0: case 4: R[x] = G[x] = B[x] = px[0]; break;
            case 2: case 6: R[x] = px[0]; G[x] = px[1]; B[x] = px[2]; break;
            case 3: {
                const size_t k = (size_t)px[0] * 3;
                if (k + 2 >= plte.size()) throw std::runtime_error("png: palette index out of range");
                R[x] = plte[k]; G[x] = plte[k + 1]; B[x] = plte[k + 2];
                break;
            }
            }
        }
    }
    return img;
}

}  // namespace nfxhost
