// filter_geojson.cpp -- `filter-geojson <input> <output> <offset_x> <offset_y> <size_x> <size_y>`
//
// The reference's second binary (src/filter_geojson.rs:16-67): keeps the features whose bbox lies inside the
// window [offset, offset+size) and re-origins their outer ring to the window.  Reproduced as written, including
// its quirk: the bbox is READ as [left, top, right, down] (filter_geojson.rs:33) but WRITTEN back as
// [top-oy, left-ox, down-oy, right-ox] (filter_geojson.rs:47-52); only coordinates[0] is shifted (54-59).
// Features whose bbox does not have exactly four numbers are dropped with a message (33-36).  The output holds
// exactly the fields the serde structs model (geojson.rs:8-24): features[].{bbox, geometry.{type, coordinates}}.
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>

#include "geojson.hpp"

using namespace nfxhost;

static void put_f32(std::string &s, float v)
{
    if (!std::isfinite(v)) { s += "null"; return; } // serde_json writes non-finite floats as null
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof buf, v);
    std::string t(buf, r.ptr);
    if (t.find_first_of(".en") == std::string::npos) t += ".0"; // serde_json/ryu keeps floats recognisable
    s += t;
}

int main(int argc, char **argv)
{
    if (argc != 7) {
        fprintf(stderr, "Usage: filter-geojson <INPUT> <OUTPUT> <OFFSET_X> <OFFSET_Y> <SIZE_X> <SIZE_Y>\n");
        return 2;
    }
    char *e = nullptr;
    unsigned long long v[4];
    for (int k = 0; k < 4; ++k) {
        v[k] = strtoull(argv[3 + k], &e, 10);
        if (*e || argv[3 + k][0] == '-' || !argv[3 + k][0]) { fprintf(stderr, "error: invalid value '%s'\n", argv[3 + k]); return 2; }
    }
    const float ox = (float)v[0], oy = (float)v[1], ex = (float)(v[0] + v[2]), ey = (float)(v[1] + v[3]);
    std::ifstream f(argv[1], std::ios::binary);
    if (!f) { fprintf(stderr, "cannot open %s\n", argv[1]); return 1; }
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string text = ss.str();
    try {
        GeoJsonReader rd(text, true);
        rd.parse();
        std::string out = "{\"features\":[";
        bool first = true;
        for (auto &ft : rd.full) {
            if (ft.bbox.size() != 4) { printf("No bbox for feature\n"); continue; }
            const float left = ft.bbox[0], top = ft.bbox[1], right = ft.bbox[2], down = ft.bbox[3];
            if (top < oy || left < ox || down > ey || right > ex) continue;
            if (!first) out += ',';
            first = false;
            out += "{\"bbox\":[";
            const float nb[4] = {top - oy, left - ox, down - oy, right - ox};
            for (int k = 0; k < 4; ++k) { if (k) out += ','; put_f32(out, nb[k]); }
            out += "],\"geometry\":{\"type\":\"" + ft.type + "\",\"coordinates\":[";
            for (size_t r = 0; r < ft.coordinates.size(); ++r) {
                if (r) out += ',';
                out += '[';
                for (size_t p = 0; p < ft.coordinates[r].size(); ++p) {
                    if (p) out += ',';
                    out += '[';
                    auto pt = ft.coordinates[r][p];
                    if (r == 0) { pt[0] -= ox; pt[1] -= oy; } // the reader guarantees two coordinates on ring 0
                    for (size_t k = 0; k < pt.size(); ++k) { if (k) out += ','; put_f32(out, pt[k]); }
                    out += ']';
                }
                out += ']';
            }
            out += "]}}";
        }
        out += "]}";
        std::ofstream o(argv[2], std::ios::binary);
        if (!o) { fprintf(stderr, "cannot create %s\n", argv[2]); return 1; }
        o << out;
    } catch (const std::exception &ex2) {
        fprintf(stderr, "%s\n", ex2.what());
        return 1;
    }
    return 0;
}
