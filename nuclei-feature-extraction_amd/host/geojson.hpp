// geojson.hpp -- streaming reader for the reference's GeoJSON wire format into packed polygon buffers.
//
// Mirrors the serde structs of the reference (geojson.rs:8-24):
//   FeatureCollection { features: [ Feature { bbox: Vec<f32>  (REQUIRED, no serde default),
//                                             geometry: { type: String, coordinates: Vec<Vec<Vec<f32>>> } } ] }
// Unknown keys are ignored at every level (serde's default), numbers are parsed as f64 and narrowed to f32,
// only coordinates[0] -- the outer ring -- is used (utils.rs:55).  One pass over the bytes, no DOM: 10 M nuclei
// are a multi-gigabyte file.
#pragma once
#include <charconv>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <string_view>
#include <vector>

namespace nfxhost {

struct Polygons {
    std::vector<uint32_t> ring_offsets{0}; // [n+1]
    std::vector<float> ring_xy;            // [sum V][2]
    size_t size() const { return ring_offsets.size() - 1; }
};

// Everything the reference's serde structs keep of a feature (geojson.rs:8-24); filled only when the reader is
// constructed with keep_all = true (filter-geojson needs bbox, type and every ring with every coordinate).
struct FullFeature {
    std::vector<float> bbox;
    std::string type;
    std::vector<std::vector<std::vector<float>>> coordinates;
};

class GeoJsonReader {
  public:
    explicit GeoJsonReader(std::string_view text, bool keep_all = false)
        : p_(text.data()), end_(text.data() + text.size()), begin_(text.data()), keep_all_(keep_all) {}
    std::vector<FullFeature> full; // keep_all only

    Polygons parse()
    {
        Polygons out;
        bool have_features = false;
        ws();
        expect('{');
        if (!peek_is('}')) {
            do {
                std::string key = string();
                expect(':');
                if (key == "features") {
                    have_features = true;
                    features(out);
                } else {
                    skip_value();
                }
            } while (comma());
        }
        expect('}');
        if (!have_features) fail("missing field `features`");
        return out;
    }

  private:
    const char *p_, *end_, *begin_;
    bool keep_all_ = false;

    [[noreturn]] void fail(const std::string &msg) const
    {
        throw std::runtime_error("geojson: " + msg + " at byte " + std::to_string(p_ - begin_));
    }
    void ws()
    {
        while (p_ < end_ && (*p_ == ' ' || *p_ == '\n' || *p_ == '\t' || *p_ == '\r')) ++p_;
    }
    bool peek_is(char c)
    {
        ws();
        return p_ < end_ && *p_ == c;
    }
    void expect(char c)
    {
        ws();
        if (p_ >= end_ || *p_ != c) fail(std::string("expected `") + c + "`");
        ++p_;
    }
    bool comma()
    {
        ws();
        if (p_ < end_ && *p_ == ',') { ++p_; return true; }
        return false;
    }
    std::string string()
    {
        ws();
        if (p_ >= end_ || *p_ != '"') fail("expected a string");
        ++p_;
        std::string s;
        while (p_ < end_ && *p_ != '"') {
            if (*p_ == '\\') {
                ++p_;
                if (p_ >= end_) fail("bad escape");
                switch (*p_) {
                case 'n': s.push_back('\n'); break;
                case 't': s.push_back('\t'); break;
                case 'r': s.push_back('\r'); break;
                case 'b': s.push_back('\b'); break;
                case 'f': s.push_back('\f'); break;
                case 'u': // keys of this format are ASCII; keep the escape verbatim
                    s.append("\\u");
                    break;
                default: s.push_back(*p_);
                }
                ++p_;
            } else {
                s.push_back(*p_++);
            }
        }
        if (p_ >= end_) fail("unterminated string");
        ++p_;
        return s;
    }
    void skip_string()
    {
        ws();
        if (p_ >= end_ || *p_ != '"') fail("expected a string");
        ++p_;
        while (p_ < end_ && *p_ != '"') p_ += (*p_ == '\\') ? 2 : 1;
        if (p_ >= end_) fail("unterminated string");
        ++p_;
    }
    double number()
    {
        ws();
        double v = 0.0;
        auto r = std::from_chars(p_, end_, v);
        if (r.ec != std::errc()) fail("expected a number");
        p_ = r.ptr;
        return v;
    }
    void skip_value()
    {
        ws();
        if (p_ >= end_) fail("unexpected end of input");
        switch (*p_) {
        case '"': skip_string(); return;
        case '{':
            ++p_;
            if (!peek_is('}')) {
                do { skip_string(); expect(':'); skip_value(); } while (comma());
            }
            expect('}');
            return;
        case '[':
            ++p_;
            if (!peek_is(']')) {
                do { skip_value(); } while (comma());
            }
            expect(']');
            return;
        case 't': literal("true"); return;
        case 'f': literal("false"); return;
        case 'n': literal("null"); return;
        default: (void)number(); return;
        }
    }
    void literal(const char *lit)
    {
        for (const char *q = lit; *q; ++q, ++p_)
            if (p_ >= end_ || *p_ != *q) fail(std::string("expected `") + lit + "`");
    }

    void features(Polygons &out)
    {
        expect('[');
        if (!peek_is(']')) {
            do { feature(out); } while (comma());
        }
        expect(']');
    }
    void feature(Polygons &out)
    {
        bool have_bbox = false, have_geometry = false;
        if (keep_all_) full.emplace_back();
        expect('{');
        if (!peek_is('}')) {
            do {
                std::string key = string();
                expect(':');
                if (key == "bbox") { // Vec<f32>, required (geojson.rs:17)
                    have_bbox = true;
                    expect('[');
                    if (!peek_is(']')) {
                        do {
                            const double v = number();
                            if (keep_all_) full.back().bbox.push_back((float)v);
                        } while (comma());
                    }
                    expect(']');
                } else if (key == "geometry") {
                    have_geometry = true;
                    geometry(out);
                } else {
                    skip_value();
                }
            } while (comma());
        }
        expect('}');
        if (!have_bbox) fail("missing field `bbox`");
        if (!have_geometry) fail("missing field `geometry`");
    }
    void geometry(Polygons &out)
    {
        bool have_type = false, have_coords = false;
        expect('{');
        if (!peek_is('}')) {
            do {
                std::string key = string();
                expect(':');
                if (key == "type") {
                    have_type = true;
                    if (keep_all_) full.back().type = string(); else skip_string();
                }
                else if (key == "coordinates") { have_coords = true; coordinates(out); }
                else skip_value();
            } while (comma());
        }
        expect('}');
        if (!have_type) fail("missing field `type`");
        if (!have_coords) fail("missing field `coordinates`");
    }
    void coordinates(Polygons &out) // Vec<Vec<Vec<f32>>>: rings -> points -> numbers
    {
        expect('[');
        size_t ring = 0;
        if (!peek_is(']')) {
            do {
                expect('[');
                if (keep_all_) full.back().coordinates.emplace_back();
                if (!peek_is(']')) {
                    do {
                        expect('[');
                        int k = 0;
                        float xy[2] = {0.f, 0.f};
                        if (keep_all_) full.back().coordinates.back().emplace_back();
                        if (!peek_is(']')) {
                            do {
                                const double v = number();
                                if (k < 2) xy[k] = (float)v; // f64 -> f32, as serde does for f32 fields
                                if (keep_all_) full.back().coordinates.back().back().push_back((float)v);
                                ++k;
                            } while (comma());
                        }
                        expect(']');
                        if (ring == 0) {
                            if (k < 2) fail("a ring vertex needs two coordinates (utils.rs:57-58 indexes [0] and [1])");
                            out.ring_xy.push_back(xy[0]);
                            out.ring_xy.push_back(xy[1]);
                        }
                    } while (comma());
                }
                expect(']');
                ++ring;
            } while (comma());
        }
        expect(']');
        if (ring == 0) fail("geometry has no ring (utils.rs:55 takes coordinates[0])");
        if (out.ring_xy.size() / 2 > 0xFFFFFFFFull) fail("more than 2^32-1 vertices");
        out.ring_offsets.push_back((uint32_t)(out.ring_xy.size() / 2));
    }
};

}  // namespace nfxhost
