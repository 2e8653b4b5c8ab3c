// geojson.hpp -- streaming reader for the reference's GeoJSON wire format into packed polygon buffers.
//
// Mirrors the serde structs of the reference (geojson.rs:8-24):
//   FeatureCollection { features: [ Feature { bbox: Vec<f32>  (REQUIRED, no serde default),
//                                             geometry: { type: String, coordinates: Vec<Vec<Vec<f32>>> } } ] }
// Unknown keys are ignored at every level (serde's default), numbers are parsed as f64 and narrowed to f32,
// only coordinates[0] -- the outer ring -- is used (utils.rs:55).  One pass over the bytes, no DOM: 10 M nuclei
// are a multi-gigabyte file.
//
// Large collections are parsed by SEVERAL THREADS (the reference's serde reader is one thread; a 10 M-nucleus file is 12 GB and a
// minute of parsing at 0.8 GB/s): the `features` array is cut into byte chunks; a first pass counts every chunk's string quotes
// (their parity is the string state at the next chunk's start), a second one, skipping strings quote to quote, how the bracket
// depth moves; a third pass with the true states collects the offsets of the array's elements; a fourth one
// parses contiguous runs of features with the very same recursive-descent code into per-thread buffers, which are then joined in
// order.  The result is the serial parser's, byte for byte.  Anything unexpected on the way (an array that never closes, a run that
// does not end where the next one starts, any parse error) abandons the attempt and the document is parsed again serially, so
// every error message, and the byte offset in it, is the serial reader's.
#pragma once
#include <algorithm>
#include <charconv>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <string_view>
#include <thread>
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
    // threads: 0 = one thread per hardware thread (at most 32) for documents of 8 MB and more, 1 = the serial reader
    explicit GeoJsonReader(std::string_view text, bool keep_all = false, int threads = 0)
        : p_(text.data()), end_(text.data() + text.size()), begin_(text.data()), keep_all_(keep_all), threads_(threads) {}
    std::vector<FullFeature> full; // keep_all only
    size_t parallel_min_bytes = (size_t)8 << 20; // (tests lower it to take the threaded path on small documents)
    int threads_used = 1;                        // how the last parse() ran

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
                    if (!features_parallel(out)) features(out);
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
    int threads_ = 0;

    // ---- the threaded path (header comment) ------------------------------------------------------------------------------------
    struct ScanState { bool in_string; long long depth; };
    // quotes that open or close a string: those not preceded by an odd run of backslashes (a backslash only occurs inside strings in
    // a valid document; an invalid one fails in the parsing pass and is parsed again serially).  [b, e) never starts behind a backslash.
    static bool is_string_quote(const char *b, const char *r)
    {
        const char *s = r;
        while (s > b && s[-1] == '\\') --s;
        return ((r - s) & 1) == 0;
    }
    static size_t count_string_quotes(const char *b, const char *e)
    {
        size_t n = 0;
        for (const char *q = b; q < e;) {
            const char *r = static_cast<const char *>(memchr(q, '"', (size_t)(e - q)));
            if (!r) break;
            n += is_string_quote(b, r) ? 1 : 0;
            q = r + 1;
        }
        return n;
    }
    // one chunk from a known starting state: strings are skipped quote to quote, brackets count outside strings (a 256-entry class
    // table skips the digits, commas and dots that make up nearly all of a nucleus file).  With `starts`, the offsets of '{' met at
    // depth 1 (the elements of the features array) are collected and the scan stops where the depth reaches 0 (the array's closing
    // bracket: `*array_end`).
    static ScanState scan_chunk(const char *b, const char *e, ScanState st, std::vector<const char *> *starts, const char **array_end)
    {
        static const struct Cls { unsigned char t[256]; Cls() { for (int i = 0; i < 256; ++i) t[i] = 0; t[(unsigned char)'"'] = 1; t[(unsigned char)'{'] = 2; t[(unsigned char)'['] = 2; t[(unsigned char)'}'] = 3; t[(unsigned char)']'] = 3; } } cls;
        for (const char *q = b; q < e;) {
            if (st.in_string) {
                const char *r = static_cast<const char *>(memchr(q, '"', (size_t)(e - q)));
                if (!r) break;
                if (is_string_quote(b, r)) st.in_string = false;
                q = r + 1;
                continue;
            }
            while (q < e && !cls.t[(unsigned char)*q]) ++q;
            if (q >= e) break;
            const unsigned char k = cls.t[(unsigned char)*q];
            if (k == 1) {
                st.in_string = true;
            } else if (k == 2) {
                if (starts && *q == '{' && st.depth == 1) starts->push_back(q);
                ++st.depth;
            } else {
                --st.depth;
                if (starts && st.depth == 0) { *array_end = q; return st; }
            }
            ++q;
        }
        return st;
    }
    // p_ stands before the '[' of the features array.  true: `out` (and `full`) hold every feature and p_ stands behind the ']';
    // false: nothing was touched -- the caller parses serially.  ONE set of threads runs all four passes, a barrier between them
    // (thread 0 does the few serial steps in between): threads are started once, not once per pass.
    struct Barrier {
        std::mutex m; std::condition_variable cv; int n, count = 0, gen = 0;
        explicit Barrier(int n_) : n(n_) {}
        void wait()
        {
            std::unique_lock<std::mutex> l(m);
            const int g = gen;
            if (++count == n) { count = 0; ++gen; cv.notify_all(); }
            else cv.wait(l, [&] { return g != gen; });
        }
    };
    bool features_parallel(Polygons &out)
    {
        ws();
        int T = threads_ > 0 ? threads_ : (int)std::min<unsigned>(32u, std::max(1u, std::thread::hardware_concurrency()));
        const size_t bytes = (size_t)(end_ - p_);
        if (T < 2 || bytes < parallel_min_bytes || p_ >= end_ || *p_ != '[') return false;
        T = (int)std::min<size_t>((size_t)T, std::max<size_t>(1, bytes / 4096));
        if (T < 2) return false;
        // chunk boundaries, none of them right behind a backslash (an escape never straddles a boundary)
        std::vector<const char *> cut((size_t)T + 1);
        cut[0] = p_; cut[(size_t)T] = end_;
        for (int k = 1; k < T; ++k) {
            const char *q = p_ + bytes / (size_t)T * (size_t)k;
            while (q < end_ && q > p_ && q[-1] == '\\') ++q;
            cut[(size_t)k] = std::max(q, cut[(size_t)k - 1]);
        }
        std::vector<size_t> quotes((size_t)T);
        std::vector<ScanState> begin_state((size_t)T), end_state((size_t)T);
        std::vector<std::vector<const char *>> starts((size_t)T);
        std::vector<const char *> closes((size_t)T, nullptr), elem;
        const char *array_end = nullptr;
        std::vector<Polygons> part((size_t)T);
        std::vector<std::vector<FullFeature>> part_full((size_t)T);
        std::vector<char> ok((size_t)T, 0);
        bool give_up = false; // set by thread 0 between two barriers, read by all behind the second one
        // a pass that throws in one thread (bad_alloc while collecting element starts) must not leave the others waiting at a barrier, nor
        // end the process in std::terminate: the thread notes its failure and still ARRIVES at every barrier; thread 0 turns any failure
        // into give_up at the next serial step, and the document is parsed again by the serial reader
        std::vector<char> failed((size_t)T, 0);
        auto any_failed = [&] { for (char f : failed) if (f) return true; return false; };
        Barrier bar(T);
        // the workers start behind a gate: if a thread cannot be created (a pid / memory limit of the cgroup) the ones that exist are
        // told to leave before any of them waits at a barrier sized for T
        std::mutex gate_m; std::condition_variable gate_cv; int gate = 0; // 0 wait, 1 go, -1 leave
        auto worker = [&](int k) {
            if (k != 0) {
                std::unique_lock<std::mutex> l(gate_m);
                gate_cv.wait(l, [&] { return gate != 0; });
                if (gate < 0) return;
            }
            // pass 1: string quotes per chunk -> string state at every chunk start
            try { quotes[(size_t)k] = count_string_quotes(cut[(size_t)k], cut[(size_t)k + 1]); } catch (...) { failed[(size_t)k] = 1; }
            bar.wait();
            if (k == 0) {
                size_t q = 0;
                for (int j = 0; j < T; ++j) { begin_state[(size_t)j] = ScanState{(q & 1) != 0, 0}; q += quotes[(size_t)j]; }
            }
            bar.wait();
            // pass 2: bracket depth per chunk -> depth at every chunk start
            try { end_state[(size_t)k] = scan_chunk(cut[(size_t)k], cut[(size_t)k + 1], begin_state[(size_t)k], nullptr, nullptr); } catch (...) { failed[(size_t)k] = 1; }
            bar.wait();
            if (k == 0) {
                if (any_failed()) give_up = true;
                long long d = 0;
                for (int j = 0; j < T; ++j) {
                    if (j > 0 && end_state[(size_t)j - 1].in_string != begin_state[(size_t)j].in_string) give_up = true; // (the two passes disagree: not a valid document)
                    begin_state[(size_t)j].depth = d;
                    d += end_state[(size_t)j].depth;
                }
            }
            bar.wait();
            if (give_up) return;
            // pass 3: the elements of the array, chunk by chunk, up to its closing bracket
            try { (void)scan_chunk(cut[(size_t)k], cut[(size_t)k + 1], begin_state[(size_t)k], &starts[(size_t)k], &closes[(size_t)k]); } catch (...) { failed[(size_t)k] = 1; }
            bar.wait();
            if (k == 0) {
                try {
                    for (int j = 0; j < T && !array_end && !any_failed(); ++j) {
                        elem.insert(elem.end(), starts[(size_t)j].begin(), starts[(size_t)j].end());
                        array_end = closes[(size_t)j]; // (a chunk behind the array's end is never consulted)
                    }
                } catch (...) { failed[0] = 1; }
                if (any_failed() || !array_end || elem.empty()) give_up = true; // unbalanced, empty or out of memory: the serial reader says what is wrong (or returns nothing)
            }
            bar.wait();
            if (give_up) return;
            // pass 4: contiguous runs of features, the same parser, private buffers
            const size_t a = elem.size() * (size_t)k / (size_t)T, b = elem.size() * ((size_t)k + 1) / (size_t)T;
            const char *stop = b < elem.size() ? elem[b] : array_end;
            try {
                GeoJsonReader sub(std::string_view(begin_, (size_t)(end_ - begin_)), keep_all_, 1);
                sub.p_ = a < elem.size() ? elem[a] : stop;
                for (size_t i = a; i < b; ++i) {
                    if (sub.p_ != elem[i]) return;      // the run does not stand where the scan saw the element: give up
                    sub.feature(part[(size_t)k]);
                    const bool more = sub.comma();
                    sub.ws();
                    if (more != (i + 1 < elem.size())) return; // a comma where the array ends, or none between two elements
                }
                if (b > a && sub.p_ != stop) return;
                part_full[(size_t)k] = std::move(sub.full);
                ok[(size_t)k] = 1;
            } catch (...) { // (a parse error or bad_alloc: ok[k] stays 0 and the document goes to the serial reader)
            }
        };
        {
            std::vector<std::thread> th;
            bool started = true;
            try {
                th.reserve((size_t)T);
                for (int k = 1; k < T; ++k) th.emplace_back(worker, k);
            } catch (...) { started = false; } // std::system_error: no more threads to be had
            { std::lock_guard<std::mutex> l(gate_m); gate = started ? 1 : -1; }
            gate_cv.notify_all();
            if (started) worker(0);
            for (std::thread &t : th) t.join();
            if (!started) return false;
        }
        if (give_up) return false;
        for (int k = 0; k < T; ++k)
            if (!ok[(size_t)k]) return false;
        // whatever stands between '[' and the first element has to be white space (the serial reader would have said otherwise)
        for (const char *q = p_ + 1; q < elem[0]; ++q)
            if (!(*q == ' ' || *q == '\n' || *q == '\t' || *q == '\r')) return false;
        size_t nv = out.ring_xy.size() / 2, nf = 0;
        for (const Polygons &pp : part) { nv += pp.ring_xy.size() / 2; nf += pp.size(); }
        if (nv > 0xFFFFFFFFull) return false; // (the serial reader reports it)
        out.ring_xy.reserve(2 * nv);
        out.ring_offsets.reserve(out.ring_offsets.size() + nf);
        for (int k = 0; k < T; ++k) {
            const uint32_t base = (uint32_t)(out.ring_xy.size() / 2);
            const Polygons &pp = part[(size_t)k];
            out.ring_xy.insert(out.ring_xy.end(), pp.ring_xy.begin(), pp.ring_xy.end());
            for (size_t i = 1; i < pp.ring_offsets.size(); ++i) out.ring_offsets.push_back(base + pp.ring_offsets[i]);
            if (keep_all_) for (FullFeature &f : part_full[(size_t)k]) full.push_back(std::move(f));
            part[(size_t)k] = Polygons(); // appended: give the run's buffers back now (the peak stays near one copy of the rings, not two)
            part_full[(size_t)k] = std::vector<FullFeature>();
        }
        p_ = array_end + 1;
        threads_used = T;
        return true;
    }

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
