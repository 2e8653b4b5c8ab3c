// cli.cpp -- `nuclei-feature-extraction <geometry> <slide> <output> [feature_sets]...`
//
// The C++ host of the MI355X engine behind the reference's command line (args.rs:76-113, readme.md:10):
//   positionals  geometry (.geojson)  slide  output  feature_sets...   (geometry color glcm glrlm gabor texture all)
//   flags        -o/--overwrite  -p/--patch-size <64>  -t/--thread-count <n>  -g/--gpus <i>...
//                -b/--batch-size <100>  -v/--verbose
// and its driver (main.rs:110-190): validate -> load geometry -> load image -> extract -> write by extension.
// Differences, all stated at run time: there is no CPU compute path (without -g the engine uses GPU 0);
// rows come out in input order (the reference's order is batch-completion order); csv, json, parquet/pqt and
// ipc/feather writers (columnar.hpp); only .png slides decode here (no JPEG / OpenSlide in this image).
// Every GPU extracts its index shard through nfx_extract_tile_streamed_v2: the slide travels block by block (only the blocks the
// shard's nuclei touch), chunks of whole reference batches are extracted while the next chunk's blocks are on their way --
// no `n x 3 ps^2` patch workspace and no per-GPU copy of the whole slide, whatever the number of nuclei.
#include <algorithm>
#include <charconv>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>
#include <sys/stat.h>
#include <thread>
#include <vector>

#include "../../include/nfx.h"
#include "columnar.hpp"
#include "geojson.hpp"
#include "png.hpp"

using namespace nfxhost;

namespace {

struct Args {
    std::string geometry, slide, output;
    std::vector<std::string> feature_sets;
    bool overwrite = false, verbose = false, have_gpus = false;
    int patch_size = 64, batch_size = 100;
    long thread_count = -1;
    std::vector<int> gpus;
};

bool g_debug = false;
void log_error(const std::string &m) { fprintf(stderr, " ERROR nuclei_feature_extraction > %s\n", m.c_str()); }
void log_info(const std::string &m) { fprintf(stderr, " INFO  nuclei_feature_extraction > %s\n", m.c_str()); }
void log_debug(const std::string &m) { if (g_debug) fprintf(stderr, " DEBUG nuclei_feature_extraction > %s\n", m.c_str()); }

[[noreturn]] void usage_error(const std::string &m)
{
    fprintf(stderr, "error: %s\n\nUsage: nuclei-feature-extraction [OPTIONS] <GEOMETRY> <SLIDE> <OUTPUT> [FEATURE_SETS]...\n", m.c_str());
    exit(2); // clap's exit code for usage errors
}

bool exists(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0; }
std::string extension(const std::string &p)
{
    const size_t slash = p.find_last_of('/');
    const size_t dot = p.find_last_of('.');
    if (dot == std::string::npos || (slash != std::string::npos && dot < slash) || dot + 1 == p.size()) return "";
    if (dot == (slash == std::string::npos ? 0 : slash + 1)) return ""; // ".hidden" has no extension (Path::extension)
    return p.substr(dot + 1);
}
bool is_int(const char *s) { if (!*s) return false; for (; *s; ++s) if (*s < '0' || *s > '9') return false; return true; }

Args parse_args(int argc, char **argv)
{
    Args a;
    std::vector<std::string> pos;
    for (int i = 1; i < argc; ++i) {
        const std::string s = argv[i];
        auto need = [&](const char *name) -> const char * {
            if (i + 1 >= argc) usage_error(std::string("a value is required for '") + name + "'");
            return argv[++i];
        };
        if (s == "-o" || s == "--overwrite") a.overwrite = true;
        else if (s == "-v" || s == "--verbose") a.verbose = true;
        else if (s == "-p" || s == "--patch-size") { const char *v = need("--patch-size"); if (!is_int(v)) usage_error("invalid value for '--patch-size'"); a.patch_size = atoi(v); }
        else if (s == "-b" || s == "--batch-size") { const char *v = need("--batch-size"); if (!is_int(v)) usage_error("invalid value for '--batch-size'"); a.batch_size = atoi(v); }
        else if (s == "-t" || s == "--thread-count") { const char *v = need("--thread-count"); if (!is_int(v)) usage_error("invalid value for '--thread-count'"); a.thread_count = atol(v); }
        else if (s == "-g" || s == "--gpus") {
            a.have_gpus = true;
            const char *v = need("--gpus");
            if (!is_int(v)) usage_error("invalid value for '--gpus'");
            a.gpus.push_back(atoi(v)); // clap 4: Vec<T> appends one value per occurrence (-g 0 -g 1)
        } else if (s == "-h" || s == "--help") {
            printf("Usage: nuclei-feature-extraction [OPTIONS] <GEOMETRY> <SLIDE> <OUTPUT> [FEATURE_SETS]...\n\n"
                   "Arguments:\n  <GEOMETRY>         Input geometry file (.geojson)\n  <SLIDE>            Input slide file\n"
                   "  <OUTPUT>           Output file (csv, parquet, pqt, json, ipc, feather)\n  [FEATURE_SETS]...  geometry color glcm glrlm gabor texture all\n\n"
                   "Options:\n  -o, --overwrite\n  -p, --patch-size <PATCH_SIZE>      [default: 64]\n  -t, --thread-count <THREAD_COUNT>\n"
                   "  -g, --gpus <GPUS>...\n  -b, --batch-size <BATCH_SIZE>      [default: 100]\n  -v, --verbose\n  -h, --help\n");
            exit(0);
        } else if (s.size() > 1 && s[0] == '-' && !is_int(s.c_str() + 1)) usage_error("unexpected argument '" + s + "'");
        else pos.push_back(s);
    }
    if (pos.size() < 3) usage_error("the following required arguments were not provided: <GEOMETRY> <SLIDE> <OUTPUT>");
    a.geometry = pos[0]; a.slide = pos[1]; a.output = pos[2];
    a.feature_sets.assign(pos.begin() + 3, pos.end());
    return a;
}

std::vector<uint8_t> read_file(const std::string &path)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) throw std::runtime_error("cannot open " + path);
    const std::streamsize n = f.tellg();
    f.seekg(0);
    std::vector<uint8_t> b((size_t)n);
    if (n && !f.read(reinterpret_cast<char *>(b.data()), n)) throw std::runtime_error("cannot read " + path);
    return b;
}

void append_float(std::string &s, float v, bool json)
{
    if (std::isnan(v)) { s += json ? "null" : "NaN"; return; }
    if (std::isinf(v)) { s += json ? "null" : (v < 0 ? "-inf" : "inf"); return; }
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof buf, v); // shortest round-trip
    s.append(buf, r.ptr);
}

void append_double(std::string &s, double v, bool json)
{
    if (std::isnan(v)) { s += json ? "null" : "NaN"; return; }
    if (std::isinf(v)) { s += json ? "null" : (v < 0 ? "-inf" : "inf"); return; }
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof buf, v); // shortest round-trip of the f64
    s.append(buf, r.ptr);
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

int main(int argc, char **argv)
{
    Args args = parse_args(argc, argv);
    // handle_verbose (args.rs:116-126)
    if (args.verbose) {
        g_debug = true;
        printf("Called Args :\n  geometry: %s\n  slide: %s\n  output: %s\n  patch_size: %d\n  batch_size: %d\n  overwrite: %d\n",
               args.geometry.c_str(), args.slide.c_str(), args.output.c_str(), args.patch_size, args.batch_size, (int)args.overwrite);
    }
    // validate_paths (args.rs:137-171)
    if (!exists(args.geometry)) { log_error("Geometry file does not exist : \"" + args.geometry + "\""); return 1; }
    if (!exists(args.slide)) { log_error("Slide file does not exist : \"" + args.slide + "\""); return 1; }
    if (exists(args.output) && !args.overwrite) {
        log_error("Output file already exists : \"" + args.output + "\"\nUse --overwrite to overwrite it");
        return 1;
    }
    const std::string oext = extension(args.output);
    if (oext.empty()) { log_error("Output file must have an extension"); return 1; }
    if (oext != "csv" && oext != "parquet" && oext != "pqt" && oext != "json" && oext != "ipc" && oext != "feather") {
        log_error("Unsupported output format. Please use one of the following : csv, parquet, json, ipc, feather");
        return 1;
    }
    // feature sets (args.rs:18-73)
    std::vector<const char *> names;
    for (auto &s : args.feature_sets) names.push_back(s.c_str());
    std::vector<int> sets(5 * std::max<size_t>(1, names.size()));
    const int n_sets = nfx_parse_feature_sets(names.data(), (int)names.size(), sets.data(), (int)sets.size());
    if (n_sets < 0) {
        for (auto &s : args.feature_sets) {
            const char *one[1] = {s.c_str()};
            int tmp[8];
            if (nfx_parse_feature_sets(one, 1, tmp, 8) < 0) usage_error("invalid value '" + s + "' for '[FEATURE_SETS]...': " + s + " is not a valid feature set");
        }
        return 2;
    }
    if (n_sets == 0) { log_error("No feature set selected (the reference panics on features[0], main.rs:76)"); return 1; }
    sets.resize((size_t)n_sets);
    if (args.patch_size < 1 || args.patch_size > NFX_MAX_PATCH_SIZE) { log_error("patch size must be in 1.." + std::to_string(NFX_MAX_PATCH_SIZE) + " for the MI355X engine"); return 1; }
    if (args.batch_size < 1) { log_error("batch size must be >= 1"); return 1; }
    // validate_gpu (args.rs:173-183): here a GPU is always required
    std::vector<int> gpus = args.gpus;
    if (!args.have_gpus) {
        log_info("no --gpus given: the reference would compute on the CPU; this engine has no CPU path and uses GPU 0");
        gpus = {0};
    }
    // the contexts (the HIP runtime's start-up: a third of a second) come up beside the loaders below
    std::vector<nfx_ctx *> ctxs;
    int bad_gpu = -1;
    std::thread ctx_thread([&]() {
        for (int g : gpus) {
            nfx_ctx *c = nullptr;
            if (nfx_ctx_create(g, &c) != NFX_OK) { bad_gpu = g; return; }
            ctxs.push_back(c);
        }
    });
    struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } ctx_joiner{ctx_thread};

    try {
        // load_input_image (main.rs:20-35) beside the geojson: the PNG's inflate is one serial stream (0.73 s for an 8192 x 8192 tile), the
        // geojson one streaming pass (0.31 s for 200 k features on one thread; large files on several) -- the wall time of the longer one
        const std::string sext = extension(args.slide);
        if (sext == "svs" || sext == "jpg" || sext == "jpeg") {
            log_error("." + sext + " slides need OpenSlide / a JPEG decoder, which this image does not have; convert the region to .png");
            return 1;
        }
        if (sext != "png") {
            log_error("Unsupported input format. Please use one of the following : svs, png, jpg, jpeg");
            return 1;
        }
        double t0 = now_s();
        ImageU8 img;
        std::string img_err;
        double img_s = 0.0;
        std::thread img_thread([&]() {
            try {
                const std::vector<uint8_t> raw = read_file(args.slide);
                img = decode_png(raw.data(), raw.size());
                img_s = now_s() - t0;
            } catch (const std::exception &e) { img_err = e.what(); }
        });
        log_info("Loading the geojson");
        Polygons poly;
        try {
            const std::vector<uint8_t> gj = read_file(args.geometry);
            // documents of 8 MB and more are parsed by several threads (-t / the hardware's count; geojson.hpp): same polygons, same errors
            GeoJsonReader reader(std::string_view(reinterpret_cast<const char *>(gj.data()), gj.size()), false, args.thread_count > 0 ? (int)std::min(args.thread_count, 32l) : 0);
            if (const char *mb = getenv("NFX_GEOJSON_PARALLEL_MIN_BYTES")) reader.parallel_min_bytes = (size_t)strtoull(mb, nullptr, 10); // (tests: the threaded path on a small file)
            poly = reader.parse();
            log_debug("geojson reader: " + std::to_string(reader.threads_used) + " thread(s)");
        } catch (...) { img_thread.join(); throw; }
        log_debug("Loaded geojson in " + std::to_string(now_s() - t0) + " s (" + std::to_string(poly.size()) + " features)");
        img_thread.join();
        if (!img_err.empty()) throw std::runtime_error(img_err);
        log_debug("Loaded image " + std::to_string(img.W) + "x" + std::to_string(img.H) + " in " + std::to_string(img_s) + " s (beside the geojson)");

        ctx_thread.join();
        if (bad_gpu >= 0) { log_error("GPU " + std::to_string(bad_gpu) + " does not exist"); return 1; }
        int F = 0;
        for (int s : sets) F += nfx_feature_set_num_columns(s);
        const int64_t n = (int64_t)poly.size();
        std::vector<float> rows((size_t)n * F), cen((size_t)n * 2);
        // The decoded slide and the row matrix are PINNED for the extraction (hipHostRegister: 3-20 ms per GB on this host, scripts/pin_probe.hip --
        // an order of magnitude cheaper than allocating pinned memory, 185 ms per GB): the slide's bands then travel asynchronously, beside the
        // kernels, and the feature kernels write the rows straight into `rows` (nfx.h: nfx_extract_tile_streamed_v2, rows_direct) instead of
        // into a device buffer that is copied home chunk by chunk.  NFX_CLI_PIN=0 leaves both pageable (the round 1-4 behaviour).
        struct Registered { // unregistered before the vectors go
            void *p = nullptr;
            explicit Registered(void *q, size_t bytes) { if (q && bytes && nfx_host_register(q, bytes) == NFX_OK) p = q; }
            ~Registered() { if (p) nfx_host_unregister(p); }
            Registered(const Registered &) = delete;
            Registered &operator=(const Registered &) = delete;
        };
        const bool pin = !([] { const char *e = getenv("NFX_CLI_PIN"); return e && atoi(e) == 0; }());
        const double t_pin = now_s();
        Registered pinned_img(pin ? img.data.data() : nullptr, img.data.size()), pinned_rows(pin ? rows.data() : nullptr, rows.size() * sizeof(float));
        if (pin) log_debug(std::string("Pinned the image (") + (pinned_img.p ? "ok" : "refused: pageable copies") + ") and the rows (" + (pinned_rows.p ? "ok" : "refused: staged on the device") +
                           ") in " + std::to_string(now_s() - t_pin) + " s");
        // columns the reference's frame holds as f64 (shape.rs:116-127: geometric_features' own return values): kept at that
        // width in every output format.  wide[j] = column of `wide_rows` that feature column j comes from, -1 for f32 columns
        std::vector<int> wide((size_t)F, -1);
        bool any_wide = false;
        {
            int col = 0;
            for (int s : sets) {
                int k = 0;
                for (int j = 0; j < nfx_feature_set_num_columns(s); ++j)
                    if (nfx_feature_set_column_dtype(s, j) == NFX_DTYPE_F64) { wide[(size_t)(col + j)] = k++; any_wide = true; } // a repeated geometry set shares the seven
                col += nfx_feature_set_num_columns(s);
            }
        }
        const int W = 7;
        std::vector<double> wide_rows(any_wide ? (size_t)n * W : 0);
        log_info("Extracting features");
        t0 = now_s();
        // index shards on reference-batch boundaries, one per GPU
        const int R = (int)ctxs.size();
        const int64_t B = args.batch_size, nb = (n + B - 1) / B, per = (nb + R - 1) / R;
        std::vector<std::thread> th;
        std::vector<std::string> errs((size_t)R);
        for (int r = 0; r < R; ++r) {
            th.emplace_back([&, r]() {
                const int64_t i0 = std::min(r * per * B, n), i1 = std::min((r + 1) * per * B, n);
                if (i1 <= i0) return;
                std::vector<uint32_t> off((size_t)(i1 - i0 + 1));
                const uint32_t base = poly.ring_offsets[(size_t)i0];
                for (int64_t i = i0; i <= i1; ++i) off[(size_t)(i - i0)] = poly.ring_offsets[(size_t)i] - base;
                nfx_stream_stats ss{};
                ss.struct_size = sizeof ss;
                const nfx_status st = nfx_extract_tile_streamed_v2(ctxs[(size_t)r], sets.data(), n_sets, i1 - i0, off.back(), off.data(),
                                                                poly.ring_xy.data() + 2 * (size_t)base, img.data.data(), img.H, img.W, args.patch_size,
                                                                args.batch_size, 0, 0, 0, rows.data() + (size_t)i0 * F, cen.data() + 2 * (size_t)i0, &ss);
                if (st != NFX_OK) { errs[(size_t)r] = nfx_last_error(ctxs[(size_t)r]); return; }
                log_debug("GPU " + std::to_string(r) + ": " + std::to_string(i1 - i0) + " nuclei in " + std::to_string(ss.chunks) + " chunks, " +
                          std::to_string(ss.blocks_copied) + " of " + std::to_string(ss.blocks_total) + " blocks (" +
                          std::to_string(ss.image_bytes_h2d >> 20) + " MiB) to the device" + (ss.image_pinned ? ", image pinned" : ", image pageable") +
                          (ss.rows_direct ? ", rows written home by the kernels" : ", rows staged and copied"));
                if (any_wide) { // the polygon columns once more, unrounded (preprocess_polygon + geometric_features::*, utils.rs:54-74, shape.rs:89-97)
                    std::vector<float> centred(2 * (size_t)off.back());
                    nfx_status s2 = nfx_preprocess_polygons(ctxs[(size_t)r], i1 - i0, off.back(), off.data(), poly.ring_xy.data() + 2 * (size_t)base,
                                                            nullptr, centred.data(), NFX_MEM_HOST);
                    if (s2 == NFX_OK)
                        s2 = nfx_polygon_features_f64(ctxs[(size_t)r], i1 - i0, off.back(), off.data(), centred.data(), wide_rows.data() + (size_t)i0 * W, NFX_MEM_HOST);
                    if (s2 != NFX_OK) errs[(size_t)r] = nfx_last_error(ctxs[(size_t)r]);
                }
            });
        }
        for (auto &t : th) t.join();
        for (auto &e : errs)
            if (!e.empty()) { log_error("extraction failed: " + e); return 1; }
        log_info("Extracted features for " + std::to_string(n) + "/" + std::to_string(n) + " patches");
        log_debug("Extracted in " + std::to_string(now_s() - t0) + " s");

        // writers (main.rs:161-189)
        std::vector<std::string> header{"centroid"};
        char name[256];
        for (int s : sets)
            for (int j = 0; j < nfx_feature_set_num_columns(s); ++j) {
                nfx_feature_set_column_name(s, j, name, sizeof name);
                header.push_back(name);
            }
        FILE *fo = fopen(args.output.c_str(), "wb");
        if (!fo) { log_error("cannot create " + args.output); return 1; }
        t0 = now_s();
        const bool json = oext == "json";
        if (oext == "csv" || json) {
            // text rows are formatted by every host thread (shortest round-trip floats are ~30 ns each: 418 columns of ten
            // million nuclei would keep one core busy for two minutes) and written in order, a block of rows at a time
            if (!json) {
                std::string line;
                for (size_t j = 0; j < header.size(); ++j) { if (j) line += ','; line += header[j]; }
                line += '\n';
                fwrite(line.data(), 1, line.size(), fo);
            }
            auto format_rows = [&](int64_t i0, int64_t i1, std::string &text) {
                char key[128];
                for (int64_t i = i0; i < i1; ++i) {
                    nfx_centroid_key(cen[2 * (size_t)i], cen[2 * (size_t)i + 1], key, sizeof key);
                    if (json) { // json lines: one object per row
                        text += "{\"centroid\":\""; text += key; text += '"';
                        for (int j = 0; j < F; ++j) {
                            text += ",\""; text += header[(size_t)j + 1]; text += "\":";
                            if (wide[(size_t)j] >= 0) append_double(text, wide_rows[(size_t)i * W + wide[(size_t)j]], true);
                            else append_float(text, rows[(size_t)i * F + j], true);
                        }
                        text += "}\n";
                    } else {
                        text += '"'; text += key; text += '"'; // the key contains a comma
                        for (int j = 0; j < F; ++j) {
                            text += ',';
                            if (wide[(size_t)j] >= 0) append_double(text, wide_rows[(size_t)i * W + wide[(size_t)j]], false);
                            else append_float(text, rows[(size_t)i * F + j], false);
                        }
                        text += '\n';
                    }
                }
            };
            const unsigned hw = std::thread::hardware_concurrency();
            const int workers = (int)std::max(1l, std::min<long>(args.thread_count > 0 ? args.thread_count : (hw ? hw : 1), 64));
            const int64_t block = 4096;
            std::vector<std::string> text((size_t)workers);
            for (int64_t base = 0; base < n; base += block * workers) {
                std::vector<std::thread> fm;
                for (int w = 0; w < workers; ++w) {
                    const int64_t i0 = std::min(base + w * block, n), i1 = std::min(i0 + block, n);
                    text[(size_t)w].clear();
                    if (i0 < i1) fm.emplace_back(format_rows, i0, i1, std::ref(text[(size_t)w]));
                }
                for (auto &t : fm) t.join();
                for (int w = 0; w < workers; ++w)
                    if (!text[(size_t)w].empty() && fwrite(text[(size_t)w].data(), 1, text[(size_t)w].size(), fo) != text[(size_t)w].size()) {
                        log_error("short write to " + args.output);
                        return 1;
                    }
            }
        } else { // parquet | pqt | ipc | feather (main.rs:169-180)
            char key[128];
            std::vector<std::string> keys((size_t)n);
            for (int64_t i = 0; i < n; ++i) {
                nfx_centroid_key(cen[2 * (size_t)i], cen[2 * (size_t)i + 1], key, sizeof key);
                keys[(size_t)i] = key;
            }
            FeatureTable table{&header, &keys, rows.data(), n, F};
            if (any_wide) { table.wide = wide.data(); table.wide_rows = wide_rows.data(); table.W = W; }
            if (oext == "parquet" || oext == "pqt") write_parquet(fo, table);
            else write_arrow_ipc(fo, table);
        }
        fclose(fo);
        log_debug("Wrote " + args.output + " in " + std::to_string(now_s() - t0) + " s");
    } catch (const std::exception &e) {
        log_error(e.what());
        return 1;
    }
    for (auto c : ctxs) nfx_ctx_destroy(c);
    return 0;
}
