// Thread scaling of the host's GeoJSON reader (host/geojson.hpp) on a synthetic nucleus collection.
//   g++ -O2 -std=c++17 -pthread -o /tmp/geojson_bench scripts/geojson_bench.cpp && /tmp/geojson_bench [features]
// Writes the collection into memory (12-48 vertices per ring, coordinates as a segmentation tool prints them), parses it with
// 1, 2, 4, 8, 16 and 32 threads and prints seconds, GB/s and a hash of the polygons (the same for every thread count).
#include "../nuclei-feature-extraction_amd/host/geojson.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

int main(int argc, char **argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 200000;
    std::string text = "{\"type\":\"FeatureCollection\",\"features\":[";
    unsigned long long s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    char buf[96];
    for (long i = 0; i < n; ++i) {
        if (i) text += ',';
        text += "{\"type\":\"Feature\",\"bbox\":[0,0,1,1],\"geometry\":{\"type\":\"Polygon\",\"coordinates\":[[";
        const int V = 12 + (int)(rnd() % 37);
        for (int v = 0; v < V; ++v) {
            snprintf(buf, sizeof buf, "%s[%.9g,%.9g]", v ? "," : "", (double)(float)((rnd() % 40000000) / 1000.0), (double)(float)((rnd() % 40000000) / 1000.0));
            text += buf;
        }
        text += "]]},\"properties\":{\"classification\":{\"name\":\"nucleus \\\"x\\\" ]\",\"colorRGB\":-1}}}";
    }
    text += "]}";
    printf("%ld features, %.0f MB\n", n, text.size() / 1e6);
    for (int th : {1, 2, 4, 8, 16, 32}) {
        const auto t0 = std::chrono::steady_clock::now();
        nfxhost::GeoJsonReader r(text, false, th);
        const nfxhost::Polygons p = r.parse();
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        unsigned long long h = 1469598103934665603ull;
        for (float v : p.ring_xy) { unsigned u; memcpy(&u, &v, 4); h = (h ^ u) * 1099511628211ull; }
        printf("threads %2d (used %2d): %.3f s, %.2f GB/s, %zu features, hash %llx\n", th, r.threads_used, dt, text.size() / dt / 1e9, p.size(), h);
    }
    return 0;
}
