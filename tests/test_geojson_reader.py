"""The C++ host's GeoJSON reader (host/geojson.hpp) against the reference's serde structs (geojson.rs:8-24): required and
ignored fields, number forms, nesting depth, and the error cases serde would reject.  A small driver is compiled around
the header; no GPU involved."""
import json
import os
import subprocess

import numpy as np
import pytest

# the host code under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only; any report aborts the driver: the test fails)
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "nuclei-feature-extraction_amd", "host", "geojson.hpp")

DRIVER = r"""
#include "%s"
#include <cstdio>
#include <fstream>
#include <iterator>
int main(int argc, char **argv)
{
    std::ifstream f(argv[1], std::ios::binary);
    std::string text((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    try {
        nfxhost::GeoJsonReader reader(text, false, argc > 2 ? atoi(argv[2]) : 1);
        if (argc > 2) reader.parallel_min_bytes = 0;   // the threaded path even on a small document
        nfxhost::Polygons p = reader.parse();
        if (argc > 2) fprintf(stderr, "threads_used %%d\n", reader.threads_used);
        printf("%%zu\n", p.size());
        for (size_t i = 0; i <= p.size(); ++i) printf("%%u ", p.ring_offsets[i]);
        printf("\n");
        for (float v : p.ring_xy) printf("%%.9g ", v);
        printf("\n");
    } catch (const std::exception &e) {
        fprintf(stderr, "%%s\n", e.what());
        return 3;
    }
    return 0;
}
"""


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    d = tmp_path_factory.mktemp("geojson")
    src = d / "driver.cpp"
    src.write_text(DRIVER % HEADER)
    exe = d / "driver"
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-pthread", *SAN, "-o", str(exe), str(src)], check=True)
    return exe


def parse(driver, tmp_path, text, threads=None):
    (tmp_path / "a.geojson").write_text(text)
    r = subprocess.run([str(driver), str(tmp_path / "a.geojson")] + ([str(threads)] if threads else []), capture_output=True, text=True)
    if r.returncode != 0:
        return r.returncode, r.stderr
    lines = r.stdout.split("\n")
    n = int(lines[0])
    off = np.array(lines[1].split(), np.int64)
    xy = np.array(lines[2].split(), np.float32).reshape(-1, 2)
    assert len(off) == n + 1
    return 0, (off, xy)


def feature(ring, extra_rings=(), **more):
    f = {"type": "Feature", "bbox": [0, 0, 1, 1], "geometry": {"type": "Polygon", "coordinates": [ring, *extra_rings]}}
    f.update(more)
    return f


def test_well_formed_collection(driver, tmp_path):
    r1 = [[10.5, 20.25], [30, 20.25], [30, 40], [10.5, 20.25]]
    r2 = [[1e2, 2.5E1, 7.0], [1.0e-1, -3], [0.1, 0.2]]                       # exponents; a third coordinate is ignored
    hole = [[11, 21], [12, 21], [12, 22], [11, 21]]
    doc = {"name": "x", "crs": {"deep": [{"er": [1, 2, {"k": None}]}, True, False]},
           "features": [feature(r1, [hole], properties={"classification": {"name": "tumor \"cell\" \\ é", "colorRGB": -1}}, id="a"),
                        feature(r2), feature([])],
           "type": "FeatureCollection"}
    rc, (off, xy) = parse(driver, tmp_path, json.dumps(doc, indent=2))
    assert rc == 0
    assert off.tolist() == [0, 4, 7, 7]                                       # outer rings only (utils.rs:55); an empty ring is kept
    want = np.array([p[:2] for p in r1 + r2], np.float64).astype(np.float32)  # f64 -> f32 narrowing, as serde does for f32 fields
    assert np.array_equal(xy, want)
    rc, (off, xy) = parse(driver, tmp_path, '{"features":[]}')
    assert rc == 0 and off.tolist() == [0]
    # key order inside a feature does not matter
    rc, (off, xy) = parse(driver, tmp_path, '{"features":[{"geometry":{"coordinates":[[[1,2],[3,4]]],"type":"Polygon"},"bbox":[]}]}')
    assert rc == 0 and off.tolist() == [0, 2] and xy.tolist() == [[1, 2], [3, 4]]


@pytest.mark.parametrize("text,message", [
    ('{"type":"FeatureCollection"}', "missing field `features`"),
    ('{"features":[{"geometry":{"type":"Polygon","coordinates":[[[1,2]]]}}]}', "missing field `bbox`"),             # geojson.rs:17
    ('{"features":[{"bbox":[0,0,1,1]}]}', "missing field `geometry`"),
    ('{"features":[{"bbox":[],"geometry":{"coordinates":[[[1,2]]]}}]}', "missing field `type`"),
    ('{"features":[{"bbox":[],"geometry":{"type":"Polygon"}}]}', "missing field `coordinates`"),
    ('{"features":[{"bbox":[],"geometry":{"type":"Polygon","coordinates":[]}}]}', "no ring"),                         # utils.rs:55 would panic
    ('{"features":[{"bbox":[],"geometry":{"type":"Polygon","coordinates":[[[1]]]}}]}', "two coordinates"),           # utils.rs:57-58 would panic
    ('{"features":[{"bbox":[],"geometry":{"type":"MultiPolygon","coordinates":[[[[1,2],[3,4]]]]}}]}', "expected a number"),  # four levels: not Vec<Vec<Vec<f32>>>
    ('{"features":[{"bbox":[],"geometry":{"type":"Point","coordinates":[1,2]}}]}', "expected"),
    ('{"features":[{"bbox":["a"],"geometry":{"type":"Polygon","coordinates":[[[1,2]]]}}]}', "expected a number"),
    ('{"features":[{"bbox":[],"geometry":{"type":"Polygon","coordinates":[[[1,2]]]}}', "expected"),                   # truncated
    ('[1,2,3]', "expected"),
    ('', "expected"),
])
def test_rejected_documents(driver, tmp_path, text, message):
    rc, err = parse(driver, tmp_path, text)
    assert rc == 3 and message in err, err


def test_threaded_reader_gives_the_serial_readers_result(driver, tmp_path):
    """Large collections are parsed by several threads (chunk scan for string state and bracket depth, element offsets, runs of
    features parsed by the same code): polygons identical to the serial reader's, whatever sits in the ignored fields -- brackets,
    braces, quotes and backslashes inside strings, nested arrays of objects -- and however the chunk boundaries fall."""
    rng = np.random.default_rng(5)
    nasty = ['}{][', 'a "quoted" \\ name', '\\', '\\\\"', '{"features":[', 'tail\\', '[[[', 'é\u00e9 \n']
    feats = []
    for i in range(1500):
        nv = int(rng.integers(0, 9))
        ring = [[float(np.float32(rng.uniform(-1e4, 1e4))), float(np.float32(rng.uniform(-1e4, 1e4)))] for _ in range(nv)]
        props = {"name": nasty[i % len(nasty)], "list": [{"k": nasty[(i + 3) % len(nasty)], "v": [i, [i, {"d": None}]]}], nasty[(i + 1) % len(nasty)]: i}
        feats.append(feature(ring, [ring[:2]] if i % 7 == 0 else (), properties=props, id=nasty[(i + 5) % len(nasty)]))
    doc = {"type": "FeatureCollection", "crs": {"x": [{"y": "]"}]}, "features": feats, "trailer": [{"z": "{"}]}
    for text in (json.dumps(doc), json.dumps(doc, indent=1)):
        rc1, (off1, xy1) = parse(driver, tmp_path, text)
        assert rc1 == 0 and len(off1) == 1501
        for threads in (2, 5, 16, 31):
            (tmp_path / "a.geojson").write_text(text)
            r = subprocess.run([str(driver), str(tmp_path / "a.geojson"), str(threads)], capture_output=True, text=True)
            assert r.returncode == 0 and f"threads_used {threads}" in r.stderr, r.stderr[-300:]
            lines = r.stdout.split("\n")
            assert np.array_equal(np.array(lines[1].split(), np.int64), off1)
            assert np.array_equal(np.array(lines[2].split(), np.float32).reshape(-1, 2), xy1)


@pytest.mark.parametrize("text,message", [
    ('{"features":[{"bbox":[],"geometry":{"type":"Polygon","coordinates":[[[1,2]]]}},{"geometry":{"type":"Polygon","coordinates":[[[1,2]]]}}]}', "missing field `bbox`"),
    ('{"features":[{"bbox":[],"geometry":{"type":"Polygon","coordinates":[[[1,2]]]}},7]}', "expected `{`"),
    ('{"features":[{"bbox":[],"geometry":{"type":"Polygon","coordinates":[[[1,2]]]}} {"bbox":[],"geometry":{"type":"Polygon","coordinates":[[[1,2]]]}}]}', "expected `]`"),
    ('{"features":[{"bbox":[],"geometry":{"type":"Polygon","coordinates":[[[1,2]]]}},]}', "expected `{`"),
    ('{"features":[x{"bbox":[],"geometry":{"type":"Polygon","coordinates":[[[1,2]]]}}]}', "expected `{`"),
    ('{"features":[{"bbox":[],"geometry":{"type":"Polygon","coordinates":[[[1,2]]]}}', "expected"),
])
def test_threaded_reader_reports_errors_as_the_serial_reader_does(driver, tmp_path, text, message):
    """Anything the threaded attempt cannot make sense of is parsed again serially: same message, same byte offset."""
    rc1, err1 = parse(driver, tmp_path, text)
    rc4, err4 = parse(driver, tmp_path, text, threads=4)
    assert rc1 == 3 and message in err1, err1
    assert rc4 == 3 and err4.strip().splitlines()[-1] == err1.strip().splitlines()[-1], (err1, err4)


def test_threaded_reader_under_thread_sanitizer(tmp_path_factory, tmp_path):
    """The four passes share nothing but read-only input, per-thread outputs and the barrier: ThreadSanitizer (CPU build) agrees."""
    d = tmp_path_factory.mktemp("geojson_tsan")
    src = d / "driver.cpp"
    src.write_text(DRIVER % HEADER)
    exe = d / "driver_tsan"
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-pthread", "-fsanitize=thread", "-o", str(exe), str(src)], check=True)
    feats = [feature([[float(i), 1.5], [2.0, float(i)], [float(i), 1.5]], properties={"n": 'a "b" ] \\'}) for i in range(400)]
    (tmp_path / "a.geojson").write_text(json.dumps({"features": feats}))
    r = subprocess.run([str(exe), str(tmp_path / "a.geojson"), "6"], capture_output=True, text=True)
    assert r.returncode == 0 and "threads_used 6" in r.stderr and "ThreadSanitizer" not in r.stderr, r.stderr[-600:]
    assert r.stdout.split("\n")[0] == "400"
