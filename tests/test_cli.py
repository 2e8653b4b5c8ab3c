"""The C++ host CLI (nuclei-feature-extraction_amd/host/cli.cpp) behind the reference's command line
(args.rs:76-113, main.rs:110-190).  CPU: argument surface and validation.  GPU: BASELINE.json configs[0]
-- 1k synthetic nuclei GeoJSON + 2048^2 PNG tile, geometry set -- end to end against the oracle."""
import csv
import importlib
import json
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

import _oracle as orc
import _parity as par

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "nuclei-feature-extraction_amd", "bin", "nuclei-feature-extraction")


def write_png(path, img):
    """img: [3,H,W] u8 -> 8-bit RGB PNG, filter 'up' on odd rows to exercise the unfilter code."""
    _, H, W = img.shape
    rgb = np.ascontiguousarray(img.transpose(1, 2, 0))
    raw = bytearray()
    prev = np.zeros(W * 3, np.uint8)
    for y in range(H):
        line = rgb[y].reshape(-1)
        if y % 2:
            raw.append(2)
            raw += ((line.astype(np.int16) - prev.astype(np.int16)) & 0xFF).astype(np.uint8).tobytes()
        else:
            raw.append(0)
            raw += line.tobytes()
        prev = line

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 8, 2, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(bytes(raw), 1)) + chunk(b"IEND", b""))


def write_geojson(path, off, xy, shuffle_keys=False):
    feats = []
    for i in range(len(off) - 1):
        ring = xy[off[i]:off[i + 1]].astype(np.float64)
        bbox = [float(ring[:, 0].min()), float(ring[:, 1].min()), float(ring[:, 0].max()), float(ring[:, 1].max())] if len(ring) else [0, 0, 0, 0]
        geom = {"type": "Polygon", "coordinates": [[[float(x), float(y)] for x, y in ring], [[0.0, 0.0], [1.0, 0.0], [0.0, 1.0]]]}
        f = {"type": "Feature", "id": f"n{i}", "bbox": bbox, "properties": {"objectType": "detection", "nested": {"a": [1, 2, {"b": None}]}}, "geometry": geom}
        if shuffle_keys and i % 2:
            f = dict(reversed(list(f.items())))
        feats.append(f)
    with open(path, "w") as fh:
        json.dump({"type": "FeatureCollection", "features": feats}, fh)


def run(*args):
    return subprocess.run([CLI, *args], capture_output=True, text=True)


def test_cli_exists_and_prints_usage():
    assert os.path.exists(CLI), "build it: make -C nuclei-feature-extraction_amd/csrc"
    r = run("--help")
    assert r.returncode == 0 and "<GEOMETRY> <SLIDE> <OUTPUT>" in r.stdout and "--batch-size" in r.stdout


def test_cli_argument_validation(tmp_path):
    g, s, o = tmp_path / "a.geojson", tmp_path / "a.png", tmp_path / "o.csv"
    assert run().returncode == 2                                    # clap: missing positionals
    r = run(str(g), str(s), str(o), "geometry")
    assert r.returncode == 1 and "Geometry file does not exist" in r.stderr          # args.rs:138-141
    g.write_text("{}")
    r = run(str(g), str(s), str(o), "geometry")
    assert r.returncode == 1 and "Slide file does not exist" in r.stderr             # args.rs:142-145
    s.write_bytes(b"")
    o.write_text("x")
    r = run(str(g), str(s), str(o), "geometry")
    assert r.returncode == 1 and "Output file already exists" in r.stderr            # args.rs:147-153
    r = run(str(g), str(s), str(tmp_path / "noext"), "geometry")
    assert r.returncode == 1 and "must have an extension" in r.stderr                # args.rs:158-161
    r = run(str(g), str(s), str(tmp_path / "o.txt"), "geometry")
    assert r.returncode == 1 and "Unsupported output format" in r.stderr             # args.rs:162-165
    r = run(str(g), str(s), str(tmp_path / "o2.csv"), "shape")
    assert r.returncode == 2 and "is not a valid feature set" in r.stderr            # args.rs:32


@pytest.mark.gpu
def test_config1_end_to_end_csv(nfx, tmp_path):
    """BASELINE.json configs[0]: 1k synthetic nuclei GeoJSON + 2048^2 tile, geometry set, ps 64, batch 100."""
    synth = importlib.import_module("nuclei-feature-extraction_amd.synth")
    n, ps, H, W = 1000, 64, 2048, 2048
    seed = synth.SEED_BASE + 0
    cx, cy = synth.centres(seed, 0, n, ps, W, H)
    off, xy = synth.polygons(seed, 0, n, ps, cx, cy)
    img = synth.tile(seed, H, W, off[:201], xy, ps)          # blobs for the first 200 nuclei are enough
    write_png(tmp_path / "tile.png", img)
    write_geojson(tmp_path / "nuclei.geojson", off, xy, shuffle_keys=True)
    out = tmp_path / "features.csv"
    r = run("-v", "-g", "0", str(tmp_path / "nuclei.geojson"), str(tmp_path / "tile.png"), str(out), "geometry", "color")
    assert r.returncode == 0, r.stderr
    rows = list(csv.reader(open(out)))
    assert rows[0] == nfx.all_column_names([0, 1])
    assert len(rows) == n + 1
    got = np.array([[float(v) for v in r_[1:]] for r_ in rows[1:]], np.float32)
    patches = np.stack([orc.crop_patch(img, *orc.preprocess_polygon(xy[off[i]:off[i + 1]])[0], ps) for i in range(n)])
    want, cen = orc.extract([0, 1], off, xy, patches, ps, batch_size=100)
    assert [r_[0] for r_ in rows[1:]] == nfx.centroids_to_key_strings(cen)           # utils.rs:226-232
    par.assert_rows_close(got, want, par.column_scales(nfx, [0, 1], ps), what="cli csv")
    # the same file through the threaded GeoJSON reader (large collections take it by themselves; here it is forced): the same rows
    out_t = tmp_path / "features_threaded.csv"
    env = dict(os.environ, NFX_GEOJSON_PARALLEL_MIN_BYTES="0")
    r = subprocess.run([CLI, "-v", "-t", "5", "-g", "0", str(tmp_path / "nuclei.geojson"), str(tmp_path / "tile.png"), str(out_t), "geometry", "color"],
                       capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "geojson reader: 5 thread(s)" in r.stderr, r.stderr[-400:]
    assert open(out_t).read() == open(out).read()
    # refuses to overwrite without -o, accepts with -o (args.rs:147-153)
    assert run(str(tmp_path / "nuclei.geojson"), str(tmp_path / "tile.png"), str(out), "geometry").returncode == 1
    assert run("-o", "-p", "32", "-b", "50", str(tmp_path / "nuclei.geojson"), str(tmp_path / "tile.png"), str(out), "GEOMETRY").returncode == 0
    assert len(open(out).readline().split(",")) == 13


@pytest.mark.gpu
def test_two_contexts_in_one_process_match_one(nfx, tmp_path):
    """-g 0 -g 0: two engine contexts driven from two host threads (the reference calls its feature sets from every
    rayon worker, main.rs:146-151); shards are cut on batch boundaries, so the file equals the single-context one."""
    synth = importlib.import_module("nuclei-feature-extraction_amd.synth")
    n, ps, H, W = 330, 48, 512, 512
    seed = synth.SEED_BASE + 9
    cx, cy = synth.centres(seed, 0, n, ps, W, H)
    off, xy = synth.polygons(seed, 0, n, ps, cx, cy)
    write_png(tmp_path / "t.png", synth.tile(seed, H, W, off[:41], xy, ps))
    write_geojson(tmp_path / "n.geojson", off, xy)
    a, b = tmp_path / "one.csv", tmp_path / "two.csv"
    common = ["-p", str(ps), "-b", "40", str(tmp_path / "n.geojson"), str(tmp_path / "t.png")]
    assert run("-g", "0", *common, str(a), "color", "geometry").returncode == 0
    assert run("-g", "0", "-g", "0", *common, str(b), "color", "geometry").returncode == 0
    assert open(a).read() == open(b).read()
    # every set: both contexts build their device tables (Gabor bank, stain look-up, GLCM tables) at the same moment
    assert run("-o", "-g", "0", *common, str(a), "all").returncode == 0
    assert run("-o", "-g", "0", "-g", "0", *common, str(b), "all").returncode == 0
    assert open(a).read() == open(b).read()


@pytest.mark.gpu
def test_every_output_format_holds_the_same_table(nfx, tmp_path):
    """main.rs:165-189: the writer is picked by extension (csv, parquet | pqt, json, ipc | feather); all six files carry
    the same centroid keys and, in the binary formats, the same bits -- float32 columns, and float64 for the seven polygon
    columns the reference's frame holds at that width (shape.rs:116-127)."""
    pa = pytest.importorskip("pyarrow")
    import pyarrow.ipc
    import pyarrow.parquet
    synth = importlib.import_module("nuclei-feature-extraction_amd.synth")
    n, ps, H, W = 150, 32, 384, 384
    seed = synth.SEED_BASE + 11
    cx, cy = synth.centres(seed, 0, n, ps, W, H)
    off, xy = synth.polygons(seed, 0, n, ps, cx, cy)
    write_png(tmp_path / "t.png", synth.tile(seed, H, W, off[:51], xy, ps))
    write_geojson(tmp_path / "n.geojson", off, xy)
    for ext in ("csv", "parquet", "pqt", "json", "ipc", "feather"):
        r = run("-g", "0", "-p", str(ps), str(tmp_path / "n.geojson"), str(tmp_path / "t.png"), str(tmp_path / f"o.{ext}"), "all")
        assert r.returncode == 0, r.stderr
    names = nfx.all_column_names([0, 1, 2, 3, 4])
    rows = list(csv.reader(open(tmp_path / "o.csv")))
    assert rows[0] == names and len(rows) == n + 1
    keys = [r_[0] for r_ in rows[1:]]
    text = np.array([[float(v) for v in r_[1:]] for r_ in rows[1:]], np.float64)   # shortest round-trip text -> the same f32 / f64
    dtypes = ["f32"] * 0
    for s_ in range(5):
        dtypes += nfx.column_dtypes(s_)
    wide = [j for j, d in enumerate(dtypes) if d == "f64"]
    assert [names[1 + j] for j in wide] == ["area", "perimeter", "equivalent_perimeter", "compacity", "convex_hull_area", "convex_deffect",
                                            "convex_perimeter"]
    tables = {
        "parquet": pa.parquet.read_table(str(tmp_path / "o.parquet")),
        "pqt": pa.parquet.read_table(str(tmp_path / "o.pqt")),
        "ipc": pa.ipc.open_file(str(tmp_path / "o.ipc")).read_all(),
        "feather": pa.ipc.open_file(str(tmp_path / "o.feather")).read_all(),
    }
    for ext, t in tables.items():
        assert t.column_names == names, ext
        assert t.column(0).to_pylist() == keys, ext
        for j in range(len(names) - 1):
            col = t.column(j + 1).to_numpy()
            assert col.dtype == (np.float64 if j in wide else np.float32), (ext, names[j + 1])
            want = text[:, j] if j in wide else text[:, j].astype(np.float32)   # the CSV's shortest round-trip text parses back to the same number
            np.testing.assert_array_equal(col, want, err_msg=f"{ext} {names[j + 1]}")
    per = tables["parquet"].column("perimeter").to_numpy()
    assert (per != per.astype(np.float32).astype(np.float64)).mean() > 0.9       # genuinely f64 values, not widened f32
    lines = (tmp_path / "o.json").read_text().strip().splitlines()
    assert [json.loads(ln)["centroid"] for ln in lines] == keys


@pytest.mark.gpu
def test_geojson_requires_bbox(tmp_path):
    (tmp_path / "g.geojson").write_text('{"features":[{"geometry":{"type":"Polygon","coordinates":[[[1,1],[5,1],[5,5],[1,1]]]}}]}')
    write_png(tmp_path / "t.png", np.full((3, 16, 16), 200, np.uint8))
    r = run(str(tmp_path / "g.geojson"), str(tmp_path / "t.png"), str(tmp_path / "o.csv"), "geometry")
    assert r.returncode == 1 and "missing field `bbox`" in r.stderr                  # geojson.rs:17


def test_filter_geojson_reproduces_the_reference_utility(tmp_path):
    """src/filter_geojson.rs:16-67 incl. its bbox order quirk (read [l,t,r,d], written [t,l,d,r])."""
    FILTER = os.path.join(ROOT, "nuclei-feature-extraction_amd", "bin", "filter-geojson")
    feats = [
        {"bbox": [110.0, 220.0, 130.0, 240.0], "geometry": {"type": "Polygon", "coordinates": [[[110.0, 220.0], [130.0, 220.0], [120.5, 240.0], [110.0, 220.0]], [[1.0, 2.0], [3.0, 4.0], [5.0, 6.0]]]}, "properties": {"x": 1}},
        {"bbox": [90.0, 220.0, 130.0, 240.0], "geometry": {"type": "Polygon", "coordinates": [[[90.0, 220.0], [130.0, 240.0], [90.0, 220.0]]]}},     # left < offset_x
        {"bbox": [110.0, 220.0, 330.0, 240.0], "geometry": {"type": "Polygon", "coordinates": [[[110.0, 220.0], [330.0, 240.0], [110.0, 220.0]]]}},   # right > offset+size
        {"bbox": [1.0, 2.0, 3.0], "geometry": {"type": "Polygon", "coordinates": [[[1.0, 2.0]]]}},                                                     # not 4 numbers
    ]
    (tmp_path / "in.geojson").write_text(json.dumps({"type": "FeatureCollection", "features": feats}))
    r = subprocess.run([FILTER, str(tmp_path / "in.geojson"), str(tmp_path / "out.geojson"), "100", "200", "200", "100"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "No bbox" in r.stdout
    out = json.loads((tmp_path / "out.geojson").read_text())
    assert list(out.keys()) == ["features"] and len(out["features"]) == 1
    f = out["features"][0]
    assert f["bbox"] == [20.0, 10.0, 40.0, 30.0]                       # [top-oy, left-ox, down-oy, right-ox]
    assert f["geometry"]["type"] == "Polygon"
    assert f["geometry"]["coordinates"][0] == [[10.0, 20.0], [30.0, 20.0], [20.5, 40.0], [10.0, 20.0]]
    assert f["geometry"]["coordinates"][1] == [[1.0, 2.0], [3.0, 4.0], [5.0, 6.0]]   # inner rings are not shifted (filter_geojson.rs:54)
    assert subprocess.run([FILTER, "a", "b"], capture_output=True).returncode == 2


RUNNER = os.path.join(ROOT, "nuclei-feature-extraction_amd", "bin", "run.sh")


def test_folder_driver_usage_and_empty_folder(tmp_path):
    """run.sh:5-38: three folders are required; a folder without .geojson files is not an error."""
    r = subprocess.run([RUNNER], capture_output=True, text=True)
    assert r.returncode == 1 and "Usage:" in r.stdout and "<geojson_folder>" in r.stdout
    (tmp_path / "g").mkdir(); (tmp_path / "s").mkdir()
    r = subprocess.run([RUNNER, str(tmp_path / "g"), str(tmp_path / "s"), str(tmp_path / "out" / "deep")], capture_output=True, text=True)
    assert r.returncode == 0 and (tmp_path / "out" / "deep").is_dir()          # mkdir -p (run.sh:41)


@pytest.mark.gpu
def test_folder_driver_processes_every_geojson(nfx, tmp_path):
    """run.sh:43-55: <name>.geojson pairs with the slide of the same name, output <name>.<ext>; EXTRACTION_ARGS before the
    positionals, EXTRACTION_FEATURES after; a failing file is reported and the loop goes on."""
    synth = importlib.import_module("nuclei-feature-extraction_amd.synth")
    ps, H, W = 32, 256, 256
    (tmp_path / "g").mkdir(); (tmp_path / "s").mkdir()
    for k, name in enumerate(["a", "b"]):
        seed = synth.SEED_BASE + 40 + k
        cx, cy = synth.centres(seed, 0, 30, ps, W, H)
        off, xy = synth.polygons(seed, 0, 30, ps, cx, cy)
        write_png(tmp_path / "s" / f"{name}.png", synth.tile(seed, H, W, off, xy, ps))
        write_geojson(tmp_path / "g" / f"{name}.geojson", off, xy)
    (tmp_path / "g" / "c.geojson").write_text("{\"type\": \"FeatureCollection\", \"features\": []}")   # no slide c.*: fails
    env = dict(os.environ, EXTRACTION_ARGS="-p 32 -b 10 -g 0", EXTRACTION_FEATURES="geometry color")
    r = subprocess.run([RUNNER, str(tmp_path / "g"), str(tmp_path / "s"), str(tmp_path / "out"), "json"], capture_output=True, text=True, env=env)
    assert r.returncode == 1, r.stdout + r.stderr
    assert r.stdout.count("Done processing") == 2 and r.stdout.count("Error processing") == 1
    for name in ("a", "b"):
        lines = (tmp_path / "out" / f"{name}.json").read_text().strip().splitlines()
        assert len(lines) == 30 and set(json.loads(lines[0]).keys()) == set(nfx.all_column_names([0, 1]))
