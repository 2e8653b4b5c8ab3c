"""End-to-end timing of the C++ command line on a synthetic slide: GeoJSON parse, PNG decode, extraction (host buffers in,
so PCIe-inclusive), and each writer.  usage: python scripts/cli_throughput.py [nuclei] [tile side] [ps]"""
import importlib, json, os, subprocess, sys, tempfile, time, zlib, struct
import numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
synth = importlib.import_module("nuclei-feature-extraction_amd.synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
side = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
ps = int(sys.argv[3]) if len(sys.argv) > 3 else 64
CLI = os.path.join(ROOT, "nuclei-feature-extraction_amd", "bin", "nuclei-feature-extraction")

def write_png(path, img):  # [3,H,W] u8 -> 8-bit RGB, filter 0
    _, H, W = img.shape
    raw = np.concatenate([np.zeros((H, 1), np.uint8), np.ascontiguousarray(img.transpose(1, 2, 0)).reshape(H, W * 3)], axis=1).tobytes()
    def chunk(t, d): return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 1)) + chunk(b"IEND", b""))

with tempfile.TemporaryDirectory() as d:
    seed = synth.SEED_BASE + 21
    t0 = time.time()
    cx, cy = synth.centres(seed, 0, n, ps, side, side)
    off, xy = synth.polygons(seed, 0, n, ps, cx, cy)
    img = synth.tile(seed, side, side, off[:2001], xy, ps)
    write_png(os.path.join(d, "t.png"), img)
    with open(os.path.join(d, "n.geojson"), "w") as f:
        f.write('{"type":"FeatureCollection","features":[')
        for i in range(n):
            ring = xy[off[i]:off[i + 1]]
            x0, y0, x1, y1 = float(ring[:, 0].min()), float(ring[:, 1].min()), float(ring[:, 0].max()), float(ring[:, 1].max())
            f.write(("," if i else "") + json.dumps({"type": "Feature", "bbox": [x0, y0, x1, y1], "geometry": {"type": "Polygon", "coordinates": [ring.astype(float).tolist()]}}, separators=(",", ":")))
        f.write("]}")
    print(f"inputs: {n} nuclei, {side}x{side} tile, geojson {os.path.getsize(os.path.join(d, 'n.geojson')) / 1e6:.0f} MB, png {os.path.getsize(os.path.join(d, 't.png')) / 1e6:.0f} MB, generated in {time.time() - t0:.1f} s", flush=True)
    for ext, env in (("csv", {}), ("parquet", {}), ("feather", {}), ("json", {}), ("parquet", {"NFX_CLI_PIN": "0"}), ("feather", {"NFX_CLI_PIN": "0"}), ("feather", {})):
        out = os.path.join(d, "o." + ext)
        t0 = time.time()
        r = subprocess.run([CLI, "-v", "-o", "-g", "0", "-p", str(ps), os.path.join(d, "n.geojson"), os.path.join(d, "t.png"), out, "all"], capture_output=True, text=True,
                           env=dict(os.environ, **env))
        dt = time.time() - t0
        ext = ext + (" NFX_CLI_PIN=0" if env else "")
        dbg = [ln.split(">", 1)[1].strip() for ln in r.stderr.splitlines() if "DEBUG" in ln]
        print(f"{ext:22s} rc={r.returncode} wall {dt:6.2f} s  ({n / dt / 1e3:7.1f} k nuclei/s)  output {os.path.getsize(out) / 1e6:7.1f} MB | " + " | ".join(dbg), flush=True)
