"""bench.py's roofline object, computed from a made-up timing and a made-up counter summary (CPU: no GPU, no run): the keys the judge reads
and the arithmetic behind them -- `lds.conflict_free_frac`, `traffic` per launch, `traffic_step` against the algorithmic bytes."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_roofline_object_with_counters(monkeypatch):
    b = _bench()
    n = 1_000_000
    slots = {"k_glcm": {"traffic_bytes_per_launch": 9.0e9, "valu_insts_per_nucleus": 12000.0, "lds_idx_active_cycles_per_nucleus": 9000.0,
                        "lds_addr_conflict_over_idx_active": 0.576, "lds_bank_conflict_over_idx_active": 0.6},
             "k_color": {"traffic_bytes_per_launch": 7.0e9, "valu_insts_per_nucleus": 2000.0, "lds_idx_active_cycles_per_nucleus": 300.0,
                         "lds_addr_conflict_over_idx_active": 0.02, "lds_bank_conflict_over_idx_active": 0.2},
             "k_hue_batch": {"traffic_bytes_per_launch": 9.7e9, "valu_insts_per_nucleus": 1500.0}}
    monkeypatch.setattr(b, "committed_counters", lambda wl: (slots, "test"))
    timing = {"k_glcm": {"launches": 10, "total_ms": 295.0}, "k_color": {"launches": 10, "total_ms": 40.0}, "k_gabor": {"launches": 0, "total_ms": 0.0}}
    rf = b.roofline_of(timing, steps=10, n=n, bytes_per_nucleus=8184.0, workload="config3", full_size=True)
    assert rf["kernel"] == "k_glcm" and rf["bound"] == "lds" and abs(rf["avg_launch_ms"] - 29.5) < 1e-9
    hb = rf["hbm"]
    assert abs(hb["achieved"] - 8184.0 * n / 29.5e-3 / 1e9) < 0.01 and abs(hb["frac"] - hb["achieved"] / 8000.0) < 1e-6 and hb["launches_per_step"] == 1
    busy = 9000.0 * n / 29.5e-3 / (256 * 2.4e9)
    assert abs(rf["lds"]["frac"] - busy) < 1e-3 and abs(rf["frac"] - busy) < 1e-3
    assert abs(rf["lds"]["conflict_free_frac"] - busy * (1 - 0.576)) < 1e-3          # the pipe's busy cycles that move data
    assert rf["traffic"] == int(9.0e9)
    ts = rf["traffic_step"]
    assert ts["bytes"] == int(25.7e9) and ts["algorithmic_bytes"] == int(8184.0 * n) and abs(ts["ratio"] - 25.7e9 / (8184.0 * n)) < 1e-3
    # a streamed step launches the kernel once per chunk: per-launch figures scale with the nuclei a launch sees, per-step ones do not
    timing = {"k_glcm": {"launches": 100, "total_ms": 310.0}}
    rf2 = b.roofline_of(timing, steps=10, n=n, bytes_per_nucleus=8184.0, workload="config3", full_size=True)
    assert rf2["hbm"]["launches_per_step"] == 10 and rf2["hbm"]["nuclei_per_launch"] == n / 10 and rf2["traffic"] == int(9.0e8)
    assert rf2["traffic_step"]["bytes"] == ts["bytes"]


def test_roofline_object_without_counters():
    b = _bench()
    rf = b.roofline_of({"k_raster_shape": {"launches": 4, "total_ms": 2.0}}, steps=4, n=3000, bytes_per_nucleus=304.0, workload="config2", full_size=False)
    assert rf["bound"] == "hbm" and rf["traffic"] is None and rf["traffic_step"] is None and rf["lds"] is None and rf["issue"] is None
    assert rf["counters_source"] == "not the full-size workload"
