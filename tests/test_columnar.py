"""Parquet and Arrow IPC writers of the C++ host (host/columnar.hpp; reference: polars writers, main.rs:169-180).

A small driver is compiled around the header, writes a deterministic table, and pyarrow reads both files back.
No GPU involved.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

# the host code under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only; any report aborts the driver: the test fails)
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"]

pa = pytest.importorskip("pyarrow")
import pyarrow.feather  # noqa: E402
import pyarrow.ipc  # noqa: E402
import pyarrow.parquet  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "nuclei-feature-extraction_amd", "host", "columnar.hpp")

DRIVER = r"""
#include "%s"
#include <cmath>
#include <cstdlib>
#include <limits>
using namespace nfxhost;
int main(int argc, char **argv)
{
    const int64_t n = atoll(argv[2]);
    const int F = atoi(argv[3]);
    std::vector<std::string> names{"centroid"}, keys((size_t)n);
    for (int j = 0; j < F; ++j) names.push_back("feature_" + std::to_string(j) + (j %% 2 ? "_0.5" : "_mean"));
    std::vector<float> rows((size_t)n * F);
    for (int64_t i = 0; i < n; ++i) {
        keys[(size_t)i] = std::to_string(i) + "," + std::to_string(i * 0.25);
        for (int j = 0; j < F; ++j) rows[(size_t)i * F + j] = (float)(i * 31 + j) * 0.125f - 7.0f;
    }
    if (n > 2 && F > 1) {
        rows[0] = std::numeric_limits<float>::quiet_NaN();
        rows[(size_t)F + 1] = std::numeric_limits<float>::infinity();
        rows[(size_t)2 * F] = -std::numeric_limits<float>::infinity();
    }
    FILE *f = fopen(argv[1], "wb");
    FeatureTable t{&names, &keys, rows.data(), n, F};
    // optional: every column j with j %% 3 == 1 is an f64 column (the reference's polygon columns), fed from a side matrix
    std::vector<int> wide((size_t)F, -1);
    int W = 0;
    for (int j = 0; j < F; ++j) if (j %% 3 == 1) wide[(size_t)j] = W++;
    std::vector<double> wide_rows((size_t)n * (W ? W : 1));
    for (int64_t i = 0; i < n; ++i)
        for (int j = 0; j < F; ++j)
            if (wide[(size_t)j] >= 0) wide_rows[(size_t)i * W + wide[(size_t)j]] = (double)(i * 31 + j) * 0.1 - 7.0;
    if (argc > 4 && W) { t.wide = wide.data(); t.wide_rows = wide_rows.data(); t.W = W; }
    const std::string path = argv[1];
    if (path.size() > 8 && path.substr(path.size() - 8) == ".parquet") write_parquet(f, t);
    else write_arrow_ipc(f, t);
    fclose(f);
    return 0;
}
"""


@pytest.fixture(scope="module")
def driver(tmp_path_factory):
    d = tmp_path_factory.mktemp("columnar")
    src = d / "driver.cpp"
    src.write_text(DRIVER % HEADER)
    exe = d / "driver"
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-Wall", "-Werror", *SAN, "-o", str(exe), str(src)], check=True)
    return exe


def expected(n, F):
    i = np.arange(n, dtype=np.int64)[:, None]
    j = np.arange(F, dtype=np.int64)[None, :]
    rows = ((i * 31 + j).astype(np.float32) * np.float32(0.125) - np.float32(7.0)).astype(np.float32)
    if n > 2 and F > 1:
        rows[0, 0] = np.nan
        rows[1, 1] = np.inf
        rows[2, 0] = -np.inf
    names = ["centroid"] + [f"feature_{k}" + ("_0.5" if k % 2 else "_mean") for k in range(F)]
    return names, rows


def check(table, n, F, wide=False):
    names, rows = expected(n, F)
    assert table.column_names == names
    assert table.num_rows == n
    assert pa.types.is_string(table.schema.field(0).type) or pa.types.is_large_string(table.schema.field(0).type)
    keys = table.column(0).to_pylist()
    if n:
        assert keys[0] == "0,0.000000" and keys[-1].startswith(f"{n - 1},")
    for k in range(F):
        got = table.column(k + 1).to_numpy()
        if wide and k % 3 == 1:   # an f64 column: values no f32 holds
            assert table.schema.field(k + 1).type == pa.float64() and got.dtype == np.float64
            np.testing.assert_array_equal(got, (np.arange(n, dtype=np.int64) * 31 + k).astype(np.float64) * 0.1 - 7.0)
            continue
        assert table.schema.field(k + 1).type == pa.float32()
        assert got.dtype == np.float32
        np.testing.assert_array_equal(got, rows[:, k])  # NaN == NaN here, bit-exact otherwise


# 131072 rows per row group / record batch: exactly one, one plus a ragged second, tiny, empty
SHAPES = [(5, 3), (0, 2), (1, 1), (131072, 2), (150001, 3), (1000, 418)]


@pytest.mark.parametrize("n,F", SHAPES)
def test_parquet_reads_back(driver, tmp_path, n, F):
    out = tmp_path / "t.parquet"
    subprocess.run([str(driver), str(out), str(n), str(F)], check=True)
    pf = pa.parquet.ParquetFile(str(out))
    assert pf.metadata.num_rows == n
    assert pf.metadata.num_row_groups == (n + 131071) // 131072
    check(pf.read(), n, F)
    if n:  # the string column carries the UTF8 annotation, floats are physical FLOAT
        assert pf.schema.column(0).physical_type == "BYTE_ARRAY"
        assert str(pf.schema.column(0).logical_type) == "String"
        assert pf.schema.column(1).physical_type == "FLOAT"


@pytest.mark.filterwarnings("ignore:pyarrow.feather:FutureWarning")
@pytest.mark.parametrize("n,F", SHAPES)
def test_arrow_ipc_reads_back(driver, tmp_path, n, F):
    out = tmp_path / "t.feather"
    subprocess.run([str(driver), str(out), str(n), str(F)], check=True)
    with pa.ipc.open_file(str(out)) as rd:
        assert rd.num_record_batches == (n + 131071) // 131072
        table = rd.read_all()
    check(table, n, F)
    table.validate(full=True)
    check(pa.feather.read_table(str(out)), n, F)  # the feather v2 entry point reads the same file


@pytest.mark.filterwarnings("ignore:pyarrow.feather:FutureWarning")
def test_pandas_sees_the_same_frame(driver, tmp_path):
    pd = pytest.importorskip("pandas")
    a, b = tmp_path / "t.parquet", tmp_path / "t.ipc"
    for p in (a, b):
        subprocess.run([str(driver), str(p), "300", "4"], check=True)
    fa, fb = pd.read_parquet(a), pd.read_feather(b)
    assert list(fa.columns) == list(fb.columns)
    assert fa["centroid"].tolist() == fb["centroid"].tolist()
    np.testing.assert_array_equal(fa.iloc[:, 1:].to_numpy(), fb.iloc[:, 1:].to_numpy())


@pytest.mark.filterwarnings("ignore:pyarrow.feather:FutureWarning")
@pytest.mark.parametrize("n,F", [(5, 3), (0, 5), (150001, 4), (1000, 418)])
def test_f64_columns_keep_their_width_in_both_formats(driver, tmp_path, n, F):
    """shape.rs:116-127: the reference's frame holds the polygon columns as f64; the writers type those columns DOUBLE."""
    a, b = tmp_path / "w.parquet", tmp_path / "w.feather"
    for p in (a, b):
        subprocess.run([str(driver), str(p), str(n), str(F), "wide"], check=True)
    pf = pa.parquet.ParquetFile(str(a))
    check(pf.read(), n, F, wide=True)
    if n:
        assert pf.schema.column(2).physical_type == "DOUBLE" and pf.schema.column(1).physical_type == "FLOAT"
    with pa.ipc.open_file(str(b)) as rd:
        table = rd.read_all()
    check(table, n, F, wide=True)
    table.validate(full=True)
