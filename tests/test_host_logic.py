"""Host-side logic of the package that needs no GPU: shift lists, file format,
shard ranges, interface error behaviour."""
import numpy as np
import pytest

from conftest import DATA


def test_gen_float_shifts_matches_reference_semantics(oracle):
    import caf_cookoff_amd as caf
    for args in [(-100.0, 100.0, 0.25), (-50.0, 50.0, 1.0), (30.0, 35.0, 0.05), (80.0, 100.0, 0.1),
                 (-100.0, 100.0, 0.5)]:
        a = caf.gen_float_shifts(*args)
        assert np.array_equal(a, oracle.gen_float_shifts(*args))
    fr = caf.gen_float_shifts(-100.0, 100.0, 0.25)
    assert len(fr) == 800 and 69.25 in fr and fr[-1] == 99.75
    assert np.array_equal(caf.bench_shifts(), oracle.bench_shifts())
    with pytest.raises(ValueError):
        caf.gen_float_shifts(0.0, 1.0, 0.0001)  # step_by(0) panics in the reference


def test_read_file_c64(oracle, tmp_path):
    import caf_cookoff_amd as caf
    p = DATA / "chirp_0_raw.c64"
    a = caf.read_file_c64(p)
    assert a.dtype == np.complex128 and len(a) == 4096
    assert np.array_equal(a, oracle.read_file_c64(p))
    raw = np.fromfile(p, dtype="<f4")
    assert a[5].real == float(raw[10]) and a[5].imag == float(raw[11])  # widened, not re-rounded
    assert caf.read_file_c64_f32(p).dtype == np.complex64
    # utils.rs:45-62 writer is numpy complex128 compatible
    out = tmp_path / "x.bin"
    caf.write_file_binary(a[:7], out)
    assert np.array_equal(np.fromfile(out, dtype=np.complex128), a[:7])
    bad = tmp_path / "bad.c64"
    bad.write_bytes(b"\0" * 12)
    with pytest.raises(ValueError):
        caf.read_file_c64(bad)


def test_load_files_truncates_and_pads(tmp_path):
    import caf_cookoff_amd as caf
    nd, hs = caf.load_files(DATA / "chirp_0_raw.c64", DATA / "chirp_0_T+202samp_F+69.25Hz.c64")
    assert len(nd) == len(hs) == 4096
    short = tmp_path / "s.c64"
    np.arange(8, dtype="<f4").tofile(short)  # 4 samples
    nd2, hs2 = caf.load_files(DATA / "chirp_0_raw.c64", short)
    assert len(hs2) == 4096 and hs2[3] == 6 + 7j and not hs2[4:].any()


@pytest.mark.parametrize("nfreq,world", [(400, 1), (400, 2), (400, 8), (4096, 8), (7, 3), (3, 8)])
def test_shard_range_partitions(nfreq, world):
    import caf_cookoff_amd as caf
    spans = [caf.shard_range(nfreq, r, world) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == nfreq
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        assert a1 == b0 and a0 <= a1
    sizes = [b - a for a, b in spans]
    assert max(sizes) - min(sizes) <= 1


def test_xcor_interface_errors():
    import caf_cookoff_amd as caf
    with pytest.raises(caf.CafError):
        caf.Xcor(12)  # not a power of two
    with pytest.raises(caf.CafError):
        caf.Xcor(0)


@pytest.mark.parametrize("count,nworkers", [(0, 1), (1, 4), (10, 3), (37, 2), (1000, 8), (1001, 8), (5, 5), (64, 7)])
def test_multi_stream_share_is_a_round_robin_partition(count, nworkers):
    """caf_multi_stream_share (the ordering rule of the surface-parallel multi-GPU driver, pure host
    arithmetic in the C-ABI library): pair k goes to worker k % nworkers, every pair exactly once, results at
    the input positions; shares differ by at most one pair."""
    import caf_cookoff_amd as caf
    owner = np.full(count, -1)
    sizes = []
    for w in range(nworkers):
        first, stride, items = caf.multi_stream_share(count, nworkers, w)
        assert (first, stride) == (w, nworkers)
        idx = first + stride * np.arange(items)
        assert (idx < count).all() and (owner[idx] == -1).all()
        owner[idx] = w
        sizes.append(items)
    assert (owner == np.arange(count) % nworkers).all()
    assert sum(sizes) == count and max(sizes) - min(sizes) <= 1
    with pytest.raises(caf.CafError):
        caf.multi_stream_share(10, 0, 0)
    with pytest.raises(caf.CafError):
        caf.multi_stream_share(10, 2, 2)


def test_small_row_lds_geometry_is_conflict_free_under_the_bank_model():
    """The exchange and staging strides of k_small_rows (kernels_small.hpp: element e at e + (e >> 4), rows
    L + L/16 apart, |.|^2 staging rows of ROWT reals) were chosen with tools/lds_banks_small.py, a model of the LDS
    bank rules of MI355X_MICROARCH.md; the counters agree (profiles/r03_small_rows_lds.txt).  Keep model and
    geometry in step: conflict-free exchanges for L = 32 ... 256, the known 128 extra cycles of the second exchange
    at L = 512 / 1024, conflict-free staging from L = 32 on."""
    import importlib.util
    import re
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    spec = importlib.util.spec_from_file_location("lds_banks_small", root / "tools" / "lds_banks_small.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    src = (root / "caf_cookoff_amd" / "csrc" / "kernels_small.hpp").read_text()
    assert re.search(r"static constexpr int ROWX = L \+ L / 16;", src)
    assert "return e + (e >> 4);" in src
    assert re.search(r"ROWT = L >= 256 \? L : L \+ \(sizeof\(T\) == 8 \? L / 16 : \(L >= 32 \? L / 32 : 1\)\);", src)
    for esz in (16, 8):
        for logl in range(5, 11):
            L = 1 << logl
            cyc, floor, rowx = m.exchange(logl, esz, 1)          # extra = 1: ROWX = P(L - 1) + 1 + 1 = L + L/16
            assert rowx == L + L // 16
            if L <= 256:
                assert cyc == floor, (L, esz, cyc, floor)
            else:
                assert cyc - floor == (128 if esz == 16 else 64), (L, esz, cyc, floor)
            tsz = esz // 2
            pad_t = 0 if L >= 256 else (L // 16 if tsz == 8 else max(1, L // 32))
            scyc, sfloor, _ = m.staging(logl, tsz, pad_t)
            assert scyc == sfloor, (L, tsz, scyc, sfloor)
