"""The generic radix-16 passes over HBM, the stand-alone operators (apply_freq_shift, xcor, find_peak), the Go / Python views,
the coarse-to-fine search and the CLI demo, against the golden vectors, the oracle and the reference's own Python output.
Every call goes through the C ABI (libcaf_hip.so); the oracle is the checker."""
import numpy as np
import pytest

from conftest import DATA
from gpu_common import FS, ROOT, TOL32, TOL64, _mmap_array, _pair, _planted

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("pinned_copies")]


# ------------------------------------------------------------- generic path --
@pytest.mark.parametrize("n", [1, 2, 8, 64, 512, 1024, 2048, 8192, 16384])
def test_generic_sizes_vs_oracle(n, eng, oracle):
    rng = np.random.default_rng(n)
    a = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    lag = n // 3
    b = np.roll(a, lag) * np.exp(2j * np.pi * 37.5 * np.arange(n) / FS) + 0.01 * rng.standard_normal(n)
    fr = np.array([-75.0, 0.0, 37.5, 75.0, 112.5])
    surf, ridx, rval, peak = eng.surface_arrays(a, b, fr, FS)
    osurf, oidx, oval = oracle.np_caf_surface(a, b, fr, FS)
    assert surf.shape == (5, 2 * n)
    assert np.max(np.abs(surf - osurf)) <= TOL64 * osurf.max()
    assert np.array_equal(ridx, oidx)
    assert (peak.freq, peak.idx) == oracle.np_find_peak(fr, oidx, oval)


@pytest.mark.parametrize("n", [8, 64, 4096])
def test_apply_freq_shift_golden(n, eng, golden):
    """mod.rs:46-65 vectors (recurrence) vs direct-phasor kernel: <= 1e-12 absolute
    on O(1) data (the recurrence itself drifts ~2e-14 from the exact phasor)."""
    a = golden[f"vec{n}_a"]
    for tag, f in (("77p77", 77.77), ("m12p5", -12.5)):
        out = eng.apply_freq_shift(a, f, FS)
        g = golden[f"vec{n}_shift_{tag}"]
        assert out[0] == a[0]  # sample 0 is multiplied by 1+0j (mod.rs:57-59)
        assert np.max(np.abs(out - g)) <= 1e-12 * max(1.0, np.max(np.abs(g)))


@pytest.mark.parametrize("n", [8, 64, 4096])
def test_xcor_golden(n, eng, golden):
    import caf_cookoff_amd as caf
    a, b = golden[f"vec{n}_a"], golden[f"vec{n}_b"]
    x = caf.Xcor(n, eng)
    out = x.clone().run(a, b)
    g = golden[f"vec{n}_xcor"]
    assert np.max(np.abs(out - g)) <= 1e-12 * np.max(np.abs(g))


def test_xcor_size_independent_properties(eng):
    """Full-size properties: circular-shift covariance and conjugate symmetry
    xcor(a,b)[k] == conj(xcor(b,a)[-k]); linearity in a."""
    rng = np.random.default_rng(5)
    n = 8192
    a = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    b = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    c = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    xab = eng.xcor(a, b)
    scale = np.max(np.abs(xab))
    assert np.max(np.abs(eng.xcor(np.roll(a, -17), b) - np.roll(xab, -17))) <= 1e-12 * scale
    xba = eng.xcor(b, a)
    assert np.max(np.abs(xab - np.conj(np.roll(xba[::-1], 1)))) <= 1e-12 * scale
    assert np.max(np.abs(eng.xcor(a + 2.5 * c, b) - (xab + 2.5 * eng.xcor(c, b)))) <= 1e-11 * scale


@pytest.mark.parametrize("n", [8, 256, 2048])
def test_c64_generic_sizes(n, eng, oracle):
    rng = np.random.default_rng(100 + n)
    a = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(np.complex64)
    b = (np.roll(a, n // 4) * np.exp(2j * np.pi * 50.0 * np.arange(n) / FS)).astype(np.complex64)
    fr = np.array([0.0, 50.0, 100.0])
    surf, ridx, rval, peak = eng.surface_arrays(a, b, fr, FS, dtype="c64")
    osurf, oidx, oval = oracle.np_caf_surface(a.astype(np.complex128), b.astype(np.complex128), fr, FS)
    assert np.max(np.abs(surf - osurf)) <= TOL32 * osurf.max()
    assert int(ridx[1]) == int(oidx[1]) and peak.row == 1


def test_generic_path_still_covers_other_big_sizes(eng, oracle):
    """n = 131072 complex128 (L = 262144) exceeds every LDS-resident form (sixteen chains of 8192 points end at
    n = 65536): generic HBM-pass path."""
    from caf_cookoff_amd.synth import make_pair
    n = 131072
    s0, s1, lag, fo = make_pair(n=n, seed=79, lag=33, foffset=5.0)
    fr = np.array([4.5, 5.0, 5.5])
    plan = eng.plan(n, fr, FS)
    assert plan.path == "generic"
    plan.close()
    surf, ridx, rval, peak = eng.surface_arrays(s0, s1, fr, FS)
    osurf, oidx, oval = oracle.np_caf_surface(s0, s1, fr, FS)
    assert np.max(np.abs(surf - osurf)) <= TOL64 * osurf.max() and (peak.freq, peak.idx) == (5.0, lag)


# ------------------------------------------------- "next" rows of SURVEY.md 8(f) --
def test_go_and_python_views(eng, oracle):
    """The other cook-off implementations' conventions as views of the same surface,
    checked against direct restatements of caf_go/caf.go:95-116 and caf_python/caf.py:15-18."""
    from scipy import signal
    nd, hs = _pair(oracle, 4)
    n = len(nd)
    fr = np.array([82.5, 83.0, 83.5])
    surf, ridx, rval, peak = eng.surface_arrays(nd, hs, fr, FS)
    go = eng.surface_view(surf, "go")
    py = eng.surface_view(surf, "python")
    assert go.shape == (3, 2 * n) and py.shape == (3, n)
    z = np.zeros(n, dtype=np.complex128)
    for r, f in enumerate(fr):
        shifted = nd * np.exp(2j * np.pi * f * np.arange(n) / FS)          # apply_fdoa (caf.go:118-126)
        corr = np.fft.ifft(np.fft.fft(np.concatenate([shifted, z])) *
                           np.conj(np.fft.fft(np.concatenate([z, hs]))))   # xcor (caf.go:95-116)
        assert np.max(np.abs(go[r] - np.abs(corr))) <= 1e-9 * np.abs(corr).max()
        same = np.abs(signal.correlate(shifted, hs, mode="same", method="fft"))  # caf.py:15-18
        assert np.max(np.abs(py[r] - same)) <= 1e-9 * same.max()
    # main.go:35 and caf.py:145 recover the same (tau, f) as find_peak
    fdx, tdx = np.unravel_index(np.argmax(go), go.shape)
    assert (n - tdx, fr[fdx]) == (peak.idx, peak.freq) == (70, 83.0)
    fdx, tmax = np.unravel_index(np.argmax(py), py.shape)
    assert (n // 2 - tmax, fr[fdx]) == (70, 83.0)


def test_python_view_vs_reference_amb_surf_fixture(eng, oracle):
    """CAF_VIEW_PYTHON against the output of the reference's OWN caf_python/caf.py amb_surf
    (caf.py:89-117) on its __main__ pair (caf.py:126-133), stored by tests/golden/make_py_fixture.py.
    The reference computes in complex64 (np.empty_like(ray), caf.py:30; scipy correlate keeps
    single precision): tolerance 2e-6 of the surface maximum (measured 2.1e-7)."""
    from conftest import GOLDEN
    g = np.load(GOLDEN / "py_amb_surf.npz")
    nd, hs = oracle.load_pair(DATA, str(g["needle"]), str(g["haystack"]))
    fr = g["freqs"]
    assert np.array_equal(fr, oracle.bench_shifts())  # np.arange(-100, 100, .5) == the Rust bench grid
    surf, ridx, rval, peak = eng.surface_arrays(nd, hs, fr, FS)
    py = eng.surface_view(surf, "python")
    assert py.shape == tuple(g["shape"])
    tol = 2e-6 * g["row_max"].max()
    assert np.max(np.abs(py[g["full_rows"]] - g["rows"])) <= tol
    assert np.max(np.abs(py.reshape(-1)[::int(g["stride"])] - g["strided"])) <= tol
    assert np.max(np.abs(py.max(axis=1) - g["row_max"])) <= tol
    fmax, tmax = np.unravel_index(py.argmax(), py.shape)                  # caf.py:144-146
    assert (len(nd) // 2 - tmax, fr[fmax]) == (int(g["tau"]), float(g["freq"])) == (70, 83.0)
    assert (peak.idx, peak.freq) == (70, 83.0)


def test_refine_peak_coarse_to_fine(eng, oracle):
    """Coarse 1 Hz grid then the fine grids of the reference's KATs (test.rs:174,212)."""
    for k, coarse, fine, want in ((2, (25.0, 40.0, 1.0), 0.05, (32.15, 169)), (4, (70.0, 100.0, 1.0), 0.1, (82.9, 70))):
        nd, hs = _pair(oracle, k)
        cf = oracle.gen_float_shifts(*coarse)
        (cfq, cidx), (ffq, fidx), ff = eng.refine_peak(nd, hs, FS, cf, fine)
        assert cidx == want[1] and abs(cfq - want[0]) <= 0.5
        assert fidx == want[1] and abs(ffq - want[0]) < 1e-9


def test_cli_demo(tmp_path):
    import subprocess, sys
    from conftest import ROOT
    dump = tmp_path / "surf.bin"
    r = subprocess.run([sys.executable, "-m", "caf_cookoff_amd", str(DATA / "chirp_0_raw.c64"),
                        str(DATA / "chirp_0_T+202samp_F+69.25Hz.c64"), "--dump-surf", str(dump), "--view", "go"],
                       capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert r.returncode == 0, r.stderr
    assert r.stdout.splitlines() == ["Frequency offset: 69.0Hz", "Time offset: 202 samples (4.208ms)"]  # main.rs:29-31
    surf = np.fromfile(dump, dtype="<f8").reshape(400, 8192)
    fdx, tdx = np.unravel_index(np.argmax(surf), surf.shape)
    assert 4096 - tdx == 202  # main.go:35


def test_apply_freq_shift_c64_and_find_peak_direct(eng, golden):
    """The c64 twin of mod.rs:46-65 (phase in f64, one rounding) and find_peak on caller rows."""
    from caf_cookoff_amd import CafSurfaceRow
    a = golden["vec4096_a"].astype(np.complex64)
    out = eng.apply_freq_shift(a, 77.77, FS)
    assert out.dtype == np.complex64 and out[0] == a[0]
    ref = golden["vec4096_shift_77p77"]
    assert np.max(np.abs(out.astype(np.complex128) - ref)) <= 2e-7 * np.max(np.abs(ref)) + 1e-9
    rows = [CafSurfaceRow(1.0, None, 10, 3.0), CafSurfaceRow(2.0, None, 20, 5.0), CafSurfaceRow(3.0, None, 30, 5.0),
            CafSurfaceRow(4.0, None, 40, 0.0)]
    assert eng.find_peak(rows) == (2.0, 20)           # first strictly-greater row wins (mod.rs:36)
    assert eng.find_peak(rows[3:]) == (0.0, 0)        # nothing above the initial 0.0 (mod.rs:32-35)


@pytest.mark.parametrize("dtype", ["c128", "c64"])
@pytest.mark.parametrize("n", [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768])
def test_xcor_every_size_vs_oracle(n, dtype, eng, oracle):
    """Xcor::run (xcor_rustfft.rs:51-78) for every power of two: one launch up to n = 16384 (kernels_xcor.hpp; complex128:
    8192), radix-2 passes beyond; against the numpy restatement, plus the defining property out[k] = sum a[m+k] conj(b[m])
    on a shifted copy (peak at the shift) and Xcor's length assertions."""
    import caf_cookoff_amd as caf
    rng = np.random.default_rng(7000 + n)
    cdt = np.complex128 if dtype == "c128" else np.complex64
    a = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(cdt)
    b = (rng.standard_normal(n) + 1j * rng.standard_normal(n)).astype(cdt)
    got = caf.Xcor(n, eng).run(a, b)
    want = oracle.np_xcor(a.astype(np.complex128), b.astype(np.complex128))
    assert got.dtype == cdt and got.shape == (n,)
    tol = (1e-10 if dtype == "c128" else 2e-4) * max(1.0, np.max(np.abs(want)))
    assert np.max(np.abs(got - want)) <= tol
    if n >= 4:
        sh = n // 3
        got2 = eng.xcor(np.roll(b, sh), b)      # a[m] = b[m - sh]  ->  peak at k = sh
        assert int(np.argmax(np.abs(got2))) == sh
    # a second call with other data through the same cached tables
    got3 = eng.xcor(b, a)
    assert np.max(np.abs(got3 - oracle.np_xcor(b.astype(np.complex128), a.astype(np.complex128)))) <= tol


def test_views_in_place_on_pinned_memory(eng):
    """caf_surface_view reads / writes memory of caf_host_alloc in place (no staging copy) and returns the same bits as
    through ordinary memory; a source that straddles the edge of a registered range still works (copy cut at the edge)."""
    rng = np.random.default_rng(3)
    surf = rng.random((12, 256))
    for view in ("go", "python"):
        ref = eng.surface_view(surf, view)
        psrc = eng.host_empty(surf.shape, np.float64)
        psrc[:] = surf
        assert np.array_equal(eng.surface_view(psrc, view), ref)
        big = _mmap_array((16, 256), np.float64)        # rows of 2 KiB: eight pages
        big[1:13] = surf
        eng.host_register(big[:6])                      # the first three pages only: big[1:13] straddles the edge
        assert np.array_equal(eng.surface_view(big[1:13], view), ref)
        eng.host_unregister(big[:6])
        del psrc


@pytest.mark.parametrize("dtype", ["c128", "c64"])
@pytest.mark.parametrize("n", [1, 2, 4, 8, 16, 64, 128, 1024, 2048, 8192])
def test_generic_radix16_passes_vs_oracle(n, dtype, oracle, monkeypatch):
    """The path of every shape no LDS-resident kernel covers (n > 131072 / 65536 in the product): mixed-radix Stockham
    passes over HBM -- radix 16 while at least 16 points remain, then 8 / 4 / 2 (k_fft_pass).  Forced here for small
    and medium sizes through the measurement library (CAF_SMALL=0, CAF_CHAIN=0) so that every pass combination
    (L = 2 ... 16384: remainders 2, 4, 8 and none) is checked against the oracle."""
    import caf_cookoff_amd as caf
    monkeypatch.setenv("CAF_SMALL", "0")
    monkeypatch.setenv("CAF_CHAIN", "0")
    meng = caf.Engine(0, lib=caf.MEASURE_LIB_PATH)
    rng = np.random.default_rng(900 + n)
    cdt = np.complex128 if dtype == "c128" else np.complex64
    tol = TOL64 if dtype == "c128" else TOL32
    fr = np.array([-50.0, 0.0, 12.5, 50.0, 333.0])
    lag = 0 if n < 4 else n // 4
    x, y = _planted(rng, n, FS, 12.5, lag, cdt)
    plan = meng.plan(n, fr, FS, dtype=dtype)
    assert plan.path == "generic" and "k_fft_pass" in plan.kernel_name
    plan.close()
    surf, ridx, rval, pk = meng.surface_arrays(x, y, fr, FS, dtype=dtype)
    osurf, oidx, oval = oracle.np_caf_surface(x.astype(np.complex128), y.astype(np.complex128), fr, FS)
    assert np.max(np.abs(surf - osurf)) <= tol * osurf.max()
    if n >= 16:
        assert (pk.freq, int(pk.idx)) == oracle.np_find_peak(fr, oidx, oval) == (12.5, lag)
    meng.close()
