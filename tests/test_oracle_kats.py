"""Pins the CPU oracle (both restatements) against every known answer the
reference's own tests hold for this path: caf_rust/tests/test.rs:14-316."""
import numpy as np
import pytest

from conftest import DATA, GOLDEN

FS = 48000


def _kat_ids(oracle_mod=None):
    from oracle import caf_oracle as O
    return O.KATS


@pytest.mark.parametrize("kat", _kat_ids(), ids=lambda k: f"chirp{k[0]}")
def test_kat_c_oracle(kat, oracle, coracle):
    k, hf, (s, e, st), exp = kat
    nd, hs = oracle.load_pair(DATA, f"chirp_{k}_raw.c64", hf)
    assert len(nd) == 4096 and len(hs) == 4096
    fr = coracle.gen_float_shifts(s, e, st)
    assert np.array_equal(fr, oracle.gen_float_shifts(s, e, st))
    _, ridx, rval = coracle.caf_surface(nd, hs, fr, FS, want_surface=False)
    freq, idx = coracle.find_peak(fr, ridx, rval)
    assert freq == exp[0]  # exact f64 equality, like assert_eq! in test.rs
    assert idx == exp[1]


@pytest.mark.parametrize("kat", _kat_ids(), ids=lambda k: f"chirp{k[0]}")
def test_kat_golden_rows_match_c_oracle(kat, oracle, coracle, golden, manifest):
    """numpy-generated per-row goldens agree with the C restatement (own FFT)."""
    k, hf, (s, e, st), exp = kat
    nd, hs = oracle.load_pair(DATA, f"chirp_{k}_raw.c64", hf)
    fr = oracle.gen_float_shifts(s, e, st)
    assert manifest["kats"][str(k)]["nfreq"] == len(fr)
    _, ridx, rval = coracle.caf_surface(nd, hs, fr, FS, want_surface=False, hoist=True, nthreads=4)
    assert np.array_equal(ridx, golden[f"kat{k}_row_idx"])
    g = golden[f"kat{k}_row_val"]
    assert np.max(np.abs(rval - g)) <= 1e-12 * g.max()


def test_kat0_numpy_restatement(oracle):
    """The numpy restatement (generator of tests/golden) on the chirp_0 KAT."""
    k, hf, (s, e, st), exp = oracle.KATS[0]
    nd, hs = oracle.load_pair(DATA, f"chirp_{k}_raw.c64", hf)
    fr = oracle.gen_float_shifts(s, e, st)
    _, ridx, rval = oracle.np_caf_surface(nd, hs, fr, FS, want_surface=False)
    assert oracle.np_find_peak(fr, ridx, rval) == exp


def test_three_fft_and_hoisted_agree(oracle, coracle):
    """Reference recomputes FFT(haystack) per row (xcor_rustfft.rs:58-59);
    hoisting it must not change a single bit of the result."""
    nd, hs = oracle.load_pair(DATA, "chirp_9_raw.c64", oracle.KATS[9][1])
    fr = oracle.bench_shifts()[::8]
    s1, i1, v1 = coracle.caf_surface(nd, hs, fr, FS, hoist=False)
    s2, i2, v2 = coracle.caf_surface(nd, hs, fr, FS, hoist=True, nthreads=3)
    assert np.array_equal(s1, s2) and np.array_equal(i1, i2) and np.array_equal(v1, v2)


def test_bench_config_goldens(oracle, coracle, golden, manifest):
    """caf_bench.rs:26-40 shape: chirp_0 and chirp_4, 400 shifts."""
    fr = oracle.bench_shifts()
    assert len(fr) == 400 and fr[0] == -100.0 and fr[-1] == 99.5
    assert np.array_equal(fr, golden["bench_freqs"])
    for k in ("0", "4"):
        m = manifest["bench"][k]
        nd, hs = oracle.load_pair(DATA, m["needle"], m["haystack"])
        surf, ridx, rval = coracle.caf_surface(nd, hs, fr, FS, hoist=True, nthreads=4)
        assert coracle.find_peak(fr, ridx, rval) == (m["best_freq"], m["best_idx"])
        assert np.array_equal(ridx, golden[f"bench{k}_row_idx"])
        tol = 1e-12 * m["surface_max"]
        assert np.max(np.abs(rval - golden[f"bench{k}_row_val"])) <= tol
        assert np.max(np.abs(surf[manifest["full_rows"]] - golden[f"bench{k}_rows"])) <= tol
        assert np.max(np.abs(surf.reshape(-1)[::manifest["stride"]] - golden[f"bench{k}_strided"])) <= tol


@pytest.mark.parametrize("n", [8, 64, 4096])
def test_shift_and_xcor_vectors(n, coracle, golden):
    a, b = golden[f"vec{n}_a"], golden[f"vec{n}_b"]
    # recurrence is restated operation-for-operation: bit-exact between C and numpy
    assert np.array_equal(coracle.apply_freq_shift(a, 77.77, FS), golden[f"vec{n}_shift_77p77"])
    assert np.array_equal(coracle.apply_freq_shift(a, -12.5, FS), golden[f"vec{n}_shift_m12p5"])
    x = coracle.xcor(a, b)
    g = golden[f"vec{n}_xcor"]
    assert np.max(np.abs(x - g)) <= 1e-13 * np.max(np.abs(g))


def test_xcor_definition_small(coracle):
    """out[k] = sum_m a[(m+k) mod n] * conj(b[m])  (xcor_rustfft.rs:51-78)."""
    rng = np.random.default_rng(1)
    n = 16
    a = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    b = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    ref = np.array([sum(a[(m + k) % n] * np.conj(b[m]) for m in range(n)) for k in range(n)])
    assert np.allclose(coracle.xcor(a, b), ref, rtol=0, atol=1e-12)


def test_fft_against_numpy(coracle):
    rng = np.random.default_rng(2)
    for n in (1, 2, 4, 8, 32, 128, 8192):
        x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        assert np.allclose(coracle.fft(x), np.fft.fft(x), rtol=0, atol=1e-11 * max(1, n) ** 0.5)
        assert np.allclose(coracle.fft(x, inverse=True), np.fft.ifft(x) * n, rtol=0, atol=1e-11 * max(1, n) ** 0.5)


def test_edge_cases(oracle, coracle):
    # mod.rs:143 / 32-35: all-zero input -> every row (idx 0, val 0.0), peak (0.0, 0)
    z = np.zeros(8, dtype=np.complex128)
    fr = np.array([5.0, 6.0])
    surf, ridx, rval = coracle.caf_surface(z, z, fr, FS)
    assert not surf.any() and not ridx.any() and not rval.any()
    assert coracle.find_peak(fr, ridx, rval) == (0.0, 0)
    # empty frequency list -> empty surface, peak (0.0, 0)
    a = np.ones(4, dtype=np.complex128)
    surf, ridx, rval = coracle.caf_surface(a, a, np.array([]), FS)
    assert surf.shape == (0, 8) and coracle.find_peak(np.array([]), ridx, rval) == (0.0, 0)
    # n = 1 (L = 2)
    surf, ridx, rval = coracle.caf_surface(np.array([2 + 0j]), np.array([3 + 0j]), np.array([0.0]), FS)
    assert surf.shape == (1, 2) and ridx[0] == 0 and rval[0] == 36.0
    # gen_float_shifts truncation semantics (test.rs:341-343)
    assert len(oracle.gen_float_shifts(30.0, 35.0, 0.05)) == 100
    assert oracle.gen_float_shifts(80.0, 100.0, 0.1)[29] == 82.9


def test_python_reference_fixture_pins_the_python_view(oracle):
    """tests/golden/py_amb_surf.npz holds outputs of the reference's OWN caf_python/caf.py amb_surf
    (caf.py:89-117; generated by tests/golden/make_py_fixture.py in the build container).  The
    restatement's |.|^2 surface, viewed in that module's convention (magnitude, n lags of scipy
    'same', reversed lag axis: out[i] = sqrt(surf[(n/2 - i) mod 2n]), caf.py:15-18,145), must
    reproduce it: values within the reference's complex64 arithmetic, (tau, f) exactly."""
    g = np.load(GOLDEN / "py_amb_surf.npz")
    nd, hs = oracle.load_pair(DATA, str(g["needle"]), str(g["haystack"]))
    fr = g["freqs"]
    surf, _, _ = oracle.np_caf_surface(nd, hs, fr, 48000)
    n = len(nd)
    py = np.sqrt(surf[:, (n // 2 - np.arange(n)) % (2 * n)])
    tol = 2e-6 * g["row_max"].max()
    assert np.max(np.abs(py[g["full_rows"]] - g["rows"])) <= tol
    assert np.max(np.abs(py.reshape(-1)[::int(g["stride"])] - g["strided"])) <= tol
    fmax, tmax = np.unravel_index(py.argmax(), py.shape)
    assert (n // 2 - tmax, fr[fmax]) == (int(g["tau"]), float(g["freq"])) == (70, 83.0)


def test_oracles_nan_semantics(oracle, coracle):
    """mod.rs:143-151: a NaN magnitude never satisfies `mag > max`: rows of NaN report (0, 0.0),
    and a finite value after a NaN still wins (both restatements)."""
    rng = np.random.default_rng(3)
    n = 64
    a = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    b = np.roll(a, 5)
    fr = np.array([-3.0, 0.0])
    x = a.copy()
    x[7] = complex(np.nan, 0.0)
    for impl in ("np", "c"):
        if impl == "np":
            s, i, v = oracle.np_caf_surface(x, b, fr, 48000)
        else:
            s, i, v = coracle.caf_surface(x, b, fr, 48000, hoist=False, nthreads=1)
        assert np.isnan(s).all() and not i.any() and not v.any()
    assert oracle.np_find_peak([1.0, 2.0, 3.0], [4, 5, 6], [float("nan"), 2.0, float("nan")]) == (2.0, 5)
    assert coracle.find_peak(np.array([1.0, 2.0, 3.0]), np.array([4, 5, 6], dtype=np.uint64),
                             np.array([np.nan, 2.0, np.nan])) == (2.0, 5)
