"""GPU parity of the trait methods frieda's three functions never call — PolyOps::extend, PolyOps::eval_at_point, FriOps::decompose
(SURVEY.md §8b's trait list behind `CpuBackend`, /root/reference/src/commit.rs:15-17, src/proof.rs:47-58) — through the C ABI against the
oracle's restatement of stwo's CpuBackend (parity unpinned: the reference holds no known answer for them), plus properties that do not
depend on the oracle: eval_at_point at a domain point equals the (pinned) circle FFT's evaluation there; the lambda of a decomposed g
is zero; a function inside the FFT space has lambda = 0."""
import ctypes as C

import numpy as np
import pytest

from util import DevBuf

pytestmark = pytest.mark.gpu
P = (1 << 31) - 1


def _check(ctx, rc):
    from frieda_amd.api import _check as chk

    chk(rc, ctx._h)


def rand_m31(rng, shape):
    return rng.integers(0, P, shape, dtype=np.uint32)


@pytest.mark.parametrize("ncols,log_coef,log_size", [(1, 0, 0), (1, 0, 5), (4, 3, 3), (4, 5, 9), (3, 10, 14), (4, 16, 20), (1, 12, 22)])
def test_circle_extend(gpu_ctx, oracle, ncols, log_coef, log_size):
    rng = np.random.default_rng(100 + log_size)
    coef = rand_m31(rng, (ncols, 1 << log_coef))
    d_in = DevBuf.from_array(gpu_ctx, coef)
    d_out = DevBuf.from_array(gpu_ctx, np.full((ncols, 1 << log_size), 0xDEADBEEF, dtype=np.uint32))
    _check(gpu_ctx, gpu_ctx._L.frieda_circle_extend(gpu_ctx._h, d_in.ptr, ncols, log_coef, log_size, d_out.ptr))
    got = d_out.to_array(np.uint32, (ncols, 1 << log_size))
    for c in range(ncols):
        assert np.array_equal(got[c], oracle.circle_extend(coef[c], log_size))


def test_circle_extend_refuses_a_smaller_size(gpu_ctx):
    d = DevBuf(gpu_ctx, 4096)
    assert gpu_ctx._L.frieda_circle_extend(gpu_ctx._h, d.ptr, 1, 5, 4, d.ptr) == 3  # stwo: assert!(log_size >= poly.log_size())
    assert gpu_ctx._L.frieda_circle_extend(gpu_ctx._h, None, 1, 3, 4, d.ptr) == 1
    assert gpu_ctx._L.frieda_circle_extend(gpu_ctx._h, d.ptr, 0, 3, 4, d.ptr) == 1


# every shape class of the two-level reduction: one thread, fewer lanes than a wave, exactly the lane bits, register levels, one full
# tile, two levels (12 + k), three levels (> 24)
@pytest.mark.parametrize("ncols,log_coef", [(1, 0), (4, 1), (4, 2), (3, 5), (4, 8), (4, 9), (2, 11), (4, 12), (4, 13), (1, 17), (4, 20), (1, 25)])
def test_circle_eval_at_point(gpu_ctx, oracle, ncols, log_coef):
    rng = np.random.default_rng(200 + log_coef)
    coef = rand_m31(rng, (ncols, 1 << log_coef))
    d = DevBuf.from_array(gpu_ctx, coef)
    px, py = rand_m31(rng, 4), rand_m31(rng, 4)  # (the method is a polynomial identity in x and y: the point need not lie on the circle)
    out = np.zeros((ncols, 4), dtype=np.uint32)
    _check(gpu_ctx, gpu_ctx._L.frieda_circle_eval_at_point(gpu_ctx._h, d.ptr, ncols, log_coef, px.ctypes.data, py.ctypes.data, out.ctypes.data))
    for c in range(ncols):
        assert np.array_equal(out[c], oracle.circle_eval_at_point(coef[c], px, py)), (c, log_coef)


@pytest.mark.parametrize("log_coef,log_domain", [(1, 3), (4, 8), (10, 14), (13, 13)])
def test_eval_at_point_agrees_with_the_transform_on_domain_points(gpu_ctx, oracle, log_coef, log_domain):
    """Oracle-independent: at a point of the evaluation domain (embedded in QM31) eval_at_point must return what PolyOps::evaluate — pinned
    by the reference's golden root — puts at that position."""
    rng = np.random.default_rng(300 + log_domain)
    coef = rand_m31(rng, (1, 1 << log_coef))
    d_c = DevBuf.from_array(gpu_ctx, coef)
    d_e = DevBuf(gpu_ctx, 4 << log_domain)
    _check(gpu_ctx, gpu_ctx._L.frieda_circle_evaluate(gpu_ctx._h, d_c.ptr, 1, log_coef, log_domain, d_e.ptr))
    ev = d_e.to_array(np.uint32, (1 << log_domain,))
    L = oracle.lib()
    for i in (0, 1, (1 << log_domain) // 2 + 3 if log_domain > 3 else 2, (1 << log_domain) - 1):
        x, y = C.c_uint32(), C.c_uint32()
        L.fo_circle_domain_at(log_domain, L.fo_bit_reverse_index(i, log_domain), C.byref(x), C.byref(y))
        px, py = np.array([x.value, 0, 0, 0], dtype=np.uint32), np.array([y.value, 0, 0, 0], dtype=np.uint32)
        out = np.zeros(4, dtype=np.uint32)
        _check(gpu_ctx, gpu_ctx._L.frieda_circle_eval_at_point(gpu_ctx._h, d_c.ptr, 1, log_coef, px.ctypes.data, py.ctypes.data, out.ctypes.data))
        assert out[0] == ev[i] and not out[1:].any()


def test_eval_at_point_rejects_bad_arguments(gpu_ctx):
    d = DevBuf(gpu_ctx, 4096)
    good = np.zeros(4, dtype=np.uint32)
    bad = np.array([P, 0, 0, 0], dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    f = gpu_ctx._L.frieda_circle_eval_at_point
    assert f(gpu_ctx._h, d.ptr, 1, 3, bad.ctypes.data, good.ctypes.data, out.ctypes.data) == 1
    assert f(gpu_ctx._h, d.ptr, 1, 3, good.ctypes.data, None, out.ctypes.data) == 1
    assert f(gpu_ctx._h, d.ptr, 1, 29, good.ctypes.data, good.ctypes.data, out.ctypes.data) == 1


@pytest.mark.parametrize("log_size", [0, 1, 2, 5, 8, 12, 13, 17, 22])
def test_fri_decompose(gpu_ctx, oracle, log_size):
    rng = np.random.default_rng(400 + log_size)
    ev = rand_m31(rng, (4, 1 << log_size))
    d_e = DevBuf.from_array(gpu_ctx, ev)
    d_g = DevBuf(gpu_ctx, 16 << log_size)
    lam = np.zeros(4, dtype=np.uint32)
    _check(gpu_ctx, gpu_ctx._L.frieda_fri_decompose(gpu_ctx._h, d_e.ptr, log_size, d_g.ptr, lam.ctypes.data))
    g = d_g.to_array(np.uint32, (4, 1 << log_size))
    og, olam = oracle.fri_decompose(ev)
    assert np.array_equal(lam, olam) and np.array_equal(g, og)
    # decomposing g again finds nothing left, in place
    _check(gpu_ctx, gpu_ctx._L.frieda_fri_decompose(gpu_ctx._h, d_g.ptr, log_size, d_g.ptr, lam.ctypes.data))
    assert not lam.any() and np.array_equal(d_g.to_array(np.uint32, (4, 1 << log_size)), g)


def test_decompose_of_a_lower_degree_function_is_the_identity(gpu_ctx):
    """Oracle-independent: lambda is the component along the (+ ... +, - ... -) pattern, which on the 2^n-point canonic domain is a
    multiple of the basis function of coefficient index 2^(n-1) (pi^(n-2)(x)); the circle-FFT basis is orthogonal to it, so a polynomial
    of 2^(n-1) coefficients evaluated on the 2^n domain has lambda = 0 and g = eval — and one with only that coefficient set is removed
    entirely."""
    rng = np.random.default_rng(9)
    n = 10
    coef = rand_m31(rng, (4, 1 << (n - 1)))
    d_c = DevBuf.from_array(gpu_ctx, coef)
    d_e = DevBuf(gpu_ctx, 16 << n)
    _check(gpu_ctx, gpu_ctx._L.frieda_circle_evaluate(gpu_ctx._h, d_c.ptr, 4, n - 1, n, d_e.ptr))
    ev = d_e.to_array(np.uint32, (4, 1 << n))
    d_g = DevBuf(gpu_ctx, 16 << n)
    lam = np.ones(4, dtype=np.uint32)
    _check(gpu_ctx, gpu_ctx._L.frieda_fri_decompose(gpu_ctx._h, d_e.ptr, n, d_g.ptr, lam.ctypes.data))
    assert not lam.any() and np.array_equal(d_g.to_array(np.uint32, (4, 1 << n)), ev)
    top = np.zeros((4, 1 << n), dtype=np.uint32)
    top[:, 1 << (n - 1)] = rand_m31(rng, 4)
    d_t = DevBuf.from_array(gpu_ctx, top)
    _check(gpu_ctx, gpu_ctx._L.frieda_circle_evaluate(gpu_ctx._h, d_t.ptr, 4, n, n, d_e.ptr))
    _check(gpu_ctx, gpu_ctx._L.frieda_fri_decompose(gpu_ctx._h, d_e.ptr, n, d_g.ptr, lam.ctypes.data))
    assert lam.all() and not d_g.to_array(np.uint32, (4, 1 << n)).any()
