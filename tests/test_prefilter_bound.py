"""The error bound behind the two-stage exact search (knn_kernels.h, "bf16 mirror as prefilter"), checked on the CPU:
|coarse - exact| <= eps = 2^-8 + 4.1 (dim + 8) 2^-24 + 2e-6 for the cosine distance computed from bf16-rounded rows,
on random rows and on rows built to sit at the worst case of the rounding (every element half an ulp off, signs aligned
with the query)."""
import numpy as np

DIM = 768
EPS = 2.0 ** -8 + 4.1 * (DIM + 8) * 2.0 ** -24 + 2e-6   # bf16 keeps 8 significant bits: unit roundoff 2^-8


def bf16_rne(x: np.ndarray) -> np.ndarray:
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def cos_dist32(q, x, xx=None):
    """fp32 arithmetic throughout (summation order differs from the kernels': that is inside the bound's gamma terms)"""
    q = q.astype(np.float32); x = x.astype(np.float32)
    dot = np.float32(0)
    for c in range(0, DIM, 64):  # chunked like the kernels, any order
        dot = np.float32(dot + np.dot(q[c:c + 64], x[c:c + 64]).astype(np.float32))
    qq = np.float32(np.dot(q, q)); xn = np.float32(np.dot(x, x)) if xx is None else np.float32(xx)
    return np.float32(1) - dot / (np.sqrt(qq) * np.sqrt(xn))


def test_bound_holds_on_random_and_on_worst_case_rows():
    rng = np.random.default_rng(0)
    worst = 0.0
    for trial in range(300):
        q = rng.standard_normal(DIM).astype(np.float32)
        if trial % 3 == 0:
            x = rng.standard_normal(DIM).astype(np.float32) * np.float32(10.0 ** rng.integers(-6, 7))
        elif trial % 3 == 1:
            x = (q + 0.05 * rng.standard_normal(DIM)).astype(np.float32)  # a near neighbour: where the ranking is decided
        else:
            # worst case of round-to-nearest: mantissa bits below bf16 = 0x7FFF / 0x8001 (just under / over half an ulp),
            # the direction of each element's error chosen to push the dot product one way
            base = (np.abs(q) * (1.0 + rng.random(DIM))).astype(np.float32)
            u = base.view(np.uint32) & np.uint32(0xFFFF0000)
            x = ((u | np.uint32(0x7FFF)).view(np.float32) * np.sign(q)).astype(np.float32)
        exact = float(cos_dist32(q, x))
        coarse = float(cos_dist32(q, bf16_rne(x), xx=np.dot(x.astype(np.float64), x.astype(np.float64))))
        err = abs(coarse - exact)
        worst = max(worst, err)
        assert err <= EPS, (trial, err, EPS)
    assert worst > 0.5 * 2.0 ** -8  # the constructed rows do come close to the bound: it is not vacuous
    assert worst > 2.0 ** -9        # ... and a bound of 2^-9 (bf16 mistaken for 9 significant bits) would be violated


def test_bf16_rounding_is_nearest_even():
    x = np.array([1.0, 1.0 + 2.0 ** -8, 1.0 + 2.0 ** -7 + 2.0 ** -8, -3.14159, 1e-30, 65504.0], np.float32)
    r = bf16_rne(x)
    assert np.all(np.abs(r - x) <= np.abs(x) * 2.0 ** -8)
    assert r[1] == np.float32(1.0) and r[2] == np.float32(1.0 + 2.0 ** -6)  # ties go to the even mantissa


# ---- the byte mirror: per-row bound eps_r = c_r * rho + e0, c_r = 0.53 s / |x|, rho = |q|_1 / |q|_2 ---------------------------

E0 = 4.1 * (DIM + 8) * 2.0 ** -24 + 2e-6


def quantise(x, g):
    xp = (x.astype(np.float32) * (np.float32(1.0) / g.astype(np.float32))).astype(np.float32)  # normalised channels
    a = np.float32(np.abs(xp).max())
    inv = np.float32(127.0) / a
    u = np.rint(np.clip(xp * inv, -127, 127)).astype(np.int32) + 128
    s = a / np.float32(127.0)
    return u.astype(np.uint8), np.float32(s)


def coarse8(q, g, u, s, xx):
    q = q.astype(np.float32)
    qp = (q * g.astype(np.float32)).astype(np.float32)  # q' = q g, so that q.x = q'.x'
    acc = np.float32(0)
    for c in range(0, DIM, 64):
        acc = np.float32(acc + np.dot(qp[c:c + 64], u[c:c + 64].astype(np.float32)).astype(np.float32))
    qsum128 = np.float32(128.0) * np.float32(qp.sum(dtype=np.float32))
    dot = np.float32(s * (acc - qsum128))
    return np.float32(1) - dot / (np.sqrt(np.float32(np.dot(q, q))) * np.sqrt(np.float32(xx)))


def test_byte_bound_holds_on_random_and_on_worst_case_rows():
    rng = np.random.default_rng(1)
    tightest = 0.0
    for trial in range(400):
        q = rng.standard_normal(DIM).astype(np.float32)
        g = np.ones(DIM, np.float32)
        if trial % 4 == 0:
            x = rng.standard_normal(DIM).astype(np.float32) * np.float32(10.0 ** rng.integers(-6, 7))
        elif trial % 4 == 1:
            x = (q + 0.05 * rng.standard_normal(DIM)).astype(np.float32)
            x[rng.integers(DIM)] *= np.float32(50.0)   # one dominant element: the scale is set by it, everything else is coarse
        elif trial % 4 == 2:
            steps = rng.integers(-100, 101, DIM).astype(np.float32)
            steps[0] = 127.0
            x = steps + np.where(q > 0, 0.49, -0.49).astype(np.float32) * (1 if trial % 8 == 2 else -1)  # every element half a step off
            x[0] = 127.0
        else:
            # outlier channels (three dimensions 80 x the rest) with channel scales that are only roughly their RMS
            x = rng.standard_normal(DIM).astype(np.float32)
            x[[7, 133, 500]] *= np.float32(80.0)
            g[[7, 133, 500]] = np.float32(80.0) * rng.uniform(0.5, 2.0, 3).astype(np.float32)
            g *= rng.uniform(0.7, 1.4, DIM).astype(np.float32)
        u, s = quantise(x, g)
        xx = float(np.dot(x.astype(np.float64), x.astype(np.float64)))
        exact = float(cos_dist32(q, x))
        coarse = float(coarse8(q, g, u, s, xx))
        rho = float(np.abs(q.astype(np.float64) * g).sum() / np.sqrt(np.dot(q.astype(np.float64), q.astype(np.float64))))
        eps = 0.53 * float(s) / np.sqrt(xx) * rho + E0
        err = abs(coarse - exact)
        assert err <= eps, (trial, err, eps)
        if trial % 4 == 2:
            tightest = max(tightest, err / eps)
        if trial % 4 == 3:
            assert eps < 2e-2  # the channel scales keep the bound small where one scale per row alone gives ~0.2
    assert tightest > 0.85  # the constructed rows use most of the bound: it is not loose by construction


# ---- the group stage 1 on the matrix pipe (knn_scan_coarse8_mfma_kernel): the query as three signed 7-bit digits, exact
# integer dot products against the signed bytes; the bound's rho grows by rho_q = 1.25 x 127 dim S / |q| -----------------------

def query_digits(q, g):
    """knn_query_digits_kernel: q' = q g, S = max|q'| / 2^20, Q = rint(q' / S) clamped, Q = 16384 a + 128 b + c"""
    qp = (q.astype(np.float32) * g.astype(np.float32)).astype(np.float32)
    S = np.float32(np.abs(qp).max()) * np.float32(2.0 ** -20)
    inv = np.float32(1.0) / S if S > 0 else np.float32(0.0)
    Q = np.rint(np.clip((qp * inv).astype(np.float32), -1048576.0, 1048576.0)).astype(np.int64)
    c = ((Q + 64) & 127) - 64
    Q1 = (Q - c) >> 7
    b = ((Q1 + 64) & 127) - 64
    a = (Q1 - b) >> 7
    assert np.abs(a).max() <= 64 and np.abs(b).max() <= 64 and np.abs(c).max() <= 64
    assert np.array_equal(16384 * a + 128 * b + c, Q)
    return a, b, c, S, qp


def coarse8_digits(q, g, u, s, xx):
    a, b, c, S, qp = query_digits(q, g)
    v = u.astype(np.int64) - 128                                   # the mirror's byte with the top bit flipped, as a signed byte
    A, B, C = int(np.dot(a, v)), int(np.dot(b, v)), int(np.dot(c, v))   # what v_mfma_i32_16x16x64_i8 accumulates: exact
    assert max(abs(A), abs(B), abs(C)) < 2 ** 24                   # exact as floats too
    D = np.float32(np.float32(np.float32(A) * np.float32(16384.0) + np.float32(B) * np.float32(128.0)) + np.float32(C))
    dot = np.float32(s * np.float32(S * D))
    sq = np.sqrt(np.float32(np.dot(q.astype(np.float32), q.astype(np.float32))))
    rho = (np.abs(qp.astype(np.float64)).sum() / float(sq) + 1.25 * 127.0 * DIM * float(S) / float(sq)) * 1.000001
    return np.float32(1) - dot / (sq * np.sqrt(np.float32(xx))), rho


def test_digit_form_of_the_byte_bound_holds_and_costs_a_percent():
    rng = np.random.default_rng(2)
    tightest, dearest = 0.0, 0.0
    for trial in range(500):
        q = rng.standard_normal(DIM).astype(np.float32)
        g = np.ones(DIM, np.float32)
        kind = trial % 5
        if kind == 0:
            x = rng.standard_normal(DIM).astype(np.float32) * np.float32(10.0 ** rng.integers(-6, 7))
        elif kind == 1:
            x = (q + 0.05 * rng.standard_normal(DIM)).astype(np.float32)
            x[rng.integers(DIM)] *= np.float32(50.0)
        elif kind == 2:   # rows at the worst case of the byte rounding (as above)
            steps = rng.integers(-100, 101, DIM).astype(np.float32)
            steps[0] = 127.0
            x = steps + np.where(q > 0, 0.49, -0.49).astype(np.float32) * (1 if trial % 10 == 2 else -1)
            x[0] = 127.0
        elif kind == 3:   # a query with one dominant element: S is set by it, every other element keeps few digits
            q[rng.integers(DIM)] = np.float32(300.0)
            x = rng.choice([-127.0, 127.0], DIM).astype(np.float32)      # |x^|_1 at its maximum: the digit term at its worst
        else:             # outlier channels with rough channel scales, query elements at the worst case of the digit rounding
            x = rng.standard_normal(DIM).astype(np.float32)
            x[[7, 133, 500]] *= np.float32(80.0)
            g[[7, 133, 500]] = np.float32(80.0) * rng.uniform(0.5, 2.0, 3).astype(np.float32)
            g *= rng.uniform(0.7, 1.4, DIM).astype(np.float32)
            S = np.float32(np.abs(q * g).max()) * np.float32(2.0 ** -20)
            q = ((np.rint(q * g / S) + 0.499 * np.sign(x)) * S / g).astype(np.float32)   # half a digit step off, signs aligned with the row
        u, s = quantise(x, g)
        xx = float(np.dot(x.astype(np.float64), x.astype(np.float64)))
        exact = float(cos_dist32(q, x))
        coarse, rho = coarse8_digits(q, g, u, s, xx)
        rho_plain = float(np.abs(q.astype(np.float64) * g).sum() / np.sqrt(np.dot(q.astype(np.float64), q.astype(np.float64))))
        eps = 0.53 * float(s) / np.sqrt(xx) * rho + E0
        err = abs(float(coarse) - exact)
        assert err <= eps, (trial, kind, err, eps)
        if kind == 2:
            tightest = max(tightest, err / eps)
        if kind != 3:
            dearest = max(dearest, rho / rho_plain - 1.0)
    assert tightest > 0.85       # still not loose by construction
    assert dearest < 0.05        # the query's own quantisation widens the band by a few percent at most on ordinary queries
