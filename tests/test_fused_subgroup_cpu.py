"""The signature's subgroup test read off the Miller loop (milagro_bls_amd/csrc/mbls_lanes.h lane_sig_verdict; reference src/signature.rs:29-31,
subgroup_check_g2): a CPU statement of the loop's point arithmetic -- the homogeneous doubling and mixed-addition formulas of
tools/gen_tower_d.py prog_miller_dbl_d / prog_miller_add_d, WITHOUT any case handling -- over the bits of |x|, on curve points inside and
outside G2, against [r]P = O from the big-integer model. Pins the argument the GPU path relies on: the verdict psi(Q) = -T (with Z != 0) is
exact, and every exceptional case of the incomplete formulas (T = -Q at the prefix 13 of |x| for points of order 13, infinity from then on)
ends with Z = 0."""
import random

import bls12_381 as M

X_ABS = M.X_ABS
H2 = 0x5d543a95414e7f1091d50792876a202cd91de4547085abaa68a205b2e5a7ddfa628f1cb4d9e82ef21537e293a6691ae1616ec6e786f0c70cf1c38e31c7238e5
add, sub, mul, sqr, neg = M.f2_add, M.f2_sub, M.f2_mul, M.f2_sqr, M.f2_neg
XI = (1, 1)


def k(a, c):
    return (a[0] * c % M.P, a[1] * c % M.P)


def dbl(T):
    """dbl_step of tools/gen_coop.py / prog_miller_dbl_d: homogeneous projective doubling on y^2 = x^3 + 4 xi"""
    Tx, Ty, Tz = T
    B = sqr(Ty); C = sqr(Tz)
    E = k(mul(XI, C), 12); F = k(E, 3)
    YZ2 = sub(sub(sqr(add(Ty, Tz)), B), C)
    x3 = k(mul(mul(Tx, Ty), sub(B, F)), 2)
    y3 = sub(sqr(add(B, F)), k(sqr(E), 12))
    z3 = k(mul(B, YZ2), 4)
    return (x3, y3, z3)


def madd(T, Q):
    """add_step with an affine Q (pair 0 of a verification): no test for T = +-Q or T = O"""
    Tx, Ty, Tz = T
    Qx, Qy = Q
    u = sub(mul(Qy, Tz), Ty); v = sub(mul(Qx, Tz), Tx)
    uu = sqr(u); vv = sqr(v); vvv = mul(v, vv); R = mul(vv, Tx)
    A = sub(sub(mul(uu, Tz), vvv), k(R, 2))
    return (mul(v, A), sub(mul(u, sub(R, A)), mul(vvv, Ty)), mul(vvv, Tz))


def loop_point(Q):
    T = (Q[0], Q[1], (1, 0))
    for bit in bin(X_ABS)[3:]:
        T = dbl(T)
        if bit == "1":
            T = madd(T, Q)
    return T


def verdict(Q):
    X, Y, Z = loop_point(Q)
    px, py = M.g2_psi(Q)
    return Z != (0, 0) and mul(px, Z) == X and mul(py, Z) == neg(Y), Z


def test_verdict_from_the_loops_running_point():
    rnd = random.Random(5)

    def curve_point():
        while True:
            x = (rnd.randrange(M.P), rnd.randrange(M.P)); y = M.f2_sqrt(add(mul(sqr(x), x), M.B2))
            if y is not None:
                return (x, y)
    assert M.g2_mul(curve_point(), H2 * M.R) is None
    g2pt = lambda: M.g2_mul(M.G2, rnd.randrange(1, M.R))
    for _ in range(3):                                             # points of G2: the loop ends at [|x|] Q, the verdict is true
        Q = g2pt()
        ok, Z = verdict(Q)
        X, Y, Zt = loop_point(Q)
        zi = M.f2_inv(Zt)
        assert ok and (mul(X, zi), mul(Y, zi)) == M.g2_mul(Q, X_ABS)
    for ell in (13, 23, 2713):                                     # outside G2: false, like [r] Q != O
        t = None
        while t is None:
            t = M.g2_mul(curve_point(), H2 * M.R // (ell * ell))
        for Q in (t, M.g2_add(g2pt(), t), M.g2_neg(t)):
            ok, Z = verdict(Q)
            assert not ok and M.g2_mul(Q, M.R) is not None
            if ell == 13 and Q in (t, M.g2_neg(t)):                # [12] Q = -Q meets the addition at the prefix 13: infinity, and it stays
                assert Z == (0, 0)
    for _ in range(3):
        Q = curve_point()
        assert verdict(Q)[0] == (M.g2_mul(Q, M.R) is None)
    Q = M.g2_mul(curve_point(), H2)                                # cofactor-cleared: in G2
    assert verdict(Q)[0] and M.g2_mul(Q, M.R) is None
