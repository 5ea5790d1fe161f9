"""The evaluation-form quotient's formulas (fr_ops.hip: k_eval_quotient_evalform), on plain integers over a small domain.

No GPU, no library: what the kernel computes -- inv_i = 1 / (z - w_i) by ONE inversion over a product tree, y by the barycentric
formula (its sum taken as z sum p_i inv_i - sum p_i), q_i = (y - p_i) inv_i, and for z = w_m the limit y = p_m, q_m = -(1 / w_m) sum_(i != m) q_i w_i -- against the definition
q(x) = (p(x) - p(z)) / (x - z) evaluated on the domain (c-kzg-4844's compute_kzg_proof_impl is the reference of
the z-on-the-domain branch). The GPU tests (tests/test_gpu_lagrange.py) then hold the kernel to the coefficient-form path and the oracle."""
import random

R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
N = 64
W = pow(7, (R - 1) // N, R)           # a primitive N-th root of unity (7 generates Fr*)


def _brp(i, bits):
    return int(format(i, "0%db" % bits)[::-1], 2)


def _poly_eval(c, x):
    acc = 0
    for a in reversed(c):
        acc = (acc * x + a) % R
    return acc


def _quotient_coefficients(c, z):
    """Ruffini: (p(x) - p(z)) / (x - z)"""
    q = [0] * len(c)
    acc = 0
    for k in range(len(c) - 1, 0, -1):
        acc = (acc * z + c[k]) % R
        q[k - 1] = acc
    return q


def _evalform(p, w, z):
    """what the kernel does, step for step"""
    n = len(p)
    d = [(z - wi) % R for wi in w]
    m = d.index(0) if 0 in d else -1
    if m >= 0:
        d[m] = 1
    # product tree, one inversion, down again
    tree = [0] * n + d
    for j in range(n - 1, 0, -1):
        tree[j] = tree[2 * j] * tree[2 * j + 1] % R
    assert tree[1] != 0
    tree[1] = pow(tree[1], -1, R)
    for j in range(1, n):
        l, r = tree[2 * j], tree[2 * j + 1]
        tree[2 * j], tree[2 * j + 1] = tree[j] * r % R, tree[j] * l % R
    inv = tree[n:]
    if m < 0:
        # sum p_i w_i inv_i, without a product by w_i: w_i / (z - w_i) = z / (z - w_i) - 1
        bary = (z * sum(p[i] * inv[i] for i in range(n)) - sum(p)) % R
        assert bary == sum(p[i] * w[i] % R * inv[i] for i in range(n)) % R
        y = (pow(z, n, R) - 1) * pow(n, -1, R) % R * bary % R
    else:
        y = p[m]
    q = [(y - p[i]) * inv[i] % R for i in range(n)]
    if m >= 0:
        q[m] = -pow(w[m], -1, R) * sum(q[i] * w[i] for i in range(n) if i != m) % R
    return y, q


def test_evaluation_form_quotient_is_the_quotient_polynomial_on_the_domain():
    assert pow(W, N, R) == 1 and pow(W, N // 2, R) == R - 1
    bits = N.bit_length() - 1
    w = [pow(W, _brp(i, bits), R) for i in range(N)]
    for i in range(0, N, 2):
        assert w[i + 1] == R - w[i]           # the pairs the kernel reads one table entry for
    rnd = random.Random(11)
    for case in range(40):
        c = [rnd.randrange(R) for _ in range(N)]
        p = [_poly_eval(c, wi) for wi in w]
        z = w[rnd.randrange(N)] if case % 2 else rnd.randrange(R)
        if case == 1:
            z = 1
        if case == 3:
            z = R - 1
        y, q = _evalform(p, w, z)
        qc = _quotient_coefficients(c, z)
        assert y == _poly_eval(c, z)
        assert q == [_poly_eval(qc, wi) for wi in w]
