#!/usr/bin/env python3
"""Two more trusted setups for the GPU parity tests, so that the kernels are not only right for tau = 1337.

    python tests/golden/make_setups.py            # writes the two files below (about 15 s)
    python tests/golden/make_setups.py --check    # regenerates and compares with the committed files

* trusted_setup_tau2.txt -- a powers-of-tau' setup in the reference's text format (/root/reference/src/srs.rs:25-82:
  "4096", "65", 4096 compressed G1 points [tau'^i]G1, 65 compressed G2 points [tau'^i]G2) for a 255-bit tau' that has
  nothing to do with 1337: tau' = sha256("lambdaworks_kzg_amd tau' r03") mod r. Everything the tau = 1337 tests do carries
  over with conftest.tau_closed_form(oracle, scalars, tau=TAU2): commitments, both kinds of proof, verification.
* trusted_setup_unstructured.txt -- NOT powers of anything: P_i = [k_i]G1 with k_i = sha256("... unstructured" | i) mod r,
  i.e. 4096 points with no relation a fixed-base table could exploit by accident (what the reference's own integration
  test commits against: /root/reference/tests/lib_test.rs:68 -> src/utils.rs:84-107 create_srs with a random secret
  -- here even the power structure is gone). Its G2 half is the tau' one (commitments never read it). A commitment to
  scalars s is [sum s_i k_i]G1 (conftest.unstructured_closed_form) and oracle.msm_affine over the decompressed points.

G1 points come from the CPU oracle (oracle.g1_generator_mul); G2 points from the few lines of Fp2 / G2 arithmetic in this
file, which first reproduces all 65 G2 points of tests/golden/trusted_setup.txt (tau = 1337) as a self-check. Nothing
here is product code and nothing is read from /root/reference.
"""
import hashlib
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
N1, N2 = 4096, 65

TAU2 = int.from_bytes(hashlib.sha256(b"lambdaworks_kzg_amd tau' r03").digest(), "big") % R


def unstructured_scalar(i):
    return int.from_bytes(hashlib.sha256(b"lambdaworks_kzg_amd unstructured r03" + i.to_bytes(4, "big")).digest(), "big") % R


# ---- Fp2 = Fp[u] / (u^2 + 1), G2: y^2 = x^3 + 4 (1 + u), Jacobian coordinates ------------------------------------

def f2_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def f2_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def f2_inv(a):
    d = pow(a[0] * a[0] + a[1] * a[1], P - 2, P)
    return (a[0] * d % P, -a[1] * d % P)


G2_GEN = ((0x024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8,
           0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e),
          (0x0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801,
           0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be))


def g2_double(p):
    x, y, z = p
    if z == (0, 0):
        return p
    a = f2_mul(x, x)
    b = f2_mul(y, y)
    c = f2_mul(b, b)
    t = f2_add(x, b)
    d = f2_sub(f2_sub(f2_mul(t, t), a), c)
    d = f2_add(d, d)
    e = f2_add(f2_add(a, a), a)
    f = f2_mul(e, e)
    x3 = f2_sub(f, f2_add(d, d))
    c8 = f2_add(c, c)
    c8 = f2_add(c8, c8)
    c8 = f2_add(c8, c8)
    y3 = f2_sub(f2_mul(e, f2_sub(d, x3)), c8)
    z3 = f2_mul(y, z)
    return (x3, y3, f2_add(z3, z3))


def g2_add_affine(p, q):
    """Jacobian p + affine q (q != infinity)."""
    x1, y1, z1 = p
    if z1 == (0, 0):
        return (q[0], q[1], (1, 0))
    z1z1 = f2_mul(z1, z1)
    u2 = f2_mul(q[0], z1z1)
    s2 = f2_mul(f2_mul(q[1], z1), z1z1)
    h = f2_sub(u2, x1)
    r = f2_sub(s2, y1)
    if h == (0, 0):
        if r == (0, 0):
            return g2_double(p)
        return ((1, 0), (1, 0), (0, 0))
    hh = f2_mul(h, h)
    hhh = f2_mul(h, hh)
    v = f2_mul(x1, hh)
    x3 = f2_sub(f2_sub(f2_mul(r, r), hhh), f2_add(v, v))
    y3 = f2_sub(f2_mul(r, f2_sub(v, x3)), f2_mul(y1, hhh))
    return (x3, y3, f2_mul(z1, h))


def g2_mul_generator(k):
    acc = ((1, 0), (1, 0), (0, 0))
    for bit in bin(k % R)[2:]:
        acc = g2_double(acc)
        if bit == "1":
            acc = g2_add_affine(acc, G2_GEN)
    return acc


def g2_compress(p):
    """ZCash serialisation: x.c1 | x.c0 big-endian, bit 7 = compressed, bit 6 = infinity, bit 5 = y is the larger root
    (c1 compared first, then c0)."""
    x, y, z = p
    if z == (0, 0):
        return bytes([0xC0]) + bytes(95)
    zi = f2_inv(z)
    zi2 = f2_mul(zi, zi)
    ax = f2_mul(x, zi2)
    ay = f2_mul(y, f2_mul(zi2, zi))
    neg = ((-ay[0]) % P, (-ay[1]) % P)
    larger = (ay[1], ay[0]) > (neg[1], neg[0])
    out = bytearray(ax[1].to_bytes(48, "big") + ax[0].to_bytes(48, "big"))
    out[0] |= 0x80 | (0x20 if larger else 0)
    return bytes(out)


def g2_powers(tau):
    out, t = [], 1
    for _ in range(N2):
        out.append(g2_compress(g2_mul_generator(t)))
        t = t * tau % R
    return out


def self_check_against_tau_1337():
    with open(os.path.join(HERE, "trusted_setup.txt")) as f:
        lines = f.read().split()
    want = [bytes.fromhex(x) for x in lines[2 + N1:2 + N1 + N2]]
    assert g2_powers(1337) == want, "the G2 arithmetic of this script does not reproduce the tau = 1337 setup"


def setup_text(g1, g2):
    return "%d\n%d\n" % (N1, N2) + "".join(x.hex() + "\n" for x in g1) + "".join(x.hex() + "\n" for x in g2)


def build_texts(n1=N1):
    from oracle import oracle as O
    O.build()
    g2 = g2_powers(TAU2)
    g1_tau, t = [], 1
    for _ in range(n1):
        g1_tau.append(O.g1_generator_mul(t))
        t = t * TAU2 % R
    g1_un = [O.g1_generator_mul(unstructured_scalar(i)) for i in range(n1)]
    return g1_tau, g1_un, g2


FILES = ("trusted_setup_tau2.txt", "trusted_setup_unstructured.txt")


def main():
    self_check_against_tau_1337()
    g1_tau, g1_un, g2 = build_texts()
    texts = (setup_text(g1_tau, g2), setup_text(g1_un, g2))
    if "--check" in sys.argv:
        for name, text in zip(FILES, texts):
            with open(os.path.join(HERE, name)) as f:
                assert f.read() == text, name
        print("committed setups match their generator")
        return
    for name, text in zip(FILES, texts):
        with open(os.path.join(HERE, name), "w") as f:
            f.write(text)
        print("wrote", name, hashlib.sha256(text.encode()).hexdigest())


if __name__ == "__main__":
    main()
