"""CPU: a second opinion on the pairing (Missing 7). oracle/ref_pairing.c is a deliberately plain optimal ate pairing --
Fp12 as Fp2[w]/(w^6 - xi) with a schoolbook product, an affine Miller loop on the twist, the final exponentiation as ONE
square-and-multiply over (p^12 - 1)/r -- that shares no code and no structure with the product's host pairing (csrc/pairing.hip: a
2-3-2 tower, projective steps with cached lines, a cyclotomic hard part). Both are asked the same questions on random [a]G1 x [b]G2
inputs: products that are one, products that are not, members at infinity, and both ZCash sign-bit encodings of the G2 points (the bit
the reference's decompress_g2_point does not read, /root/reference/src/compression.rs:105-139; /root/reference/src/lib.rs:407-453 and
src/utils.rs:224-236 are where the reference pairs). Nothing in the product imports the oracle; the product's side is its host-only
hook lwkzg_pairing_product_is_one."""
import random

from conftest import P, R, SETUP_PATH, TAU


def _g1(oracle, k):
    """([k]G1 compressed, affine x | y with all-zero for infinity)"""
    c = oracle.g1_generator_mul(k % R)
    xy, inf = oracle.g1_decompress(c)
    return c, (bytes(96) if inf else xy)


def _g1_neg(c, xy):
    if xy == bytes(96):
        return c, xy
    y = int.from_bytes(xy[48:], "big")
    b = bytearray(c)
    b[0] ^= 0x20
    return bytes(b), xy[:48] + ((P - y) % P).to_bytes(48, "big")


def _g2(oracle, k):
    xy = oracle.g2_generator_mul(k % R)
    return oracle.g2_compress(xy), xy


def _g2_neg_xy(xy):
    if xy == bytes(192):
        return xy
    y0, y1 = int.from_bytes(xy[96:144], "big"), int.from_bytes(xy[144:], "big")
    return xy[:96] + ((P - y0) % P).to_bytes(48, "big") + ((P - y1) % P).to_bytes(48, "big")


def test_final_exponent_constant(oracle):
    assert oracle.final_exponent() == (P ** 12 - 1) // R and (P ** 12 - 1) % R == 0


def test_g2_generator_and_compression_against_the_setup_file(oracle):
    """the oracle's G2 generator and its ZCash compression are the first two G2 lines of the reference's trusted setup: G2 and [tau]G2"""
    lines = open(SETUP_PATH).read().split()
    g2 = [bytes.fromhex(x) for x in lines[2 + 4096:2 + 4096 + 65]]
    for k in (0, 1, 2, 64):
        c, xy = _g2(oracle, pow(TAU, k, R))
        assert oracle.g2_on_curve(xy) and c == g2[k], k
    assert _g2(oracle, R)[0] == bytes([0xc0]) + bytes(95)        # [r]G2 = O


def test_oracle_pairing_is_bilinear_and_non_degenerate(oracle):
    rnd = random.Random(2024)
    a, b = rnd.randrange(1, R), rnd.randrange(1, R)
    (_, A), (_, nAB) = _g1(oracle, a), _g1_neg(*_g1(oracle, a * b))
    (_, Q), (_, H) = _g2(oracle, b), _g2(oracle, 1)
    assert oracle.pairing_product_is_one([A, nAB], [Q, H]) is True               # e(aG, bH) e(-abG, H) = 1
    assert oracle.pairing_product_is_one([A], [H]) is False                      # e(aG, H) != 1
    assert oracle.pairing_product_is_one([A, _g1(oracle, a * b)[1]], [Q, H]) is False
    assert oracle.pairing_product_is_one([bytes(96), A], [H, bytes(192)]) is True  # infinity on either side contributes one


def test_product_pairing_agrees_with_the_oracle_on_random_products(K, oracle):
    from lambdaworks_kzg_amd import capi
    rnd = random.Random(77)
    n_true = n_false = 0
    for case in range(24):
        npairs = 2 + case % 3                                # 2, 3, 4 pairs (the hook's maximum)
        a = [rnd.randrange(1, R) for _ in range(npairs - 1)]
        b = [rnd.randrange(1, R) for _ in range(npairs - 1)]
        honest = case % 2 == 0
        total = sum(x * y for x, y in zip(a, b)) % R
        g1c, g1a, g2c, g2a = [], [], [], []
        for x, y in zip(a, b):
            c, xy = _g1(oracle, x)
            qc, qxy = _g2(oracle, y)
            g1c.append(c); g1a.append(xy); g2c.append(qc); g2a.append(qxy)
        # the closing pair: e(-[sum a_i b_i (+ 1)]G1, G2)
        c, xy = _g1_neg(*_g1(oracle, total + (0 if honest else 1)))
        qc, qxy = _g2(oracle, 1)
        g1c.append(c); g1a.append(xy); g2c.append(qc); g2a.append(qxy)
        if case % 5 == 4:                                    # a member at infinity: on the G1 side, or on the G2 side
            if case % 2:
                g1c[0], g1a[0] = bytes([0xc0]) + bytes(47), bytes(96)
            else:
                g2c[0], g2a[0] = bytes([0xc0]) + bytes(95), bytes(192)
        want = oracle.pairing_product_is_one(g1a, g2a)
        got = capi.pairing_product_is_one(b"".join(g1c), b"".join(g2c))
        assert got is want, (case, npairs, honest)
        n_true += want
        n_false += not want
        # the other encoding of the first G2 point: the sign bit flipped names -Q, which the product must honour
        if g2a[0] != bytes(192):
            flipped = bytearray(g2c[0])
            flipped[0] ^= 0x20
            want2 = oracle.pairing_product_is_one(g1a, [_g2_neg_xy(g2a[0])] + g2a[1:])
            got2 = capi.pairing_product_is_one(b"".join(g1c), bytes(flipped) + b"".join(g2c[1:]))
            assert got2 is want2, (case, "sign bit")
            if npairs == 2 and honest and case % 5 != 4:
                assert want is True and want2 is False       # e(aG, -bH) e(-abG, H) = e(G, H)^(-2ab) != 1
    assert n_true >= 8 and n_false >= 8
