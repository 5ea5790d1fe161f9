import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, GOLDEN)

SETUP_PATH = os.path.join(GOLDEN, "trusted_setup.txt")
R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
TAU = 1337  # the secret of tests/golden/trusted_setup.txt (consensus-specs testing setup), SURVEY 0.4


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def hx(s):
    return bytes.fromhex(s[2:] if s.startswith("0x") else s)


@pytest.fixture(scope="session")
def vectors():
    with open(os.path.join(GOLDEN, "ckzg_vectors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def oracle_setup(oracle):
    # subgroup checks of all 4096 points take ~2.5 s; test_oracle_setup_subgroup covers them once
    return oracle.Settings.from_file(SETUP_PATH, check_subgroup=False)


@pytest.fixture(scope="session")
def K():
    import lambdaworks_kzg_amd as K
    K.lib()
    return K


@pytest.fixture(scope="session")
def gpu_setup(K):
    """What a consumer of the nine reference symbols gets: a plain load, engine chosen by the library (on an empty
    MI355X the 13-bit direct table, engine.hip: direct_from_env)."""
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    assert not os.environ.get("LWKZG_DIRECT_BITS"), "the GPU tests expect the library's own choice of engine"
    ts = K.TrustedSetup.from_file(SETUP_PATH)
    ts.default_bits = ts.direct_table_bits()
    yield ts
    ts.free()


@pytest.fixture(scope="session")
def bucket_setup(K):
    """The low-memory engine (9 MB table, Pippenger buckets), forced."""
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    ts = K.TrustedSetup.from_file(SETUP_PATH)
    ts.enable_direct_table(0)
    assert ts.direct_table_bits() == 0
    yield ts
    ts.free()


@pytest.fixture(scope="session", params=["default", "bucket"])
def engine_setup(request, gpu_setup, bucket_setup):
    """The two engines a plain load can end up on."""
    return gpu_setup if request.param == "default" else bucket_setup


def tau_closed_form(oracle, scalars, tau=TAU):
    """[sum s_i tau^i] G compressed: the closed-form commitment for a powers-of-tau setup (default: the tau = 1337 one)."""
    acc, t = 0, 1
    for s in scalars:
        acc = (acc + s * t) % R
        t = t * tau % R
    return oracle.g1_generator_mul(acc)


SETUP_TAU2_PATH = os.path.join(GOLDEN, "trusted_setup_tau2.txt")                   # tests/golden/make_setups.py
SETUP_UNSTRUCTURED_PATH = os.path.join(GOLDEN, "trusted_setup_unstructured.txt")


def unstructured_closed_form(oracle, scalars):
    """[sum s_i k_i] G compressed: the closed-form commitment for tests/golden/trusted_setup_unstructured.txt (P_i = [k_i]G)."""
    import make_setups as M
    if not hasattr(unstructured_closed_form, "k"):
        unstructured_closed_form.k = [M.unstructured_scalar(i) for i in range(4096)]
    return oracle.g1_generator_mul(sum(s * k for s, k in zip(scalars, unstructured_closed_form.k)) % R)
