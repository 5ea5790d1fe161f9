"""GPU: environment-driven opt-in of the direct table (own module: the parity module's table fixtures, up to 240 GB, are
released before this one starts a second process on the same device)."""
import pytest

import blobs as B
from conftest import SETUP_PATH

pytestmark = pytest.mark.gpu


def test_direct_table_opt_in_from_environment(K, gpu_setup):
    """LWKZG_DIRECT_BITS: how a consumer that only knows the reference's nine symbols opts in (fresh process)."""
    import subprocess
    import sys
    import os
    from conftest import ROOT
    blob = B.synthetic_blob(4242)
    want = K.blob_to_kzg_commitment(blob, gpu_setup).hex()
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import blobs as B; import lambdaworks_kzg_amd as K; "
            "ts = K.TrustedSetup.from_file(%r); print(ts.direct_table_bits(), K.blob_to_kzg_commitment(B.synthetic_blob(4242), ts).hex())"
            % (ROOT, os.path.join(ROOT, "tests", "golden"), SETUP_PATH))
    env = dict(os.environ, LWKZG_DIRECT_BITS="14")
    out = subprocess.check_output([sys.executable, "-c", code], env=env).decode().split()
    assert out[-2:] == ["14", want]
