#!/bin/bash
# r06 call 16: ONE blob_to_kzg_commitment with nothing copied (the parse kernel reads the pinned slot, the sum is stored into pinned memory, counters
# cleared by the parse kernel, the redo flag a pinned word) against r05's copies and fills; the cooperative kernel's parity tests through it
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_coop.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "single or one_blob or sixteen or threads or lib_test or fuzz or mirror or vectors" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_fuzz_seeds.py tests/test_gpu_setups.py -x -q -m gpu 2>&1 | tail -3
python /dev/stdin <<'PY' 2>&1 | tail -4
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import blobs as B
import lambdaworks_kzg_amd as K
ts = K.TrustedSetup.from_file('tests/golden/trusted_setup.txt')
blob = B.synthetic_blob(1)
for _ in range(50): K.blob_to_kzg_commitment(blob, ts)
for rep in range(3):
    t = time.perf_counter()
    for _ in range(500): K.blob_to_kzg_commitment(blob, ts)
    print("zero-copy commit: %.4f ms per call" % ((time.perf_counter() - t) / 500 * 1e3))
PY
LWKZG_EXPERIMENTAL=1 LWKZG_ZERO_COPY=0 python /dev/stdin <<'PY' 2>&1 | tail -4
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import blobs as B
import lambdaworks_kzg_amd as K
ts = K.TrustedSetup.from_file('tests/golden/trusted_setup.txt')
blob = B.synthetic_blob(1)
for _ in range(50): K.blob_to_kzg_commitment(blob, ts)
for rep in range(3):
    t = time.perf_counter()
    for _ in range(500): K.blob_to_kzg_commitment(blob, ts)
    print("r05 arm commit: %.4f ms per call" % ((time.perf_counter() - t) / 500 * 1e3))
PY
python tools/single_blob_timing.py 2>&1 | tail -8
