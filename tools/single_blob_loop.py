"""N single-blob calls of each of the reference's symbols, for a profiler to sit on (tools/collect_profiles.sh: rocprofv3 --kernel-trace --stats
and --pmc passes over k_coop_msm_asm): `python tools/single_blob_loop.py [calls]`. Default engine (what a plain load selects)."""
import sys
sys.path.insert(0, '.')
sys.path.insert(0, 'tests/golden')
import blobs as B
import lambdaworks_kzg_amd as K

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
ts = K.TrustedSetup.from_file('tests/golden/trusted_setup.txt')
blob = B.synthetic_blob(1)
c = K.blob_to_kzg_commitment(blob, ts)
z = blob[32:64]
for _ in range(n):
    assert K.blob_to_kzg_commitment(blob, ts) == c
p = K.compute_blob_kzg_proof(blob, c, ts)
for _ in range(n):
    assert K.compute_blob_kzg_proof(blob, c, ts) == p
for _ in range(n):
    K.compute_kzg_proof(blob, z, ts)
print("%d commitments, %d blob proofs, %d point proofs of one blob each" % (n, n, n))
