"""lambdaworks_kzg_amd -- MI355X-native KZG / EIP-4844 blob-commitment engine.

Host-side mirror of lambdaclass/lambdaworks_kzg's C ABI for the blob-commitment hot path
(blob_to_kzg_commitment / compute_kzg_proof / compute_blob_kzg_proof + trusted-setup load) over
hand-written HIP kernels for gfx950. See DESIGN.md; the C header is include/lambdaworks_kzg_amd.h.
"""
from .capi import (  # noqa: F401
    BYTES_PER_BLOB, BYTES_PER_COMMITMENT, BYTES_PER_PROOF, C_KZG_BADARGS, C_KZG_ERROR, C_KZG_MALLOC, C_KZG_OK,
    FIELD_ELEMENTS_PER_BLOB, MODE_CKZG, MODE_REFERENCE, KzgError, KZGSettings, TrustedSetup,
    blob_to_kzg_commitment, blob_to_kzg_commitment_batch, blob_to_kzg_commitment_batch_device, commit_and_prove_batch_device,
    compute_blob_kzg_proof, compute_blob_kzg_proof_batch, compute_blob_kzg_proof_batch_device,
    compute_kzg_proof, compute_kzg_proof_batch, get_mode, knob_report, lib, set_device, set_mode,
    VerifyShard, verify_shards_finish,
    verify_blob_kzg_proof, verify_blob_kzg_proof_batch, verify_blob_kzg_proof_batch_device, verify_kzg_proof,
)

__all__ = [n for n in dir() if not n.startswith("_")]
