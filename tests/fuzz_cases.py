"""The reference's fuzz seeds (/root/reference/fuzz/*/corpus/, harness shape fuzz/base_fuzz.h:17-34 and fuzz/*/fuzz.c) as a
replayable fixture: what each harness would pass to its symbol for each seed, and what the CPU oracle answers.

tests/golden/fuzz_seeds.json is the index (target, seed name, size, sha256, the oracle's answer in both modes);
tests/golden/fuzz_seeds.xz holds the bytes of the seeds whose harness really calls the symbol (xz of their concatenation:
libFuzzer's mutations share most of their bytes). Both are written by tests/golden/make_fuzz_seeds.py. Test infrastructure."""
import hashlib
import json
import lzma
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BLOB = 131072
# fuzz/Makefile:65-85 and the INPUT_SIZE of each fuzz.c
INPUT_SIZE = {"blob_to_kzg_commitment": BLOB, "compute_kzg_proof": BLOB + 32, "compute_blob_kzg_proof": BLOB + 48,
              "verify_kzg_proof": 48 + 32 + 32 + 48, "verify_blob_kzg_proof": BLOB + 48 + 48,
              "verify_blob_kzg_proof_batch": BLOB + 48 + 48}
TARGETS = list(INPUT_SIZE)


def harness_calls(target, size):
    """Does the reference's harness call the symbol for a seed of this size? (every fuzz.c: only at exactly INPUT_SIZE;
    the batch harness always calls, with count = size / INPUT_SIZE)"""
    return target == "verify_blob_kzg_proof_batch" or size == INPUT_SIZE[target]


def batch_split(data):
    """verify_blob_kzg_proof_batch/fuzz.c: count = size / INPUT_SIZE; blobs at 0, commitments at BLOB * count, proofs at
    (BLOB + 48) * count"""
    count = len(data) // INPUT_SIZE["verify_blob_kzg_proof_batch"]
    return count, data[:BLOB * count], data[BLOB * count:BLOB * count + 48 * count], data[(BLOB + 48) * count:(BLOB + 96) * count]


def load_seeds():
    """[(target, name, bytes or None)] in index order; bytes is None for seeds whose harness never reads them."""
    with open(os.path.join(GOLDEN, "fuzz_seeds.json")) as f:
        index = json.load(f)
    with lzma.open(os.path.join(GOLDEN, "fuzz_seeds.xz")) as f:
        blob = f.read()
    out, off = [], 0
    for e in index["seeds"]:
        data = None
        if e["stored"]:
            data = blob[off:off + e["size"]]
            off += e["size"]
            assert hashlib.sha256(data).hexdigest() == e["sha256"], e["name"]
        out.append((e, data))
    assert off == len(blob)
    return index, out


def oracle_verify_blob(O, s, blob, commitment, proof, mode, tau):
    """verify_blob_kzg_proof (/root/reference/src/lib.rs:456-505) from the oracle's pieces: challenge, evaluation, and the
    known-tau form of the pairing check. Returns (rc, ok)."""
    bad = O.ERROR if mode == O.MODE_R else O.BADARGS
    if O.g1_decompress(commitment) is None:
        return bad, False
    rc, z = O.compute_challenge(blob, commitment, mode)
    if rc != O.OK:
        return rc, False
    rc, _, y = O.compute_kzg_proof(blob, z, s, mode)
    if rc != O.OK:
        return rc, False
    if O.g1_decompress(proof) is None:
        return bad, False
    return O.verify_kzg_proof_known_tau(commitment, z, y, proof, tau, mode)


def oracle_answer(O, s, target, data, mode, tau=1337):
    """What the symbol must return for this seed: {"rc": .., "out": hex or bool or None}."""
    if target == "blob_to_kzg_commitment":
        rc, cm = O.blob_to_kzg_commitment(data, s, mode)
        return {"rc": rc, "out": cm.hex() if rc == O.OK else None}
    if target == "compute_kzg_proof":
        rc, pr, y = O.compute_kzg_proof(data[:BLOB], data[BLOB:], s, mode)
        return {"rc": rc, "out": (pr + y).hex() if rc == O.OK else None}
    if target == "compute_blob_kzg_proof":
        rc, pr = O.compute_blob_kzg_proof(data[:BLOB], data[BLOB:], s, mode)
        return {"rc": rc, "out": pr.hex() if rc == O.OK else None}
    if target == "verify_kzg_proof":
        rc, ok = O.verify_kzg_proof_known_tau(data[:48], data[48:80], data[80:112], data[112:160], tau, mode)
        return {"rc": rc, "out": ok if rc == O.OK else None}
    if target == "verify_blob_kzg_proof":
        rc, ok = oracle_verify_blob(O, s, data[:BLOB], data[BLOB:BLOB + 48], data[BLOB + 48:], mode, tau)
        return {"rc": rc, "out": ok if rc == O.OK else None}
    assert target == "verify_blob_kzg_proof_batch"
    count, blobs, comms, proofs = batch_split(data)
    if count == 0:      # lib.rs:538-543: OK with ok = false; c-kzg-4844 (mode C) accepts the empty batch
        return {"rc": O.OK, "out": mode == O.MODE_C}
    verdict = True
    for i in range(count):
        rc, ok = oracle_verify_blob(O, s, blobs[BLOB * i:BLOB * (i + 1)], comms[48 * i:48 * i + 48], proofs[48 * i:48 * i + 48], mode, tau)
        if rc != O.OK:
            return {"rc": rc, "out": None}
        verdict = verdict and ok
    return {"rc": O.OK, "out": verdict}
