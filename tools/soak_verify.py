"""Self-consistency soak of the proof / verification pipeline through the host C ABI: batches of random sizes (every
internal route: host-thread validation and linear combinations, validation kernels + piece-split GPU linear combinations,
the sliced long path), honest batches must verify, batches with one element swapped must not. One JSON line per batch."""
import json, os, random, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests/golden')
import torch
import blobs as B
import lambdaworks_kzg_amd as K
budget_s = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
ts = K.TrustedSetup.from_file('tests/golden/trusted_setup.txt')
if os.environ.get('LWKZG_DIRECT'):
    ts.reserve(256); ts.enable_direct_table(int(os.environ['LWKZG_DIRECT']))
rnd = random.Random(20261002)
sizes = [2, 3, 6, 9, 17, 40, 64, 65, 100, 257, 511, 512, 700, 1024, 1025, 1500, 2300]
t_end = time.time() + budget_s
seed, total, bad = 10 ** 6, 0, 0
while time.time() < t_end:
    n = rnd.choice(sizes)
    mode = rnd.choice([K.MODE_REFERENCE, K.MODE_CKZG])
    K.set_mode(mode)
    data = B.synthetic_batch(seed, n, big_endian=(mode == K.MODE_REFERENCE)); seed += n
    comms = b"".join(K.blob_to_kzg_commitment_batch(data, ts))
    proofs = b"".join(K.compute_blob_kzg_proof_batch(data, comms, ts))
    ok_honest = K.verify_blob_kzg_proof_batch(data, comms, proofs, n, ts)
    i, j = rnd.randrange(n), rnd.randrange(n)
    while j == i:
        j = rnd.randrange(n)
    what = rnd.choice(["proof", "commitment", "blob"])
    if what == "proof":
        ok_tampered = K.verify_blob_kzg_proof_batch(data, comms, proofs[:48 * i] + proofs[48 * j:48 * j + 48] + proofs[48 * i + 48:], n, ts)
    elif what == "commitment":
        ok_tampered = K.verify_blob_kzg_proof_batch(data, comms[:48 * i] + comms[48 * j:48 * j + 48] + comms[48 * i + 48:], proofs, n, ts)
    else:
        bb = B.BYTES_PER_BLOB
        ok_tampered = K.verify_blob_kzg_proof_batch(data[:bb * i] + data[bb * j:bb * j + bb] + data[bb * i + bb:], comms, proofs, n, ts)
    # r06: the device-resident form on the same inputs (vmsm.hip behind r, validation beside the hash)
    dv = lambda b: torch.frombuffer(bytearray(b), dtype=torch.uint8).cuda()
    db, dc, dp = dv(data), dv(comms), dv(proofs)
    torch.cuda.synchronize()
    dev_honest = K.verify_blob_kzg_proof_batch_device(db.data_ptr(), dc.data_ptr(), dp.data_ptr(), n, ts)
    dp2 = dv(proofs[:48 * i] + proofs[48 * j:48 * j + 48] + proofs[48 * i + 48:])
    torch.cuda.synchronize()
    dev_tampered = K.verify_blob_kzg_proof_batch_device(db.data_ptr(), dc.data_ptr(), dp2.data_ptr(), n, ts)
    # single-blob entry points on one element
    one = K.verify_blob_kzg_proof(data[B.BYTES_PER_BLOB * i:B.BYTES_PER_BLOB * (i + 1)], comms[48 * i:48 * i + 48], proofs[48 * i:48 * i + 48], ts)
    wrong = ok_honest is not True or ok_tampered is not False or one is not True or dev_honest is not True or dev_tampered is not False
    bad += int(wrong); total += n
    print(json.dumps({"n": n, "mode": "reference" if mode == K.MODE_REFERENCE else "ckzg", "honest": ok_honest, "tampered": what,
                      "tampered_verdict": ok_tampered, "single": one, "device_honest": dev_honest, "device_tampered": dev_tampered, "mismatch": wrong}), flush=True)
print(json.dumps({"summary": True, "blobs": total, "mismatching_batches": bad}))
sys.exit(1 if bad else 0)
