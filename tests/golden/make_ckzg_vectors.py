#!/usr/bin/env python3
"""Capture the c-kzg-4844 conformance vectors the reference repo carries (and never reads)
into one small JSON manifest.

Run in the build container only (needs /root/reference and PyYAML):
    python tests/golden/make_ckzg_vectors.py
Source: /root/reference/tests/<suite>/small/<case>/data.yaml (208 files, ~50 MB of YAML).
All ten distinct blob byte strings in those files are formula generated, so a
blob is stored as an id + sha256; tests/golden/blobs.py regenerates the bytes and checks the
digest. Everything else (z, y, commitments, proofs, expected outputs) is stored verbatim.
"""
import glob
import hashlib
import json
import os
import sys

import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import blobs as B  # noqa: E402

REF = "/root/reference/tests"
SUITES = ["blob_to_kzg_commitment", "compute_kzg_proof", "compute_blob_kzg_proof",
          "verify_kzg_proof", "verify_blob_kzg_proof", "verify_blob_kzg_proof_batch"]


def blob_ref(hexstr, table):
    raw = bytes.fromhex(hexstr[2:])
    h = hashlib.sha256(raw).hexdigest()
    for name in B.BLOB_IDS:
        if B.BLOB_SHA256[name] == h:
            assert B.make_blob(name) == raw
            table[name] = h
            return name
    raise SystemExit("blob with sha256 %s matches no formula" % h)


def main():
    out = {"source": "c-kzg-4844 test vectors as vendored in lambdaclass/lambdaworks_kzg tests/*/small/*/data.yaml",
           "ckzg_version_pin": "github.com/ethereum/c-kzg-4844 @ da83e45e9cef (fuzz/gen_corpus/go.mod:6)",
           "blob_sha256": {}, "suites": {}}
    for suite in SUITES:
        cases = []
        for f in sorted(glob.glob(os.path.join(REF, suite, "small", "*", "data.yaml"))):
            d = yaml.safe_load(open(f))
            inp = dict(d["input"])
            if "blob" in inp:
                inp["blob"] = blob_ref(inp["blob"], out["blob_sha256"])
            if "blobs" in inp:
                inp["blobs"] = [blob_ref(b, out["blob_sha256"]) for b in inp["blobs"]]
            cases.append({"case": os.path.basename(os.path.dirname(f)), "input": inp, "output": d["output"]})
        out["suites"][suite] = cases
    path = os.path.join(HERE, "ckzg_vectors.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=0, sort_keys=True)
    print("wrote", path, os.path.getsize(path), "bytes;",
          {k: len(v) for k, v in out["suites"].items()})


if __name__ == "__main__":
    main()
