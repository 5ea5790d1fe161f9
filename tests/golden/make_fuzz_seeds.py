#!/usr/bin/env python3
"""Capture the reference's fuzz seeds (/root/reference/fuzz/<target>/corpus/*, 115 inputs) as a replayable fixture.

Run in the build container only (needs /root/reference):
    python tests/golden/make_fuzz_seeds.py

Writes tests/golden/fuzz_seeds.json (index + the CPU oracle's answer for every seed in both modes) and
tests/golden/fuzz_seeds.xz (the bytes of the seeds whose harness calls its symbol; see tests/fuzz_cases.py). Only seed
BYTES travel -- data the reference's fuzzers hold, like its YAML vectors -- none of the harness code does."""
import glob
import hashlib
import json
import lzma
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_cases as F  # noqa: E402

REF = "/root/reference/fuzz"


def main():
    from oracle import oracle as O
    O.build()
    s = O.Settings.from_file(os.path.join(HERE, "trusted_setup.txt"), check_subgroup=False)
    seeds, stored = [], []
    for target in F.TARGETS:
        for path in sorted(glob.glob(os.path.join(REF, target, "corpus", "*"))):
            data = open(path, "rb").read()
            calls = F.harness_calls(target, len(data))
            keep = calls and not (target == "verify_blob_kzg_proof_batch" and len(data) < F.INPUT_SIZE[target])
            e = {"target": target, "name": os.path.basename(path), "size": len(data), "sha256": hashlib.sha256(data).hexdigest(),
                 "harness_calls": calls, "stored": keep}
            if calls:
                e["expect"] = {"reference": F.oracle_answer(O, s, target, data, O.MODE_R), "ckzg": F.oracle_answer(O, s, target, data, O.MODE_C)}
            if keep:
                stored.append(data)
            seeds.append(e)
    index = {"source": "/root/reference/fuzz/*/corpus (harness: fuzz/base_fuzz.h:17-34, fuzz/*/fuzz.c; sizes fuzz/Makefile:65-85)",
             "oracle": "oracle/ref_kzg.c on tests/golden/trusted_setup.txt (tau = 1337)", "seeds": seeds}
    with open(os.path.join(HERE, "fuzz_seeds.json"), "w") as f:
        json.dump(index, f, indent=1)
    with lzma.open(os.path.join(HERE, "fuzz_seeds.xz"), "w", preset=9 | lzma.PRESET_EXTREME) as f:
        f.write(b"".join(stored))
    print("%d seeds, %d call their symbol, %d stored (%d bytes)" % (len(seeds), sum(e["harness_calls"] for e in seeds), len(stored), sum(map(len, stored))))


if __name__ == "__main__":
    main()
