"""r06 call 23: how fast do the host threads hash on their own? challenge_digests_host over 512 / 4096 blobs in one call, by grain."""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import blobs as B
from lambdaworks_kzg_amd import capi
n = 4096
h = B.synthetic_batch(9000, n)
c = bytes(48 * n)
for m in (512, 4096, 512, 4096):
    ts = []
    for rep in range(6):
        t = time.perf_counter(); capi.challenge_digests_host(h[:m * B.BYTES_PER_BLOB] if m < n else h, c[:48 * m]); ts.append(time.perf_counter() - t)
    print("grain %s: %d blobs: best %.2f ms (%.1f GB/s), all %s" % (os.environ.get("LWKZG_HOST_HASH_GRAIN", "default"), m, min(ts) * 1e3, m * 131072 / min(ts) / 1e9, ["%.2f" % (x * 1e3) for x in ts]))
