#!/usr/bin/env python3
"""The 1 -> 8 GPU curve of every BASELINE config, PROJECTED from one-GPU measurements.

No 8-GPU node has ever run this repository, so nothing here is a scaling MEASUREMENT. What can be measured on ONE MI355X is one
rank's share of each config at G = 1, 2, 4, 8 -- the shards are independent, every rank holds the full setup, and the data path has no
collective except the two small all-gathers of the batch verification and the one-point gather of the tiled MSM -- plus the cost of
issuing those collectives through RCCL at world size 1 (launch + synchronisation floor; the bytes are priced separately at one xGMI
link, 153 GB/s, the ring's per-link bound). From these:

    configs[1]  commitments, 1024 blobs per rank (weak):      value(G) = G * 1024 / t_commit(1024)
    configs[2]  compute_blob_kzg_proof, 256 blobs (strong):   T(G) = t_proof(256 / G)
    configs[3]  verify_blob_kzg_proof_batch, 4096 (strong):   T(G) = t_begin(4096 / G) + allgather(160 B x 4096) + t_partial(4096 / G of 4096)
                                                                     + allgather(329 B x G) + t_finish(G partial sums)
    configs[4]  2^20-term MSM, 256 tiles (strong):            T(G) = t_msm(256 / G tiles) + allgather(48 B x G) + host sum

and efficiency(G) = T(1) / (G * T(G)) for the strong-scaling shapes. Which of them are latency-bound (a shard that is a flat latency
chain gains nothing from being smaller) is the point of the table.

    python tools/scaling_projection.py [--out profiles/r06_scaling_projection.json] [--md]

Writes the JSON and prints the markdown table DESIGN.md section 7 carries. Every number is labelled "projection, unmeasured on hardware".
"""
import argparse, json, os, statistics, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

XGMI_LINK_GBS = 153.0          # one xGMI link, the per-link bound of a ring collective (MI355X_MICROARCH.md)
GS = (1, 2, 4, 8)


def rccl_world1():
    """launch + completion floor of the collectives the path issues, through RCCL at world size 1, in a fresh process"""
    code = r"""
import os, sys, time, json, statistics, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29731")
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
def t(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(dev); xs = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(dev); xs.append((time.perf_counter() - t0) * 1e3)
    return statistics.median(xs)
out = {}
for name, nbytes in (("broadcast_setup_image", int(sys.argv[1])), ("allgather_records_4096", 160 * 4096), ("allgather_partials", 329), ("allgather_point", 48)):
    buf = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    if name.startswith("broadcast"):
        out[name] = {"bytes": nbytes, "world1_ms": t(lambda: dist.broadcast(buf, src=0))}
    else:
        parts = [torch.empty_like(buf)]
        out[name] = {"bytes": nbytes, "world1_ms": t(lambda: dist.all_gather(parts, buf))}
out["nccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
print(json.dumps(out)); dist.destroy_process_group()
"""
    from lambdaworks_kzg_amd import capi
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, "-c", code, str(capi.setup_image_bytes())], env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if p.returncode != 0 or not lines:
        return {"error": (p.stderr or p.stdout)[-600:]}
    return json.loads(lines[-1])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_scaling_projection.json"))
    ap.add_argument("--reps", type=int, default=12)
    a = ap.parse_args()
    coll = rccl_world1()          # before this process touches the GPU (a fresh process initialises RCCL)
    import numpy as np
    import torch
    import blobs as B
    import lambdaworks_kzg_amd as K
    from lambdaworks_kzg_amd import capi
    dev = torch.device("cuda:0")
    ts = K.TrustedSetup.from_file(os.path.join(ROOT, "tests", "golden", "trusted_setup.txt"))
    bits = ts.direct_table_bits()
    stream = torch.cuda.current_stream(dev).cuda_stream
    to_dev = lambda b: torch.from_numpy(np.frombuffer(b, dtype=np.uint8).copy()).to(dev)

    def med(fn, reps=a.reps, warm=3):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize(dev)
        xs = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize(dev)
            xs.append((time.perf_counter() - t0) * 1e3)
        return statistics.median(xs)

    def xgmi_ms(total_bytes, G):   # ring all-gather / broadcast: every link carries (G - 1) / G of the total
        return 0.0 if G == 1 else total_bytes * (G - 1) / G / (XGMI_LINK_GBS * 1e9) * 1e3

    out = {"label": "PROJECTION from one-GPU measurements -- unmeasured on multi-GPU hardware", "device_engine_bits": bits,
           "xgmi_link_GBps_assumed": XGMI_LINK_GBS, "collectives_rccl_world_size_1": coll, "configs": {}}
    ag = lambda key: (coll.get(key, {}) or {}).get("world1_ms", 0.0) if isinstance(coll, dict) else 0.0

    # ---- configs[1]: commitments, weak scaling, 1024 blobs per rank -----------------------------------------------------------------------
    n = 1024
    d_b = to_dev(B.synthetic_batch(0, n)); d_o = torch.empty(48 * n, dtype=torch.uint8, device=dev); d_s = torch.zeros(n, dtype=torch.int32, device=dev)
    t_commit = med(lambda: K.blob_to_kzg_commitment_batch_device(d_o.data_ptr(), d_b.data_ptr(), n, ts, stream, d_s.data_ptr()))
    out["configs"]["configs[1] blob_to_kzg_commitment, 1024 blobs per rank (weak)"] = {
        "per_rank_ms": t_commit, "collectives_in_timed_region": "none (one broadcast of the %d-byte setup image at load: %.3f ms through RCCL at world size 1 + %.3f ms of xGMI at 8 ranks)"
                                                                % (coll.get("broadcast_setup_image", {}).get("bytes", 0) if isinstance(coll, dict) else 0, ag("broadcast_setup_image"),
                                                                   xgmi_ms(coll.get("broadcast_setup_image", {}).get("bytes", 0) if isinstance(coll, dict) else 0, 8)),
        "G": {str(G): {"value_ops_per_s": G * n / t_commit * 1e3, "efficiency": 1.0} for G in GS},
        "bound": "ALU per rank; shards independent -> linear by construction. What one box cannot show: eight ranks' table builds and host threads sharing one host"}

    # ---- configs[2]: blob proofs, 256 blobs in all, strong scaling ---------------------------------------------------------------------------
    nb = 256
    d_b2 = to_dev(B.synthetic_batch(7000, nb)); d_c2 = torch.empty(48 * nb, dtype=torch.uint8, device=dev); d_p2 = torch.empty(48 * nb, dtype=torch.uint8, device=dev)
    K.blob_to_kzg_commitment_batch_device(d_c2.data_ptr(), d_b2.data_ptr(), nb, ts, stream, d_s.data_ptr())
    torch.cuda.synchronize(dev)
    proof = {}
    for G in GS:
        m = nb // G
        proof[G] = med(lambda: K.compute_blob_kzg_proof_batch_device(d_p2.data_ptr(), d_b2.data_ptr(), d_c2.data_ptr(), m, ts, stream, d_s.data_ptr()), reps=2 * a.reps)
    out["configs"]["configs[2] compute_blob_kzg_proof, 256 blobs (strong)"] = {
        "G": {str(G): {"shard_blobs": nb // G, "rank_ms": proof[G], "value_proofs_per_s": nb / proof[G] * 1e3, "efficiency": proof[1] / (G * proof[G])} for G in GS},
        "bound": "LATENCY: a shard of 32-128 blobs is hash (host threads) -> quotient -> one small MSM -> fold -> finalize, chains that do not shrink with the shard"}

    # ---- configs[3]: batch verification, 4096 blobs in all, strong scaling -------------------------------------------------------------------
    nv = 4096
    h_bl = B.synthetic_batch(9000, nv)
    h_cm = b"".join(K.blob_to_kzg_commitment_batch(h_bl, ts)); h_pr = b"".join(K.compute_blob_kzg_proof_batch(h_bl, h_cm, ts))
    d_bl, d_cm, d_pr = to_dev(h_bl), to_dev(h_cm), to_dev(h_pr)
    whole = capi.VerifyShard.from_device(d_bl.data_ptr(), d_cm.data_ptr(), d_pr.data_ptr(), nv, ts)
    records_all = whole.records
    whole.free()
    ver = {}
    for G in GS:
        m = nv // G
        # every shard's partial sums (untimed) so that the finish sees what it would see
        partials = []
        for k in range(G):
            sh = capi.VerifyShard.from_device(d_bl.data_ptr() + k * m * B.BYTES_PER_BLOB, d_cm.data_ptr() + 48 * k * m, d_pr.data_ptr() + 48 * k * m, m, ts)
            partials.append(sh.partial(records_all, nv, k * m)); sh.free()
        held = []

        def begin():
            held.append(capi.VerifyShard.from_device(d_bl.data_ptr(), d_cm.data_ptr(), d_pr.data_ptr(), m, ts))
        t_begin = med(begin)
        sh0 = held[-1]
        t_partial = med(lambda: sh0.partial(records_all, nv, 0))
        for s_ in held:
            s_.free()
        t_finish = med(lambda: capi.verify_shards_finish(b"".join(partials), G, nv, ts))
        assert capi.verify_shards_finish(b"".join(partials), G, nv, ts) is True
        c_ms = (ag("allgather_records_4096") + xgmi_ms(160 * nv, G) + ag("allgather_partials") + xgmi_ms(329 * G, G)) if G > 1 else 0.0
        total = t_begin + t_partial + t_finish + c_ms
        ver[G] = {"shard_blobs": m, "begin_ms": t_begin, "partial_ms": t_partial, "finish_ms": t_finish, "collectives_ms": c_ms, "rank_ms": total,
                  "value_blobs_per_s": nv / total * 1e3}
    for G in GS:
        ver[G]["efficiency"] = ver[1]["rank_ms"] / (G * ver[G]["rank_ms"])
    out["configs"]["configs[3] verify_blob_kzg_proof_batch, 4096 blobs (strong)"] = {
        "G": {str(G): ver[G] for G in GS},
        "bound": "LATENCY: every shard pays the flat 3.2 ms challenge hash, the evaluation, the host's transcript hash over ALL 4096 records, the bucket MSM's chain and the pairing check; only the evaluation and the MSM's bucket lists shrink with the shard"}

    # ---- configs[4]: one 2^20-term MSM, 256 tiles, strong scaling -------------------------------------------------------------------------------
    tiles = 256
    d_t = to_dev(B.synthetic_batch(5000, tiles)); d_o48 = torch.empty(48, dtype=torch.uint8, device=dev)
    msm = {}
    for G in GS:
        m = tiles // G
        t_m = med(lambda: capi.g1_msm_tiled_device(d_o48.data_ptr(), d_t.data_ptr(), m * 4096, ts))
        pts = [bytes(d_o48.cpu().numpy().tobytes())] * G
        t_sum = med(lambda: capi.g1_sum_compressed(b"".join(pts)), warm=1)
        c_ms = (ag("allgather_point") + xgmi_ms(48 * G, G)) if G > 1 else 0.0
        msm[G] = {"shard_tiles": m, "msm_ms": t_m, "host_sum_ms": t_sum, "collectives_ms": c_ms, "rank_ms": t_m + t_sum + c_ms}
    for G in GS:
        msm[G]["value_terms_per_s"] = tiles * 4096 / msm[G]["rank_ms"] * 1e3
        msm[G]["efficiency"] = msm[1]["rank_ms"] / (G * msm[G]["rank_ms"])
    out["configs"]["configs[4] 2^20-term G1 MSM over the tiled setup (strong)"] = {
        "G": {str(G): msm[G] for G in GS},
        "bound": "ALU down to ~64 tiles per rank, then the launch set's flat part (parse, fold, finalize, the host's decompress-and-add of G points)"}

    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(out, open(a.out, "w"), indent=1)
    # ---- the table --------------------------------------------------------------------------------------------------------------------------
    print("| config (projection, unmeasured on hardware) | G = 1 | G = 2 | G = 4 | G = 8 | expected efficiency at 8 | bound |")
    print("|---|---|---|---|---|---|---|")
    for name, c in out["configs"].items():
        cells = []
        for G in GS:
            g = c["G"][str(G)]
            v = next(v for k, v in g.items() if k.startswith("value_"))
            ms = g.get("rank_ms")
            unit = next(k for k in g if k.startswith("value_"))[len("value_"):].replace("_per_s", "/s")
            sv = "%.3g M" % (v / 1e6) if v >= 1e6 else "%.3g k" % (v / 1e3)
            cells.append("%s %s%s" % (sv, unit, "" if ms is None else " (%.2f ms)" % ms))
        print("| %s | %s | %.2f | %s |" % (name, " | ".join(cells), c["G"]["8"]["efficiency"], c["bound"].split(":")[0]))
    ts.free()


if __name__ == "__main__":
    main()
