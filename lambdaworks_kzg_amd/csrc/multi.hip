// multi.hip -- ONE process, SEVERAL GPUs: the node-level form of the batch entry points, for the reference's own kind of
// caller. lambdaworks_kzg is driven by plain C calls (/root/reference/fuzz/base_fuzz.h:17-34, src/lib.rs:253-283); a C / Rust
// / Go consumer has no torch.distributed, so the sharding of SURVEY section 8(e) must exist below the C ABI as well
// (VERDICT r03): blob k of a batch of B goes to device floor(k G / B) -- contiguous shards --, every device holds the whole
// setup and its own MSM table, there is NO reduction and no data-path collective. The one exchange is the delivery of the
// prepared setup image (10.3 MB: blst arrays, 9 MB fixed-base table, twiddles, affine points) from the device that parsed and
// validated the file to the others, device to device (hipMemcpyPeer: xGMI on an MI355X node); each device then builds its
// own direct table, all of them at the same time. One host thread per device drives that device's shard through the
// single-device entry points, so everything they do (slicing, host hashing, coalescing, error mapping) is what a
// one-GPU caller gets. Batch verification keeps the reference's form -- ONE Fiat-Shamir scalar over all blobs, ONE random
// linear combination, ONE pairing check (src/lib.rs:639-692) -- through the lwkzg_verify_shard_* steps, with the gather
// of the 160-byte records and the 328-byte partial sums being plain host memory here.
// (lambdaworks_kzg_amd/dist.py is the same scheme as one process per GPU over RCCL, for Python drivers such as bench.py.)
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "engine.h"

namespace lwk {
extern thread_local int tl_device_override;  // engine.hip: the device the next context of THIS thread is created on (-1: the default)
namespace {
// restored on every way out of the scope (an exception included): a thread that kept the override would create its later, unrelated
// contexts on the wrong GPU
struct DeviceOverride {
    explicit DeviceOverride(int dev) { tl_device_override = dev; }
    ~DeviceOverride() { tl_device_override = -1; }
    DeviceOverride(const DeviceOverride &) = delete;
    DeviceOverride &operator=(const DeviceOverride &) = delete;
};
}  // namespace
}
using namespace lwk;

struct LwkzgMulti {
    std::vector<KZGSettings> s;   // one loaded setup per entry of `dev`
    std::vector<int> dev;
};

namespace {

// THE shard rule of this library (lwkzg_shard_range, which dist.py calls too): item i of n belongs to part floor(i parts / n), i.e.
// part k owns [ceil(k n / parts), ceil((k + 1) n / parts)) -- contiguous, covering [0, n) exactly once, sizes differing by at most one
inline void shard_range(size_t n, size_t k, size_t parts, size_t &lo, size_t &hi) {
    lo = (n * k + parts - 1) / parts;
    hi = (n * (k + 1) + parts - 1) / parts;
}

// f(k) on one host thread per device; the first failing shard (lowest k) decides the return code and its thread's error text
// becomes the caller's (set_error is thread-local).
template <class F>
C_KZG_RET on_every_device(size_t parts, F f) {
    std::vector<int> rc(parts, C_KZG_OK);
    std::vector<std::string> err(parts);
    auto body = [&](size_t k) {
        try {
            rc[k] = f(k);
        } catch (const std::bad_alloc &) {
            set_error("out of host memory");
            rc[k] = C_KZG_MALLOC;
        } catch (...) {
            set_error("unexpected exception in a device thread");
            rc[k] = C_KZG_ERROR;
        }
        if (rc[k] != C_KZG_OK) err[k] = get_error();
    };
    {
        std::vector<SideTask> th(parts > 0 ? parts - 1 : 0);  // (SideTask runs inline when no thread can be had)
        for (size_t k = 1; k < parts; k++) th[k - 1].start([&body, k] { body(k); });
        if (parts) body(0);
    }
    for (size_t k = 0; k < parts; k++)
        if (rc[k] != C_KZG_OK) {
            set_error("device shard %zu: %s", k, err[k].c_str());
            return (C_KZG_RET)rc[k];
        }
    return C_KZG_OK;
}

// nothing may unwind across the C ABI: the entry points size host vectors by the number of devices and of blobs
template <class F>
C_KZG_RET guarded_multi(const char *what, F &&f) {
    try {
        return f();
    } catch (const std::bad_alloc &) {
        set_error("%s: out of host memory", what);
        return C_KZG_MALLOC;
    } catch (...) {
        set_error("%s: unexpected exception", what);
        return C_KZG_ERROR;
    }
}

// the shards of a verification are released on every way out of it
struct ShardsGuard {
    std::vector<LwkzgVerifyShard *> &v;
    ~ShardsGuard() {
        for (LwkzgVerifyShard *sh : v) lwkzg_verify_shard_free(sh);
    }
};

bool devices_ok(const int *devices, size_t n) {
    if (!devices || n == 0 || n > 64) {
        set_error("lwkzg_multi: need 1 .. 64 device ordinals");
        return false;
    }
    const int visible = lwkzg_device_count();
    for (size_t k = 0; k < n; k++)
        if (devices[k] < 0 || devices[k] >= visible) {
            set_error("lwkzg_multi: device %d of %d visible", devices[k], visible);
            return false;
        }
    return true;
}

// m->s[0] is loaded on m->dev[0]: deliver its image to every other entry and import it there
C_KZG_RET replicate(LwkzgMulti *m) {
    const size_t n = m->dev.size();
    if (n == 1) return C_KZG_OK;
    const size_t bytes = lwkzg_setup_image_bytes();
    void *img0 = nullptr;
    LWK_HIP(hipSetDevice(m->dev[0]));
    LWK_HIP(hipMalloc(&img0, bytes));
    C_KZG_RET rc = lwkzg_setup_export_device(&m->s[0], img0, nullptr);
    if (rc == C_KZG_OK)
        rc = on_every_device(n, [&](size_t k) -> int {
            if (k == 0) return C_KZG_OK;
            void *img = nullptr;
            LWK_HIP(hipSetDevice(m->dev[k]));
            LWK_HIP(hipMalloc(&img, bytes));
            hipError_t e = m->dev[k] == m->dev[0] ? hipMemcpy(img, img0, bytes, hipMemcpyDeviceToDevice)
                                                  : hipMemcpyPeer(img, m->dev[k], img0, m->dev[0], bytes);  // xGMI between two GPUs of a node
            int r = C_KZG_OK;
            if (e != hipSuccess) {
                set_error("setup image to device %d: %s", m->dev[k], hipGetErrorString(e));
                r = C_KZG_ERROR;
            } else {
                DeviceOverride on(m->dev[k]);
                r = lwkzg_setup_import_device(&m->s[k], img);  // (chooses and builds this device's MSM engine like a load does)
            }
            hipFree(img);
            return r;
        });
    hipSetDevice(m->dev[0]);
    hipFree(img0);
    return rc;
}

template <class Load>
C_KZG_RET multi_new(LwkzgMulti **out, const int *devices, size_t n_devices, Load load_first) {
    if (!out) return C_KZG_BADARGS;
    *out = nullptr;
    if (!devices_ok(devices, n_devices)) return C_KZG_BADARGS;
    LwkzgMulti *m = new (std::nothrow) LwkzgMulti;
    if (!m) return C_KZG_MALLOC;
    C_KZG_RET rc = C_KZG_OK;
    try {
        m->dev.assign(devices, devices + n_devices);
        m->s.assign(n_devices, KZGSettings{nullptr, nullptr, nullptr});
        {
            DeviceOverride on(m->dev[0]);
            rc = load_first(&m->s[0]);
        }
        if (rc == C_KZG_OK) rc = replicate(m);
    } catch (const std::bad_alloc &) {
        rc = C_KZG_MALLOC;
    }
    if (rc != C_KZG_OK) {
        const std::string keep = get_error();
        lwkzg_multi_free(m);
        set_error("%s", keep.c_str());
        return rc;
    }
    *out = m;
    return C_KZG_OK;
}

// shards of a host-pointer batch through a single-device batch entry point; first_bad = the lowest offending index of the whole batch
template <class Call>
C_KZG_RET sharded_batch(const LwkzgMulti *m, size_t n, size_t *first_bad, Call call) {
    if (!m) return C_KZG_BADARGS;
    if (first_bad) *first_bad = (size_t)-1;
    return guarded_multi("lwkzg_multi batch", [&]() -> C_KZG_RET {
        const size_t parts = m->s.size();
        std::vector<size_t> bad(parts, (size_t)-1);
        C_KZG_RET rc = on_every_device(parts, [&](size_t k) -> int {
            size_t lo, hi;
            shard_range(n, k, parts, lo, hi);
            if (hi == lo) return C_KZG_OK;
            size_t fb = (size_t)-1;
            const int r = call(&m->s[k], lo, hi - lo, &fb);
            if (r != C_KZG_OK && fb != (size_t)-1) bad[k] = lo + fb;
            return r;
        });
        if (first_bad)
            for (size_t k = 0; k < parts && *first_bad == (size_t)-1; k++) *first_bad = bad[k];
        return rc;
    });
}

}  // namespace

extern "C" {

C_KZG_RET lwkzg_shard_range(size_t n_items, size_t parts, size_t k, size_t *first, size_t *count) {
    if (!first || !count || parts == 0 || k >= parts) return C_KZG_BADARGS;
    size_t lo, hi;
    shard_range(n_items, k, parts, lo, hi);
    *first = lo;
    *count = hi - lo;
    return C_KZG_OK;
}

C_KZG_RET lwkzg_multi_load(LwkzgMulti **out, const uint8_t *g1_bytes, size_t n1, const uint8_t *g2_bytes, size_t n2, const int *devices,
                           size_t n_devices) {
    return multi_new(out, devices, n_devices, [&](KZGSettings *s) { return load_trusted_setup(s, g1_bytes, n1, g2_bytes, n2); });
}

C_KZG_RET lwkzg_multi_load_file(LwkzgMulti **out, FILE *in, const int *devices, size_t n_devices) {
    return multi_new(out, devices, n_devices, [&](KZGSettings *s) { return load_trusted_setup_file(s, in); });
}

void lwkzg_multi_free(LwkzgMulti *m) {
    if (!m) return;
    for (KZGSettings &s : m->s)
        if (s.fs || s.g1_values || s.g2_values) free_trusted_setup(&s);
    delete m;
}

size_t lwkzg_multi_device_count(const LwkzgMulti *m) { return m ? m->dev.size() : 0; }

int lwkzg_multi_device(const LwkzgMulti *m, size_t k) { return m && k < m->dev.size() ? m->dev[k] : -1; }

const KZGSettings *lwkzg_multi_settings(const LwkzgMulti *m, size_t k) { return m && k < m->s.size() ? &m->s[k] : nullptr; }

C_KZG_RET lwkzg_multi_set_mode(const LwkzgMulti *m, int mode) {
    if (!m || mode < -1 || mode > LWKZG_MODE_CKZG) return C_KZG_BADARGS;
    for (const KZGSettings &s : m->s)
        if (lwkzg_settings_set_mode(&s, mode) < 0) return C_KZG_ERROR;
    return C_KZG_OK;
}

C_KZG_RET lwkzg_multi_enable_direct_table(const LwkzgMulti *m, int window_bits) {
    if (!m) return C_KZG_BADARGS;
    return on_every_device(m->s.size(), [&](size_t k) -> int { return lwkzg_enable_direct_table(&m->s[k], window_bits); });
}

C_KZG_RET lwkzg_multi_blob_to_kzg_commitment_batch(KZGCommitment *out, const Blob *blobs, size_t n, const LwkzgMulti *m, size_t *first_bad) {
    if ((!out || !blobs) && n) return C_KZG_BADARGS;
    return sharded_batch(m, n, first_bad, [&](const KZGSettings *s, size_t lo, size_t cnt, size_t *fb) {
        return lwkzg_blob_to_kzg_commitment_batch(out + lo, blobs + lo, cnt, s, fb);
    });
}

C_KZG_RET lwkzg_multi_compute_blob_kzg_proof_batch(KZGProof *out, const Blob *blobs, const Bytes48 *commitments, size_t n, const LwkzgMulti *m,
                                                   size_t *first_bad) {
    if ((!out || !blobs || !commitments) && n) return C_KZG_BADARGS;
    return sharded_batch(m, n, first_bad, [&](const KZGSettings *s, size_t lo, size_t cnt, size_t *fb) {
        return lwkzg_compute_blob_kzg_proof_batch(out + lo, blobs + lo, commitments + lo, cnt, s, fb);
    });
}

C_KZG_RET lwkzg_multi_compute_kzg_proof_batch(KZGProof *proofs_out, Bytes32 *ys_out, const Blob *blobs, const Bytes32 *zs, size_t n,
                                              const LwkzgMulti *m, size_t *first_bad) {
    if ((!proofs_out || !ys_out || !blobs || !zs) && n) return C_KZG_BADARGS;
    return sharded_batch(m, n, first_bad, [&](const KZGSettings *s, size_t lo, size_t cnt, size_t *fb) {
        return lwkzg_compute_kzg_proof_batch(proofs_out + lo, ys_out + lo, blobs + lo, zs + lo, cnt, s, fb);
    });
}

// ---- the same calls for shards that are ALREADY in HBM: device k's shard is n_per_device[k] blobs behind blobs_dev[k] (pointers on
// device k), results in place behind out48_dev[k]. A node-level caller that produces or receives its blobs on the GPUs never touches
// pageable host memory; nothing crosses PCIe but the verdicts. first_bad counts through the shards in device order.
}  // extern "C" (a template cannot have C linkage)
namespace {
template <class Call>
C_KZG_RET sharded_device_batch(const LwkzgMulti *m, const size_t *n_per_device, size_t *first_bad, Call call) {
    if (!m || !n_per_device) return C_KZG_BADARGS;
    if (first_bad) *first_bad = (size_t)-1;
    return guarded_multi("lwkzg_multi device batch", [&]() -> C_KZG_RET {
        const size_t parts = m->s.size();
        std::vector<size_t> bad(parts, (size_t)-1);
        C_KZG_RET rc = on_every_device(parts, [&](size_t k) -> int {
            const size_t cnt = n_per_device[k];
            if (!cnt) return C_KZG_OK;
            std::vector<int32_t> h(cnt);   // (before the device allocation: a bad_alloc here must not leak it)
            LWK_HIP(hipSetDevice(m->dev[k]));
            int32_t *d_status = nullptr;
            LWK_HIP(hipMalloc((void **)&d_status, cnt * sizeof(int32_t)));
            int r = call(k, cnt, d_status);
            if (r == C_KZG_OK && (hipDeviceSynchronize() != hipSuccess || hipMemcpy(h.data(), d_status, cnt * sizeof(int32_t), hipMemcpyDeviceToHost) != hipSuccess)) {
                set_error("device %d: %s", m->dev[k], hipGetErrorString(hipGetLastError()));
                r = C_KZG_ERROR;
            }
            hipFree(d_status);
            if (r != C_KZG_OK) return r;
            for (size_t i = 0; i < cnt; i++)
                if (h[i] != 0) {
                    bad[k] = i;
                    set_error("blob %zu of device %d's shard rejected (status %d)", i, m->dev[k], h[i]);
                    return h[i];   // (the status words are C_KZG_RET values: ERROR in reference mode, BADARGS in c-kzg mode)
                }
            return C_KZG_OK;
        });
        if (first_bad) {
            size_t base = 0;
            for (size_t k = 0; k < parts && *first_bad == (size_t)-1; k++) {
                if (bad[k] != (size_t)-1) *first_bad = base + bad[k];
                base += n_per_device[k];
            }
        }
        return rc;
    });
}
}  // namespace
extern "C" {

C_KZG_RET lwkzg_multi_blob_to_kzg_commitment_batch_device(void *const *out48_dev, const void *const *blobs_dev, const size_t *n_per_device,
                                                          const LwkzgMulti *m, size_t *first_bad) {
    if (!out48_dev || !blobs_dev) return C_KZG_BADARGS;
    return sharded_device_batch(m, n_per_device, first_bad, [&](size_t k, size_t cnt, int32_t *d_status) -> int {
        return lwkzg_blob_to_kzg_commitment_batch_device(out48_dev[k], blobs_dev[k], cnt, &m->s[k], nullptr, d_status);
    });
}

C_KZG_RET lwkzg_multi_compute_blob_kzg_proof_batch_device(void *const *out48_dev, const void *const *blobs_dev, const void *const *commitments48_dev,
                                                          const size_t *n_per_device, const LwkzgMulti *m, size_t *first_bad) {
    if (!out48_dev || !blobs_dev || !commitments48_dev) return C_KZG_BADARGS;
    return sharded_device_batch(m, n_per_device, first_bad, [&](size_t k, size_t cnt, int32_t *d_status) -> int {
        return lwkzg_compute_blob_kzg_proof_batch_device(out48_dev[k], blobs_dev[k], commitments48_dev[k], cnt, &m->s[k], nullptr, d_status);
    });
}

// ONE batch (one r, one linear combination, one pairing check) whose shards live on the devices: the records and the partial sums are
// all that reaches the host
C_KZG_RET lwkzg_multi_verify_blob_kzg_proof_batch_device(bool *ok, const void *const *blobs_dev, const void *const *commitments48_dev,
                                                         const void *const *proofs48_dev, const size_t *n_per_device, const LwkzgMulti *m) {
    if (!ok) return C_KZG_BADARGS;
    *ok = false;
    if (!m || !blobs_dev || !commitments48_dev || !proofs48_dev || !n_per_device) return C_KZG_BADARGS;
    return guarded_multi("lwkzg_multi_verify_blob_kzg_proof_batch_device", [&]() -> C_KZG_RET {
        const size_t parts = m->s.size();
        std::vector<LwkzgVerifyShard *> shard(parts, nullptr);
        ShardsGuard free_shards{shard};
        std::vector<size_t> first(parts + 1, 0);
        for (size_t k = 0; k < parts; k++) first[k + 1] = first[k] + n_per_device[k];
        const size_t n = first[parts];
        if (n == 0) return verify_blob_kzg_proof_batch(ok, nullptr, nullptr, nullptr, 0, &m->s[0]);   // (the empty batch's mode-dependent verdict)
        std::vector<uint8_t> records(n * LWKZG_VERIFY_RECORD_BYTES + 1), partials(parts * LWKZG_VERIFY_PARTIAL_BYTES);
        C_KZG_RET rc = on_every_device(parts, [&](size_t k) -> int {
            return lwkzg_verify_shard_begin_device(&shard[k], records.data() + first[k] * LWKZG_VERIFY_RECORD_BYTES, blobs_dev[k], commitments48_dev[k],
                                                   proofs48_dev[k], n_per_device[k], &m->s[k], nullptr);
        });
        if (rc == C_KZG_OK)
            rc = on_every_device(parts, [&](size_t k) -> int {
                return lwkzg_verify_shard_partial(partials.data() + k * LWKZG_VERIFY_PARTIAL_BYTES, shard[k], records.data(), n, first[k]);
            });
        if (rc != C_KZG_OK) return rc;
        return lwkzg_verify_shards_finish(ok, partials.data(), parts, n, &m->s[0]);
    });
}

// verify_blob_kzg_proof_batch (src/lib.rs:525-692) over the devices: per-blob work sharded, ONE r, ONE linear combination, ONE pairing check
C_KZG_RET lwkzg_multi_verify_blob_kzg_proof_batch(bool *ok, const Blob *blobs, const Bytes48 *commitments, const Bytes48 *proofs, size_t n,
                                                  const LwkzgMulti *m) {
    if (!ok) return C_KZG_BADARGS;
    *ok = false;
    if (!m || ((!blobs || !commitments || !proofs) && n)) return C_KZG_BADARGS;
    return guarded_multi("lwkzg_multi_verify_blob_kzg_proof_batch", [&]() -> C_KZG_RET {
        const size_t parts = m->s.size();
        std::vector<LwkzgVerifyShard *> shard(parts, nullptr);
        ShardsGuard free_shards{shard};
        std::vector<uint8_t> records(n * LWKZG_VERIFY_RECORD_BYTES + 1), partials(parts * LWKZG_VERIFY_PARTIAL_BYTES);
        C_KZG_RET rc = on_every_device(parts, [&](size_t k) -> int {
            size_t lo, hi;
            shard_range(n, k, parts, lo, hi);
            return lwkzg_verify_shard_begin(&shard[k], records.data() + lo * LWKZG_VERIFY_RECORD_BYTES, blobs + lo, commitments + lo, proofs + lo, hi - lo,
                                            &m->s[k]);
        });
        if (rc == C_KZG_OK)   // (the "all-gather" of the records: they already sit in one host array)
            rc = on_every_device(parts, [&](size_t k) -> int {
                size_t lo, hi;
                shard_range(n, k, parts, lo, hi);
                return lwkzg_verify_shard_partial(partials.data() + k * LWKZG_VERIFY_PARTIAL_BYTES, shard[k], records.data(), n, lo);
            });
        if (rc != C_KZG_OK) return rc;
        return lwkzg_verify_shards_finish(ok, partials.data(), parts, n, &m->s[0]);
    });
}

// BASELINE configs[4]: out = sum_k scalars[k] * g1[k mod 4096] over n_terms (a positive multiple of 4096) host-resident big-endian
// scalars: whole tiles per device, one 48-byte partial sum back from each, added on the host.
C_KZG_RET lwkzg_multi_g1_msm_tiled(uint8_t out48[48], const uint8_t *scalars_be, size_t n_terms, const LwkzgMulti *m) {
    if (!m || !out48 || !scalars_be || n_terms == 0 || n_terms % 4096) return C_KZG_BADARGS;
    return guarded_multi("lwkzg_multi_g1_msm_tiled", [&]() -> C_KZG_RET {
    const size_t parts = m->s.size(), tiles = n_terms / 4096;
    std::vector<uint8_t> partial(parts * 48, 0);
    std::vector<int> have(parts, 0);
    C_KZG_RET rc = on_every_device(parts, [&](size_t k) -> int {
        size_t lo, hi;
        shard_range(tiles, k, parts, lo, hi);
        if (hi == lo) return C_KZG_OK;
        const size_t bytes = (hi - lo) * 4096 * 32;
        uint8_t *d_sc = nullptr, *d_out = nullptr;
        LWK_HIP(hipSetDevice(m->dev[k]));
        LWK_HIP(hipMalloc((void **)&d_sc, bytes + 48));
        d_out = d_sc + bytes;
        int r = C_KZG_ERROR;
        if (hipMemcpy(d_sc, scalars_be + lo * 4096 * 32, bytes, hipMemcpyHostToDevice) == hipSuccess) {
            r = lwkzg_g1_msm_tiled_device(d_out, d_sc, (hi - lo) * 4096, &m->s[k], nullptr);
            if (r == C_KZG_OK && (hipDeviceSynchronize() != hipSuccess || hipMemcpy(&partial[k * 48], d_out, 48, hipMemcpyDeviceToHost) != hipSuccess)) {
                set_error("tiled MSM on device %d: %s", m->dev[k], hipGetErrorString(hipGetLastError()));
                r = C_KZG_ERROR;
            }
        } else {
            set_error("scalar upload to device %d failed", m->dev[k]);
        }
        hipFree(d_sc);
        have[k] = r == C_KZG_OK;
        return r;
    });
    if (rc != C_KZG_OK) return rc;
    std::vector<uint8_t> pts;
    for (size_t k = 0; k < parts; k++)
        if (have[k]) pts.insert(pts.end(), partial.begin() + k * 48, partial.begin() + k * 48 + 48);
    return lwkzg_g1_sum_compressed(out48, pts.data(), pts.size() / 48);
    });
}

}  // extern "C"
