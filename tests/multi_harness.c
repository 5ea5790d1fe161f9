/* A C consumer of the node-level entry points (lwkzg_multi_*, include/lambdaworks_kzg_amd.h): what a caller of the reference's
 * C ABI (/root/reference/fuzz/base_fuzz.h:17-34, src/lib.rs:253-283) writes to spread its batches over the GPUs of a node, with
 * no Python and no launcher. argv: setup file, blobs file (n x 131072 bytes), comma-separated device ordinals (an ordinal may
 * repeat: several contexts on one GPU), mode (0 reference / 1 c-kzg). Prints hex for the pytest wrapper, which compares with
 * the single-device calls. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lambdaworks_kzg_amd.h"

static void hex(const char *tag, const uint8_t *b, size_t n) {
    printf("%s ", tag);
    for (size_t i = 0; i < n; i++) printf("%02x", b[i]);
    printf("\n");
}

#define CHECK(call, code)                                                         \
    do {                                                                          \
        C_KZG_RET rc_ = (call);                                                   \
        if (rc_ != C_KZG_OK) {                                                    \
            fprintf(stderr, "%s: %d (%s)\n", #call, (int)rc_, lwkzg_last_error()); \
            return code;                                                          \
        }                                                                         \
    } while (0)

int main(int argc, char **argv) {
    if (argc < 5) return 2;
    int devices[64];
    size_t nd = 0;
    for (char *tok = strtok(argv[3], ","); tok && nd < 64; tok = strtok(NULL, ",")) devices[nd++] = atoi(tok);
    const int mode = atoi(argv[4]);

    FILE *fb = fopen(argv[2], "rb");
    if (!fb) return 3;
    fseek(fb, 0, SEEK_END);
    const size_t n = (size_t)ftell(fb) / sizeof(Blob);
    fseek(fb, 0, SEEK_SET);
    Blob *blobs = malloc(n * sizeof(Blob));
    if (!blobs || fread(blobs, sizeof(Blob), n, fb) != n) return 4;
    fclose(fb);

    LwkzgMulti *m = NULL;
    FILE *fp = fopen(argv[1], "r");
    if (!fp) return 5;
    CHECK(lwkzg_multi_load_file(&m, fp, devices, nd), 6);
    fclose(fp);
    printf("devices %zu\n", lwkzg_multi_device_count(m));
    CHECK(lwkzg_multi_set_mode(m, mode), 7);

    KZGCommitment *c = malloc(n * sizeof *c);
    KZGProof *p = malloc(n * sizeof *p), *pz = malloc(n * sizeof *pz);
    Bytes32 *z = malloc(n * sizeof *z), *y = malloc(n * sizeof *y);
    size_t bad = 0;
    CHECK(lwkzg_multi_blob_to_kzg_commitment_batch(c, blobs, n, m, &bad), 8);
    hex("commitments", (const uint8_t *)c, 48 * n);
    CHECK(lwkzg_multi_compute_blob_kzg_proof_batch(p, blobs, c, n, m, &bad), 9);
    hex("blob_proofs", (const uint8_t *)p, 48 * n);
    for (size_t i = 0; i < n; i++) { /* z_i: a small canonical scalar in either byte order */
        memset(z[i].bytes, 0, 32);
        z[i].bytes[mode ? 0 : 31] = (uint8_t)(i + 2);
        z[i].bytes[mode ? 1 : 30] = (uint8_t)(i >> 8);
    }
    CHECK(lwkzg_multi_compute_kzg_proof_batch(pz, y, blobs, z, n, m, &bad), 10);
    hex("point_proofs", (const uint8_t *)pz, 48 * n);
    hex("ys", (const uint8_t *)y, 32 * n);
    bool ok = false;
    CHECK(lwkzg_multi_verify_blob_kzg_proof_batch(&ok, blobs, c, p, n, m), 11);
    printf("verify_batch %d\n", ok ? 1 : 0);
    if (n > 1) { /* proofs of blobs 0 and n - 1 swapped: they sit on different devices */
        KZGProof t = p[0];
        p[0] = p[n - 1];
        p[n - 1] = t;
        CHECK(lwkzg_multi_verify_blob_kzg_proof_batch(&ok, blobs, c, p, n, m), 12);
        printf("verify_batch_swapped %d\n", ok ? 1 : 0);
    }
    /* the per-device settings are ordinary KZGSettings: the reference's own symbol on the LAST device's copy */
    KZGCommitment c1;
    CHECK(blob_to_kzg_commitment(&c1, &blobs[0], lwkzg_multi_settings(m, lwkzg_multi_device_count(m) - 1)), 13);
    hex("commitment0_on_last_device", c1.bytes, 48);
    if (mode == 0) { /* BASELINE configs[4] in miniature: the blobs read as one long MSM over the tiled setup */
        uint8_t out[48];
        CHECK(lwkzg_multi_g1_msm_tiled(out, (const uint8_t *)blobs, n * 4096, m), 14);
        hex("tiled_msm", out, 48);
    }
    /* a bad blob: the lowest offending index of the WHOLE batch comes back, whichever device saw it */
    if (n > 2 && mode == 1) {
        memset(blobs[n - 2].bytes, 0xff, 32);
        C_KZG_RET rc = lwkzg_multi_blob_to_kzg_commitment_batch(c, blobs, n, m, &bad);
        printf("bad_blob_rc %d first_bad %zu\n", (int)rc, bad);
    }
    lwkzg_multi_free(m);
    free(blobs); free(c); free(p); free(pz); free(z); free(y);
    return 0;
}
