/* The reference's own integration tests (/root/reference/tests/lib_test.rs), restated as a plain C program against the
 * drop-in library: same inputs, same expectations, in the order of the Rust file. Exit code 0 = all assertions held.
 *   :19-87    constant polynomial 1, z = 1  ->  y = 1, proof = point at infinity, verify_kzg_proof accepts
 *   :89-167   polynomial x (coefficients [0, 1, 0, ...], big-endian), z = 2  ->  y = 2, proof == g1[0], commitment == g1[1]
 *   :169-260  verify_blob_kzg_proof_batch accepts the two blobs with their commitments and blob proofs
 *   :262-291  g1[0] compresses to 97f1d3a7...c6bb; the setup read from the text file round-trips through load_trusted_setup
 * Reference semantics (LWKZG_MODE unset): big-endian scalars are monomial coefficients. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lambdaworks_kzg_amd.h"

#define CHECK(cond)                                                        \
    do {                                                                   \
        if (getenv("MIRROR_TRACE")) fprintf(stderr, "line %d\n", __LINE__);   \
        if (!(cond)) {                                                     \
            fprintf(stderr, "line %d: %s failed (%s)\n", __LINE__, #cond, lwkzg_last_error()); \
            return 1;                                                      \
        }                                                                  \
    } while (0)

static int hexval(int c) { return c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1; }

/* the compressed bytes of the text file: line 1 = 4096, line 2 = 65, then one hex point per line */
static int read_setup_bytes(const char *path, uint8_t *g1, uint8_t *g2) {
    FILE *f = fopen(path, "r");
    if (!f) return 0;
    char line[512];
    if (!fgets(line, sizeof line, f) || atoi(line) != 4096) return 0;
    if (!fgets(line, sizeof line, f) || atoi(line) != 65) return 0;
    for (int i = 0; i < 4096 + 65; i++) {
        if (!fgets(line, sizeof line, f)) return 0;
        const int nb = i < 4096 ? 48 : 96;
        uint8_t *dst = i < 4096 ? g1 + 48 * i : g2 + 96 * (i - 4096);
        for (int k = 0; k < nb; k++) dst[k] = (uint8_t)(hexval(line[2 * k]) * 16 + hexval(line[2 * k + 1]));
    }
    fclose(f);
    return 1;
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    KZGSettings s, s2;
    FILE *fp = fopen(argv[1], "r");
    CHECK(fp != NULL);
    CHECK(load_trusted_setup_file(&s, fp) == C_KZG_OK);
    fclose(fp);

    static uint8_t g1[4096 * 48], g2[65 * 96];
    CHECK(read_setup_bytes(argv[1], g1, g2));

    Blob *blobs = calloc(2, sizeof(Blob));
    Bytes48 commitments[2], proofs[2];
    KZGProof proof;
    Bytes32 z, y;
    bool ok = false;

    /* lib_test.rs:19-87 */
    blobs[0].bytes[31] = 1; /* coefficient 0 = 1 */
    memset(z.bytes, 0, 32);
    z.bytes[31] = 1;
    CHECK(compute_kzg_proof(&proof, &y, &blobs[0], &z, &s) == C_KZG_OK);
    CHECK(memcmp(y.bytes, z.bytes, 32) == 0); /* y = 1 */
    CHECK(proof.bytes[0] == 0xc0);
    for (int i = 1; i < 48; i++) CHECK(proof.bytes[i] == 0);
    CHECK(blob_to_kzg_commitment((KZGCommitment *)&commitments[0], &blobs[0], &s) == C_KZG_OK);
    CHECK(memcmp(commitments[0].bytes, g1, 48) == 0); /* commitment to 1 is the generator */
    CHECK(verify_kzg_proof(&ok, &commitments[0], &z, &y, (Bytes48 *)&proof, &s) == C_KZG_OK && ok);

    /* lib_test.rs:89-167 */
    blobs[1].bytes[63] = 1; /* coefficient 1 = 1: p(x) = x */
    z.bytes[31] = 2;
    CHECK(compute_kzg_proof(&proof, &y, &blobs[1], &z, &s) == C_KZG_OK);
    CHECK(memcmp(y.bytes, z.bytes, 32) == 0);          /* y = 2 */
    CHECK(memcmp(proof.bytes, g1, 48) == 0);            /* quotient 1 -> g1[0] */
    CHECK(blob_to_kzg_commitment((KZGCommitment *)&commitments[1], &blobs[1], &s) == C_KZG_OK);
    CHECK(memcmp(commitments[1].bytes, g1 + 48, 48) == 0); /* commitment to x is g1[1] */
    CHECK(verify_kzg_proof(&ok, &commitments[1], &z, &y, (Bytes48 *)&proof, &s) == C_KZG_OK && ok);
    y.bytes[31] ^= 1;
    CHECK(verify_kzg_proof(&ok, &commitments[1], &z, &y, (Bytes48 *)&proof, &s) == C_KZG_OK && !ok);

    /* lib_test.rs:169-260 */
    for (int i = 0; i < 2; i++) {
        CHECK(compute_blob_kzg_proof((KZGProof *)&proofs[i], &blobs[i], &commitments[i], &s) == C_KZG_OK);
        CHECK(verify_blob_kzg_proof(&ok, &blobs[i], &commitments[i], &proofs[i], &s) == C_KZG_OK && ok);
    }
    CHECK(verify_blob_kzg_proof_batch(&ok, blobs, commitments, proofs, 2, &s) == C_KZG_OK && ok);
    CHECK(verify_blob_kzg_proof_batch(&ok, blobs, commitments, proofs, 0, &s) == C_KZG_OK && !ok); /* lib.rs:538-543 */
    {
        Bytes48 swapped[2];
        swapped[0] = proofs[1];
        swapped[1] = proofs[0];
        CHECK(verify_blob_kzg_proof_batch(&ok, blobs, commitments, swapped, 2, &s) == C_KZG_OK && !ok);
    }

    /* lib_test.rs:262-291 */
    static const uint8_t gen[48] = {0x97, 0xf1, 0xd3, 0xa7, 0x31, 0x97, 0xd7, 0x94, 0x26, 0x95, 0x63, 0x8c, 0x4f, 0xa9, 0xac, 0x0f,
                                    0xc3, 0x68, 0x8c, 0x4f, 0x97, 0x74, 0xb9, 0x05, 0xa1, 0x4e, 0x3a, 0x3f, 0x17, 0x1b, 0xac, 0x58,
                                    0x6c, 0x55, 0xe8, 0x3f, 0xf9, 0x7a, 0x1a, 0xef, 0xfb, 0x3a, 0xf0, 0x0a, 0xdb, 0x22, 0xc6, 0xbb};
    CHECK(memcmp(g1, gen, 48) == 0);
    CHECK(load_trusted_setup(&s2, g1, 4096, g2, 65) == C_KZG_OK);
    CHECK(memcmp(s.g1_values, s2.g1_values, 4096 * sizeof(g1_t)) == 0);
    CHECK(memcmp(s.g2_values, s2.g2_values, 65 * sizeof(g2_t)) == 0);
    CHECK(load_trusted_setup(&s2, g1, 4095, g2, 65) == C_KZG_BADARGS); /* lib.rs:716-718; s2 untouched */
    {
        KZGCommitment c2;
        CHECK(blob_to_kzg_commitment(&c2, &blobs[1], &s2) == C_KZG_OK);
        CHECK(memcmp(c2.bytes, commitments[1].bytes, 48) == 0);
    }
    CHECK(free_trusted_setup(&s2) == C_KZG_OK);
    CHECK(free_trusted_setup(&s) == C_KZG_OK);
    free(blobs);
    printf("lib_test mirror: all assertions held\n");
    return 0;
}
