/* A C consumer of the drop-in library, written the way the reference's fuzz harnesses use the API
 * (/root/reference/fuzz/base_fuzz.h:17-34, fuzz/blob_to_kzg_commitment/fuzz.c): load the setup from a FILE*,
 * call the c-kzg-4844 symbols, free. Prints hex results for the pytest wrapper to compare with the oracle. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lambdaworks_kzg_amd.h"

static void hex(const char *tag, const uint8_t *b, size_t n) {
    printf("%s ", tag);
    for (size_t i = 0; i < n; i++) printf("%02x", b[i]);
    printf("\n");
}

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    KZGSettings s;
    FILE *fp = fopen(argv[1], "r");
    if (!fp) return 3;
    C_KZG_RET rc = load_trusted_setup_file(&s, fp);
    fclose(fp);
    if (rc != C_KZG_OK) {
        fprintf(stderr, "load_trusted_setup_file: %d (%s)\n", rc, lwkzg_last_error());
        return 4;
    }
    Blob *blob = malloc(sizeof(Blob));
    FILE *fb = fopen(argv[2], "rb");
    if (!fb || fread(blob->bytes, 1, sizeof blob->bytes, fb) != sizeof blob->bytes) return 5;
    fclose(fb);

    KZGCommitment c;
    KZGProof p, p2;
    Bytes32 z, y;
    bool ok = false;
    if (blob_to_kzg_commitment(&c, blob, &s) != C_KZG_OK) return 6;
    hex("commitment", c.bytes, 48);
    if (compute_blob_kzg_proof(&p, blob, &c, &s) != C_KZG_OK) return 7;
    hex("blob_proof", p.bytes, 48);
    if (verify_blob_kzg_proof(&ok, blob, &c, &p, &s) != C_KZG_OK) return 8;
    printf("verify_blob %d\n", ok ? 1 : 0);
    memset(z.bytes, 0, 32);
    z.bytes[31] = 2;
    if (compute_kzg_proof(&p2, &y, blob, &z, &s) != C_KZG_OK) return 9;
    hex("proof", p2.bytes, 48);
    hex("y", y.bytes, 32);
    if (verify_kzg_proof(&ok, &c, &z, &y, &p2, &s) != C_KZG_OK) return 10;
    printf("verify %d\n", ok ? 1 : 0);
    y.bytes[31] ^= 1;
    if (verify_kzg_proof(&ok, &c, &z, &y, &p2, &s) != C_KZG_OK) return 11;
    printf("verify_wrong_y %d\n", ok ? 1 : 0);
    /* g1_values is the reference's blst_p1 array: the x limb of the generator, most significant first */
    printf("g1_0_x_limb0 %016llx\n", (unsigned long long)s.g1_values[0].x.l[0]);
    free(blob);
    if (free_trusted_setup(&s) != C_KZG_OK) return 12;
    return 0;
}
