/* The library's ONE shard rule from C (lwkzg_shard_range, include/lambdaworks_kzg_amd.h): what csrc/multi.hip cuts its batches by and
 * what lambdaworks_kzg_amd/dist.py calls. Prints "n parts k first count" for the cases on the command line (pairs "n parts"); the CPU
 * test compares with the closed form [ceil(k n / parts), ceil((k + 1) n / parts)) and with dist.shard_range. Needs no GPU. */
#include <stdio.h>
#include <stdlib.h>

#include "lambdaworks_kzg_amd.h"

int main(int argc, char **argv) {
    for (int a = 1; a + 1 < argc; a += 2) {
        const size_t n = (size_t)strtoull(argv[a], NULL, 10), parts = (size_t)strtoull(argv[a + 1], NULL, 10);
        size_t next = 0;
        for (size_t k = 0; k < parts; k++) {
            size_t first = 99, count = 99;
            if (lwkzg_shard_range(n, parts, k, &first, &count) != C_KZG_OK) return 2;
            if (first != next) return 3; /* contiguous, in order */
            next = first + count;
            printf("%zu %zu %zu %zu %zu\n", n, parts, k, first, count);
        }
        if (next != n) return 4; /* covers [0, n) exactly once */
    }
    size_t f, c;
    if (lwkzg_shard_range(8, 0, 0, &f, &c) != C_KZG_BADARGS || lwkzg_shard_range(8, 2, 2, &f, &c) != C_KZG_BADARGS ||
        lwkzg_shard_range(8, 2, 0, NULL, &c) != C_KZG_BADARGS)
        return 5;
    return 0;
}
