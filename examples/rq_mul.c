/* examples/rq_mul.c — the boundary used from plain C99: one Rq product and one batched forward
 * NTT through libfhe_ntt.so, nothing but include/fhe_ntt.h.
 *
 *   gcc -std=c99 -O2 -o rq_mul examples/rq_mul.c -Lfhe-study_amd -lfhe_ntt -Wl,-rpath,$PWD/fhe-study_amd
 *   ./rq_mul            (needs an MI355X; prints the reference's KAT, arith/src/ring_nq.rs:674-682)
 */
#include <stdio.h>
#include <stdlib.h>

#include "../include/fhe_ntt.h"

#define CHECK(call)                                                              \
    do {                                                                         \
        int rc_ = (call);                                                        \
        if (rc_ != FHE_OK) {                                                     \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, fhe_last_error());     \
            return 1;                                                            \
        }                                                                        \
    } while (0)

int main(void) {
    const uint64_t q = 65537, n = 4;
    const fhe_ntt_plan *plan = NULL;
    uint64_t a[4] = {1, 2, 3, 4}, c[4], c_evals[4];
    size_t i;

    if (fhe_ntt_device_count() < 1) {
        fprintf(stderr, "no HIP device: the library has no CPU path\n");
        return 2;
    }
    CHECK(fhe_ntt_plan_get(q, n, &plan));
    /* c = a * a in Z_q[X]/(X^4+1); operands are coefficients (flags 0), evals of c kept */
    CHECK(fhe_rq_mul(plan, a, 0, a, 0, c, c_evals, NULL, NULL, 1));
    printf("[1,2,3,4]^2 mod (X^4+1, 65537) = [%llu, %llu, %llu, %llu]   (expected 65513 65517 65531 20)\n",
           (unsigned long long)c[0], (unsigned long long)c[1], (unsigned long long)c[2], (unsigned long long)c[3]);
    if (c[0] != 65513 || c[1] != 65517 || c[2] != 65531 || c[3] != 20) return 1;

    /* a batch: 1000 polynomials of 4096 coefficients mod q61, forward then inverse */
    {
        const uint64_t q61 = 2305843009211596801ull, n2 = 4096;
        const size_t batch = 1000;
        uint64_t *x = (uint64_t *)malloc(batch * n2 * 8), *y = (uint64_t *)malloc(batch * n2 * 8);
        uint64_t s = 88172645463325252ull;
        if (!x || !y) return 1;
        for (i = 0; i < batch * n2; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x[i] = s % q61; }
        CHECK(fhe_ntt_plan_get(q61, n2, &plan));
        CHECK(fhe_ntt_forward(plan, x, y, batch));
        CHECK(fhe_ntt_inverse(plan, y, y, batch));
        for (i = 0; i < batch * n2; i++)
            if (x[i] != y[i]) { fprintf(stderr, "round trip mismatch at %zu\n", i); return 1; }
        printf("round trip of %zu polynomials (n = %llu, q = 2^61 - 2^21 + 1): bit-exact\n", batch, (unsigned long long)n2);
        free(x); free(y);
    }
    CHECK(fhe_ntt_shutdown());
    return 0;
}
