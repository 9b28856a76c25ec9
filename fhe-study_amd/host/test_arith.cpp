// test_arith.cpp — the reference's own NTT-path tests, restated in C++ over arith.hpp
// (and therefore over the C ABI and the HIP kernels).  Needs a GPU; run by
// tests/test_host_cpp.py under `-m gpu`.  Exit code 0 = all passed.
#include <cstdio>
#include <random>

#include "arith.hpp"

using namespace arith;

static int failures = 0;
#define EXPECT(cond)                                                        \
    do {                                                                    \
        if (!(cond)) { printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); failures++; } \
    } while (0)

// arith/src/ntt.rs:194-215
static void test_ntt() {
    RingParam param{(1ull << 16) + 1, 4};
    Rq a = Rq::from_vec_u64(param, {1, 2, 3, 4});
    Rq a_ntt = NTT::ntt(a);
    Rq a_intt = NTT::intt(a_ntt);
    EXPECT(a == a_intt);
    EXPECT((a_ntt.coeffs_v == std::vector<uint64_t>{7489, 56514, 17185, 49890}));
}

// arith/src/ntt.rs:217-234 (thread_rng there; a fixed seed here)
static void test_ntt_loop() {
    RingParam param{(1ull << 16) + 1, 512};
    std::mt19937_64 rng(0xF4E5);
    for (int it = 0; it < 1000; it++) {
        std::vector<uint64_t> c(param.n);
        for (auto &x : c) x = rng() % param.q;
        Rq a(param, c);
        EXPECT(a == NTT::intt(NTT::ntt(a)));
    }
}

// arith/src/ring_nq.rs:667-704
static void test_mul_opt(const RingParam &param, std::vector<uint64_t> a, std::vector<uint64_t> b,
                         std::vector<uint64_t> expected_c) {
    Rq ra = Rq::from_vec_u64(param, a), rb = Rq::from_vec_u64(param, b);
    Rq want = Rq::from_vec_u64(param, expected_c);
    Rq c = ra.mul(rb);  // mul_mut
    EXPECT(c == want);
    EXPECT(ra.evals_v.has_value() && rb.evals_v.has_value() && c.evals_v.has_value());
    EXPECT((ra * rb) == want);  // now through the cached evals
}
static void test_mul() {
    RingParam param{(1ull << 16) + 1, 4};
    test_mul_opt(param, {1, 2, 3, 4}, {1, 2, 3, 4}, {65513, 65517, 65531, 20});
    test_mul_opt(param, {0, 0, 0, 2}, {0, 0, 0, 2}, {0, 0, 65533, 0});
}

// arith/src/ring_nq.rs:626-650 (fold + mod q of from_vec_u64)
static void test_from_vec_fold() {
    Rq p = Rq::from_vec_u64(RingParam{7, 4}, {0, 1, 2, 3, 4, 5});
    EXPECT((p.coeffs_v == std::vector<uint64_t>{3, 3, 2, 3}));
}

static void test_panics() {
    bool threw = false;
    try { NTT::ntt(Rq(RingParam{65537, 3}, {1, 2, 3})); } catch (const Panic &e) { threw = e.code == FHE_E_BAD_N; }
    EXPECT(threw);  // assert!(n.is_power_of_two()), ntt.rs:116
    threw = false;
    try { NTT::ntt(Rq(RingParam{65537, 4}, {1, 2, 3, 4}) ); Rq x(RingParam{65537, 4}, {1, 2, 3, 4}); Rq y(RingParam{65537, 8}, std::vector<uint64_t>(8, 1)); (void)(x * y); }
    catch (const Panic &e) { threw = e.code == FHE_E_PARAM_MISMATCH; }
    EXPECT(threw);  // assert_eq!(lhs.param, rhs.param), ring_nq.rs:587
}

// q = 2^61 - 2^21 + 1, n = 1024 (BASELINE.json configs[0]): product == schoolbook
static void test_q61_against_schoolbook() {
    const uint64_t q = 2305843009211596801ull;
    RingParam param{q, 1024};
    std::mt19937_64 rng(7);
    std::vector<uint64_t> a(1024), b(1024), want(1024, 0);
    for (auto &x : a) x = rng() % q;
    for (auto &x : b) x = rng() % q;
    for (size_t i = 0; i < 1024; i++)
        for (size_t j = 0; j < 1024; j++) {
            uint64_t p = (uint64_t)(((unsigned __int128)a[i] * b[j]) % q);
            size_t k = i + j;
            if (k < 1024) { want[k] += p; if (want[k] >= q) want[k] -= q; }
            else { uint64_t &w = want[k - 1024]; w = w >= p ? w - p : q + w - p; }
        }
    Rq c = Rq(param, a) * Rq(param, b);
    EXPECT(c.coeffs_v == want);
}

int main() {
    if (fhe_ntt_device_count() < 1) { printf("no HIP device\n"); return 2; }
    test_ntt();
    test_ntt_loop();
    test_mul();
    test_from_vec_fold();
    test_panics();
    test_q61_against_schoolbook();
    printf(failures ? "%d FAILURES\n" : "all host C++ tests passed%.0d\n", failures);
    return failures ? 1 : 0;
}
