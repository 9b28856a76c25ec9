// test_bfv.cpp — the reference's BFV multiplication tests restated in C++ over bfv.hpp:
//   bfv/src/lib.rs:504-554  test_tensor      (decrypt of the 3-term tensor = m1*m2)
//   bfv/src/lib.rs:556-601  test_mul_relin   (decrypt(RLWE::mul(c1, c2)) = m1*m2 mod (t, X^n+1))
// Key generation, encryption and decryption below are TEST scaffolding restating
// bfv/src/lib.rs:118-181,202-225 (scheme logic, out of scope of the library); every ring product
// in them goes through the library too.  Needs a GPU; run by tests/test_host_cpp.py under `-m gpu`.
#include <cstdio>
#include <random>

#include "bfv.hpp"

using namespace bfv;

static int failures = 0;
#define EXPECT(cond)                                                        \
    do {                                                                    \
        if (!(cond)) { printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); failures++; } \
    } while (0)

static std::mt19937_64 rng(0xBF5);
static const double ERR_SIGMA = 3.2;   // bfv/src/lib.rs:17

static Rq rand_u64(const RingParam &p, uint64_t bound) {
    std::vector<uint64_t> c(p.n);
    for (auto &x : c) x = rng() % bound;
    return Rq(p, c);
}
static Rq rand_f64(const RingParam &p, double lo_or_mean, double hi_or_sigma, bool normal) {
    std::vector<uint64_t> c(p.n);
    std::normal_distribution<double> nd(lo_or_mean, hi_or_sigma);
    std::uniform_real_distribution<double> ud(lo_or_mean, hi_or_sigma);
    for (auto &x : c) x = zq_from_f64(p.q, normal ? nd(rng) : ud(rng));
    return Rq(p, c);
}
static Rq add(const Rq &x, const Rq &y) {
    std::vector<uint64_t> c(x.param.n);
    for (size_t i = 0; i < c.size(); i++) { uint64_t v = x.coeffs_v[i] + y.coeffs_v[i]; c[i] = v >= x.param.q ? v - x.param.q : v; }
    return Rq(x.param, c);
}
static Rq neg(const Rq &x) {
    std::vector<uint64_t> c(x.param.n);
    for (size_t i = 0; i < c.size(); i++) c[i] = x.coeffs_v[i] ? x.param.q - x.coeffs_v[i] : 0;
    return Rq(x.param, c);
}
static Rq mul_by_u64(const Rq &x, uint64_t s) {
    std::vector<uint64_t> c(x.param.n);
    for (size_t i = 0; i < c.size(); i++) c[i] = (uint64_t)(((unsigned __int128)x.coeffs_v[i] * (s % x.param.q)) % x.param.q);
    return Rq(x.param, c);
}
static Rq remodule(const Rq &x, uint64_t q) {          // Rq::remodule: same values, reduced mod the new q
    std::vector<uint64_t> c(x.param.n);
    for (size_t i = 0; i < c.size(); i++) c[i] = x.coeffs_v[i] % q;
    return Rq(RingParam{q, x.param.n}, c);
}
static Rq mul_div_round(const Rq &x, uint64_t num, uint64_t den) {   // ring_nq.rs:100-113
    std::vector<uint64_t> c(x.param.n);
    for (size_t i = 0; i < c.size(); i++) c[i] = zq_from_f64(x.param.q, ((double)num * (double)x.coeffs_v[i]) / (double)den);
    return Rq(x.param, c);
}

struct Keys { Rq s; Rq pk0, pk1; };
static Keys new_key(const Param &param) {               // lib.rs:118-139
    Rq s = rand_u64(param.ring, 2);
    s.compute_evals();
    Rq a = rand_u64(param.ring, param.ring.q);
    Rq e = rand_f64(param.ring, 0.0, ERR_SIGMA, true);
    return Keys{s, add(neg(a) * s, e), a};
}
static RLWE encrypt(const Param &param, const Keys &k, const Rq &m) {   // lib.rs:142-163
    Rq u = rand_f64(param.ring, -1.0, 1.0, false);
    Rq e1 = rand_f64(param.ring, 0.0, ERR_SIGMA, true), e2 = rand_f64(param.ring, 0.0, ERR_SIGMA, true);
    Rq md = mul_by_u64(remodule(m, param.ring.q), param.ring.q / param.t);
    return RLWE{add(add(k.pk0 * u, e1), md), add(k.pk1 * u, e2)};
}
static Rq decrypt(const Param &param, const Rq &s, const RLWE &c) {     // lib.rs:165-181
    Rq cs = add(c.c0, c.c1 * s);
    return remodule(mul_div_round(cs, param.t, param.ring.q), param.t);
}
static RLK rlk_key(const Param &param, const Rq &s) {                   // lib.rs:202-225
    const uint64_t pq = param.p * param.ring.q;
    RingParam rp{pq, param.ring.n};
    Rq s_pq = remodule(s, pq);
    Rq a = rand_u64(rp, pq);
    Rq e = rand_f64(rp, 0.0, ERR_SIGMA, true);
    Rq r0 = add(neg(add(tmp_naive_mul(a, s_pq), e)), mul_by_u64(tmp_naive_mul(s_pq, s_pq), param.p));
    return RLK{r0, a};
}
// (m1.to_r() * m2.to_r()).to_rq(t): negacyclic schoolbook over Z, then mod t
static Rq naive_product_mod_t(const Rq &m1, const Rq &m2, uint64_t t) {
    const size_t n = m1.param.n;
    std::vector<long long> r(n, 0);
    for (size_t i = 0; i < n; i++)
        for (size_t j = 0; j < n; j++) {
            long long p = (long long)m1.coeffs_v[i] * (long long)m2.coeffs_v[j];
            if (i + j < n) r[i + j] += p; else r[i + j - n] -= p;
        }
    std::vector<uint64_t> c(n);
    for (size_t i = 0; i < n; i++) { long long v = r[i] % (long long)t; c[i] = (uint64_t)(v < 0 ? v + (long long)t : v); }
    return Rq(RingParam{t, n}, c);
}

// bfv/src/lib.rs:556-601
static void test_mul_relin() {
    const uint64_t q = (1ull << 16) + 1;
    Param param{RingParam{q, 16}, 2, q * q};
    for (int it = 0; it < 200; it++) {
        Rq m1 = rand_u64(param.pt(), param.t), m2 = rand_u64(param.pt(), param.t);
        Keys k = new_key(param);
        RLK rlk = rlk_key(param, k.s);
        RLWE c1 = encrypt(param, k, m1), c2 = encrypt(param, k, m2);
        RLWE c3 = RLWE::mul(param.t, rlk, c1, c2);
        Rq m3 = decrypt(param, k.s, c3);
        EXPECT(m3 == naive_product_mod_t(m1, m2, param.t));
    }
}

// bfv/src/lib.rs:504-554: decrypt the three-term tensor with (1, s, s^2)
static void test_tensor() {
    const uint64_t q = (1ull << 16) + 1;
    Param param{RingParam{q, 16}, 2, q * q};
    for (int it = 0; it < 200; it++) {
        Rq m1 = rand_u64(param.pt(), param.t), m2 = rand_u64(param.pt(), param.t);
        Keys k = new_key(param);
        RLWE c1 = encrypt(param, k, m1), c2 = encrypt(param, k, m2);
        auto [t0, t1, t2] = RLWE::tensor(param.t, c1, c2);
        Rq s2 = k.s * k.s;
        Rq cs = add(add(t0, t1 * k.s), t2 * s2);
        Rq m3 = remodule(mul_div_round(cs, param.t, q), param.t);
        EXPECT(m3 == naive_product_mod_t(m1, m2, param.t));
    }
}

int main() {
    if (fhe_ntt_device_count() < 1) { printf("no HIP device\n"); return 2; }
    test_tensor();
    test_mul_relin();
    printf(failures ? "%d FAILURES\n" : "all host C++ bfv tests passed%.0d\n", failures);
    return failures ? 1 : 0;
}
