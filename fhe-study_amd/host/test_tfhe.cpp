// test_tfhe.cpp — the reference's external-product test restated in C++ over tfhe.hpp:
//   tfhe/src/tggsw.rs:157-196  test_external_product  (n = 64, k = 4, t = 16, beta = 2, l = 64:
//                              decode(decrypt(TGGSW(m1) * TGLWE(m2))) == m1*m2 mod (t, X^n+1))
// Key generation, encryption, encoding and decoding are TEST scaffolding restating
// tfhe/src/tglwe.rs:40-86, tggsw.rs:16-33,97-118, gfhe/src/glwe.rs:140-154,175-180 and
// arith/src/torus.rs:32-35,68-70; every Tn product in them goes through the library.
#include <cmath>
#include <cstdio>
#include <random>

#include "tfhe.hpp"

using namespace tfhe;

static int failures = 0;
#define EXPECT(cond)                                                        \
    do {                                                                    \
        if (!(cond)) { printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); failures++; } \
    } while (0)

static std::mt19937_64 rng(0x7F4E);
static const double ERR_SIGMA = 3.2;   // tfhe/src/lib.rs:14

static uint64_t t64_from_f64(double r) {   // T64::rand, torus.rs:32-35: `r.round() as u64` saturates at 0
    double v = std::round(r);
    return v <= 0.0 ? 0 : (uint64_t)v;
}
static Tn rand_key(size_t n) {             // Uniform(0, 2)
    std::uniform_real_distribution<double> d(0.0, 2.0);
    Tn t{std::vector<uint64_t>(n)};
    for (auto &x : t.coeffs) x = t64_from_f64(d(rng));
    return t;
}
static Tn rand_err(size_t n) {
    std::normal_distribution<double> d(0.0, ERR_SIGMA);
    Tn t{std::vector<uint64_t>(n)};
    for (auto &x : t.coeffs) x = t64_from_f64(d(rng));
    return t;
}
static Tn rand_uniform(size_t n) {
    Tn t{std::vector<uint64_t>(n)};
    for (auto &x : t.coeffs) x = rng();
    return t;
}
static uint64_t mul_div_round(uint64_t v, uint64_t num, uint64_t den) {   // torus.rs:68-70
    double r = std::round(((double)num * (double)v) / (double)den);
    return r >= 18446744073709551616.0 ? UINT64_MAX : (uint64_t)r;
}

struct Param { size_t n, k; uint64_t t; };
using SecretKey = std::vector<Tn>;

// GLWE::encrypt_s, glwe.rs:140-154, with a uniform mask (the reference draws it from the key
// distribution; a uniform one is the harder case for the product)
static TGLWE encrypt_s(const Param &p, const SecretKey &sk, const Tn &m) {
    TGLWE c;
    Tn acc{std::vector<uint64_t>(p.n, 0)};
    for (size_t i = 0; i < p.k; i++) {
        c.a.push_back(rand_uniform(p.n));
        acc = acc + c.a[i] * sk[i];
    }
    c.b = acc + m + rand_err(p.n);
    return c;
}
static Tn decrypt(const SecretKey &sk, const TGLWE &c) {   // glwe.rs:175-180
    Tn acc{std::vector<uint64_t>(c.b.n(), 0)};
    for (size_t i = 0; i < sk.size(); i++) acc = acc + c.a[i] * sk[i];
    return c.b - acc;
}
static TGLev tglev_encrypt_s(const Param &p, uint32_t l, const SecretKey &sk, const Tn &m) {   // tggsw.rs:97-118
    TGLev lev;
    for (uint64_t i = 1; i <= l; i++) lev.rows.push_back(encrypt_s(p, sk, i < 64 ? m * (UINT64_MAX / (1ull << i)) : m));
    return lev;
}
static TGGSW tggsw_encrypt_s(const Param &p, uint32_t l, const SecretKey &sk, const Tn &m) {   // tggsw.rs:16-33
    TGGSW g;
    for (size_t i = 0; i < p.k; i++) g.a.push_back(tglev_encrypt_s(p, l, sk, (-sk[i]) * m));
    g.b = tglev_encrypt_s(p, l, sk, m);
    return g;
}

// tfhe/src/tggsw.rs:157-196
static void test_external_product() {
    Param param{64, 4, 16};
    const uint32_t l = 64;
    for (int it = 0; it < 10; it++) {
        SecretKey sk;
        for (size_t i = 0; i < param.k; i++) sk.push_back(rand_key(param.n));
        std::vector<uint64_t> m1(param.n), m2(param.n);
        for (auto &x : m1) x = rng() % param.t;
        for (auto &x : m2) x = rng() % param.t;
        Tn p1{m1};                                           // TGLev::encode: the message itself
        Tn p2{std::vector<uint64_t>(param.n)};               // TGLWE::encode: scaled by delta
        const uint64_t delta = UINT64_MAX / param.t;
        for (size_t i = 0; i < param.n; i++) p2.coeffs[i] = m2[i] * delta;

        TGGSW tgsw = tggsw_encrypt_s(param, l, sk, p1);
        TGLWE tlwe = encrypt_s(param, sk, p2);
        TGLWE res = tgsw * tlwe;
        Tn rec = decrypt(sk, res);

        // TGLWE::decode, tglwe.rs:59-63, then Rq::from_vec_u64 mod t
        std::vector<uint64_t> got(param.n), want(param.n);
        for (size_t i = 0; i < param.n; i++) got[i] = mul_div_round(rec.coeffs[i], param.t, UINT64_MAX) % param.t;
        std::vector<long long> acc(param.n, 0);
        for (size_t i = 0; i < param.n; i++)
            for (size_t j = 0; j < param.n; j++) {
                long long pr = (long long)(m1[i] * m2[j]);
                if (i + j < param.n) acc[i + j] += pr; else acc[i + j - param.n] -= pr;
            }
        for (size_t i = 0; i < param.n; i++) { long long v = acc[i] % (long long)param.t; want[i] = (uint64_t)(v < 0 ? v + (long long)param.t : v); }
        EXPECT(got == want);
    }
}

// Tn * Tn against the definition (ring_torus.rs:266-298) at full-range operands
static void test_tn_mul() {
    for (size_t n : {2, 64, 1024}) {
        Tn a = rand_uniform(n), b = rand_uniform(n);
        Tn want{std::vector<uint64_t>(n, 0)};
        for (size_t i = 0; i < n; i++)
            for (size_t j = 0; j < n; j++) {
                const uint64_t p = a.coeffs[i] * b.coeffs[j];
                if (i + j < n) want.coeffs[i + j] += p; else want.coeffs[i + j - n] -= p;
            }
        EXPECT((a * b) == want);
    }
}

// TGLWE * Tn and TGLev * Vec<Tn> against the compositions the reference writes
// (tglwe.rs:182-194: r_i * plaintext; tggsw.rs:139-149: zip(v, tlwes).map(glwe_i * a_d_i).sum())
static void test_tglwe_and_tglev_products() {
    const size_t n = 64, k = 3, l = 5;
    auto rand_tglwe = [&] { TGLWE c; for (size_t i = 0; i < k; i++) c.a.push_back(rand_uniform(n)); c.b = rand_uniform(n); return c; };
    TGLWE c = rand_tglwe();
    Tn p = rand_uniform(n);
    TGLWE cp = c * p;
    EXPECT(cp.b == c.b * p);
    for (size_t i = 0; i < k; i++) EXPECT(cp.a[i] == c.a[i] * p);

    TGLev g;
    std::vector<Tn> v;
    for (size_t d = 0; d < l; d++) { g.rows.push_back(rand_tglwe()); v.push_back(rand_uniform(n)); }
    TGLWE sum = g.rows[0] * v[0];
    for (size_t d = 1; d < l; d++) {
        TGLWE term = g.rows[d] * v[d];
        for (size_t i = 0; i < k; i++) sum.a[i] = sum.a[i] + term.a[i];
        sum.b = sum.b + term.b;
    }
    TGLWE gv = g * v;
    EXPECT(gv.b == sum.b);
    for (size_t i = 0; i < k; i++) EXPECT(gv.a[i] == sum.a[i]);
}

int main() {
    if (fhe_ntt_device_count() < 1) { printf("no HIP device\n"); return 2; }
    test_tn_mul();
    test_tglwe_and_tglev_products();
    test_external_product();
    printf(failures ? "%d FAILURES\n" : "all host C++ tfhe tests passed%.0d\n", failures);
    return failures ? 1 : 0;
}
