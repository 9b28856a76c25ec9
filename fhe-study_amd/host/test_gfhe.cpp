// test_gfhe.cpp — the reference's tests of the batch surfaces around Rq products, restated in C++
// over gfhe.hpp (and therefore over the C ABI and the HIP kernels):
//   gfhe/src/glwe.rs:582-626  test_key_switch   (encrypt under sk, switch to sk2, decrypt, decode)
//   gfhe/src/glwe.rs:493-527  GLWE * R against the schoolbook product
// plus the definitions themselves as cross-checks: each one-call surface must equal the
// composition of single `Rq * Rq` products the reference writes (tuple_ring.rs:117-155,
// glev.rs:68-80, glwe.rs:126-137).  Needs a GPU; run by tests/test_host_cpp.py under `-m gpu`.
//
// Key generation, encryption and encoding below are TEST scaffolding (the reference's scheme
// logic, out of scope of the library): small secrets, rounded Gaussian noise, fixed seeds.
#include <cmath>
#include <cstdio>
#include <random>

#include "gfhe.hpp"

using namespace gfhe;

static int failures = 0;
#define EXPECT(cond)                                                        \
    do {                                                                    \
        if (!(cond)) { printf("FAIL %s:%d: %s\n", __FILE__, __LINE__, #cond); failures++; } \
    } while (0)

static std::mt19937_64 rng(0x6F4E);

// ---- scaffolding: the reference's samplers and encoders ----------------------------------------
static uint64_t from_f64(uint64_t q, double e) {   // Zq::from_f64, zq.rs:32-39
    long long v = (long long)std::llround(e);
    long long r = v % (long long)q;
    return (uint64_t)(r < 0 ? r + (long long)q : r);
}
static Rq rand_key(const RingParam &p) {           // R::rand(rng, Uniform(0,2), ring): glwe.rs:81,144
    std::uniform_real_distribution<double> d(0.0, 2.0);
    std::vector<uint64_t> c(p.n);
    for (auto &x : c) x = from_f64(p.q, d(rng));
    return Rq(p, c);
}
static Rq rand_err(const RingParam &p, double sigma) {   // R::rand(rng, Normal(0, sigma), ring)
    std::normal_distribution<double> d(0.0, sigma);
    std::vector<uint64_t> c(p.n);
    for (auto &x : c) x = from_f64(p.q, d(rng));
    return Rq(p, c);
}
static Rq rand_uniform(const RingParam &p, uint64_t bound) {
    std::vector<uint64_t> c(p.n);
    for (auto &x : c) x = rng() % bound;
    return Rq(p, c);
}
static TR rand_tr(size_t k, const RingParam &p) {
    TR t{k, {}};
    for (size_t i = 0; i < k; i++) t.r.push_back(rand_key(p));
    return t;
}
static Rq mul_by_u64(const Rq &a, uint64_t s) {    // Rq::mul_by_u64, ring_nq.rs:274-281
    std::vector<uint64_t> c(a.param.n);
    for (size_t i = 0; i < c.size(); i++) c[i] = (uint64_t)(((unsigned __int128)a.coeffs_v[i] * (s % a.param.q)) % a.param.q);
    return Rq(a.param, c);
}
static Rq encode(const Param &param, const Rq &m) {   // GLWE::encode, glwe.rs:185-190
    return mul_by_u64(Rq(param.ring, m.coeffs_v), param.ring.q / param.t);
}
static Rq decode(const Param &param, const Rq &p) {   // GLWE::decode, glwe.rs:192-196 (mul_div_round then remodule)
    std::vector<uint64_t> c(p.param.n);
    for (size_t i = 0; i < c.size(); i++)
        c[i] = from_f64(param.ring.q, std::round(((double)param.t * (double)p.coeffs_v[i]) / (double)param.ring.q)) % param.t;
    return Rq(param.pt(), c);
}
static GLWE encrypt_s(const Param &param, const SecretKey &sk, const Rq &m) {   // glwe.rs:140-154
    TR a = rand_tr(param.k, param.ring);
    Rq e = rand_err(param.ring, param.err_sigma);
    return GLWE{a, add(add(a * sk.s, m), e)};
}
static GLev glev_encrypt_s(const Param &param, uint32_t beta, uint32_t l, const SecretKey &sk, const Rq &m) {   // glev.rs:36-56
    GLev g;
    uint64_t bi = 1;
    for (uint32_t i = 1; i <= l; i++) {
        bi *= beta;
        g.rows.push_back(encrypt_s(param, sk, mul_by_u64(m, param.ring.q / bi)));
    }
    return g;
}
static KSK new_ksk(const Param &param, uint32_t beta, uint32_t l, const SecretKey &sk, const SecretKey &new_sk) {   // glwe.rs:107-125
    KSK k;
    for (size_t i = 0; i < sk.s.k; i++) k.levs.push_back(glev_encrypt_s(param, beta, l, new_sk, sk.s.r[i]));
    return k;
}
// Zq::decompose, base 2 (zq.rs:176-190), per coefficient and transposed (Rq::decompose, ring_nq.rs:67-78)
static std::vector<Rq> decompose2(const Rq &a, uint32_t l) {
    std::vector<std::vector<uint64_t>> d(l, std::vector<uint64_t>(a.param.n));
    for (size_t j = 0; j < a.param.n; j++) {
        const uint64_t v = a.coeffs_v[j];
        for (uint32_t i = 0; i < l; i++) d[i][j] = v >= (1ull << (l & 63)) ? 1 % a.param.q : (v >> (l - 1 - i)) & 1;
    }
    std::vector<Rq> out;
    for (auto &row : d) out.emplace_back(a.param, row);
    return out;
}

// ---- gfhe/src/glwe.rs:582-626 ----------------------------------------------------------------------
static void test_key_switch() {
    Param param{3.2, RingParam{(1ull << 16) + 1, 128}, 16, 2};
    const uint32_t beta = 2, l = 16;
    for (int it = 0; it < 3; it++) {
        SecretKey sk{rand_tr(param.k, param.ring)}, sk2{rand_tr(param.k, param.ring)};
        KSK ksk = new_ksk(param, beta, l, sk, sk2);          // switches from sk to sk2
        Rq m = rand_uniform(param.pt(), param.t);
        Rq p = encode(param, m);
        GLWE c = encrypt_s(param, sk, p);
        GLWE c2 = c.key_switch(param, beta, l, ksk);
        Rq m_recovered = decode(param, c2.decrypt(sk2));
        EXPECT(m == m_recovered);
        EXPECT(decode(param, c.decrypt(sk)) == m);           // and the unswitched one under its own key

        // the definition, glwe.rs:126-137, from single products
        GLWE rhs{TR::zero(param.k, param.ring), zero(param.ring)};
        for (size_t i = 0; i < param.k; i++) rhs = rhs + ksk.levs[i] * decompose2(c.a.r[i], l);
        GLWE want = GLWE{TR::zero(param.k, param.ring), c.b} - rhs;
        EXPECT(c2.b == want.b);
        for (size_t i = 0; i < param.k; i++) EXPECT(c2.a.r[i] == want.a.r[i]);
    }
}

// each batch surface == the composition of Rq products it is defined as
static void test_surfaces_equal_their_definitions() {
    for (RingParam p : {RingParam{(1ull << 16) + 1, 64}, RingParam{2305843009211596801ull, 1024}}) {
        const size_t k = 3, l = 4;
        TR a{k, {}}, b{k, {}};
        for (size_t i = 0; i < k; i++) { a.r.push_back(rand_uniform(p, p.q)); b.r.push_back(rand_uniform(p, p.q)); }
        Rq s = rand_uniform(p, p.q);
        // TR . TR, tuple_ring.rs:117-134
        Rq dot = zero(p);
        for (size_t i = 0; i < k; i++) dot = add(dot, a.r[i] * b.r[i]);
        EXPECT((a * b) == dot);
        // TR * R, tuple_ring.rs:137-155
        TR as = a * s;
        for (size_t i = 0; i < k; i++) EXPECT(as.r[i] == a.r[i] * s);
        // GLWE * R, glwe.rs:263-280
        GLWE c{a, rand_uniform(p, p.q)};
        GLWE cs = c * s;
        EXPECT(cs.b == c.b * s);
        for (size_t i = 0; i < k; i++) EXPECT(cs.a.r[i] == c.a.r[i] * s);
        // GLev * Vec<R>, glev.rs:68-80
        GLev g;
        std::vector<Rq> v;
        for (size_t d = 0; d < l; d++) {
            TR m{k, {}};
            for (size_t i = 0; i < k; i++) m.r.push_back(rand_uniform(p, p.q));
            g.rows.push_back(GLWE{m, rand_uniform(p, p.q)});
            v.push_back(rand_uniform(p, p.q));
        }
        GLWE sum{TR::zero(k, p), zero(p)};
        for (size_t d = 0; d < l; d++) sum = sum + g.rows[d] * v[d];
        GLWE gv = g * v;
        EXPECT(gv.b == sum.b);
        for (size_t i = 0; i < k; i++) EXPECT(gv.a.r[i] == sum.a.r[i]);
    }
}

static void test_panics() {
    RingParam p{(1ull << 16) + 1, 8};
    TR a = TR::zero(2, p), b = TR::zero(3, p);
    bool threw = false;
    try { (void)(a * b); } catch (const Panic &) { threw = true; }     // debug_assert_eq!(self.k, other.k)
    EXPECT(threw);
    threw = false;
    Param param{3.2, p, 2, 2};
    try { (void)GLWE{a, zero(p)}.key_switch(param, 2, 4, KSK{}); } catch (const Panic &) { threw = true; }   // zip_eq panics
    EXPECT(threw);
}

int main() {
    if (fhe_ntt_device_count() < 1) { printf("no HIP device\n"); return 2; }
    test_surfaces_equal_their_definitions();
    test_key_switch();
    test_panics();
    printf(failures ? "%d FAILURES\n" : "all host C++ gfhe tests passed%.0d\n", failures);
    return failures ? 1 : 0;
}
