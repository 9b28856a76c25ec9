// gfhe.hpp — C++ host mirror of the reference's batch surfaces around `Rq` products
// (SURVEY.md §8f row N3), over the C ABI of libfhe_ntt.so:
//
//   reference (Rust)                                         here (C++)
//   arith::TR<R>{k, r}              tuple_ring.rs:17-21       gfhe::TR
//   &TR * &TR -> R (dot product)    tuple_ring.rs:117-134     operator*(TR, TR)      fhe_tr_dot
//   &TR * &R  -> TR                 tuple_ring.rs:137-155     operator*(TR, Rq)      fhe_tr_mul_r
//   gfhe::GLWE<R>(TR<R>, R)         gfhe/src/glwe.rs:57       gfhe::GLWE{a, b}
//   GLWE * R                        glwe.rs:263-280           operator*(GLWE, Rq)    fhe_tr_mul_r (k+1 rows)
//   GLWE::key_switch                glwe.rs:126-137           GLWE::key_switch       fhe_glwe_key_switch
//   GLWE::decrypt                   glwe.rs:175-180           GLWE::decrypt
//   gfhe::GLev<R>(Vec<GLWE<R>>)     gfhe/src/glev.rs:13       gfhe::GLev
//   GLev * Vec<R> -> GLWE           glev.rs:68-80             operator*(GLev, vector<Rq>)  fhe_glev_mul
//   KSK<R>(Vec<GLev<R>>)            glwe.rs:66                gfhe::KSK
//
// Every product runs on the GPU through the library.  The additions and subtractions between
// products are the reference's own element-wise loops (zq.rs:219-231,259-276) and stay on the
// host here, as they would in a shim that swaps only the product bodies.
#pragma once
#include "arith.hpp"

namespace gfhe {

using arith::Panic;
using arith::RingParam;
using arith::Rq;

// element-wise glue exactly as Zq::add / Zq::sub (zq.rs:219-231,259-276)
inline Rq add(const Rq &x, const Rq &y) {
    if (x.param != y.param) throw Panic(FHE_E_PARAM_MISMATCH, "Rq + Rq: different RingParam");
    std::vector<uint64_t> c(x.param.n);
    for (size_t i = 0; i < c.size(); i++) {
        const uint64_t v = x.coeffs_v[i] + y.coeffs_v[i];
        c[i] = v >= x.param.q ? v - x.param.q : v;
    }
    return Rq(x.param, std::move(c));
}
inline Rq sub(const Rq &x, const Rq &y) {
    if (x.param != y.param) throw Panic(FHE_E_PARAM_MISMATCH, "Rq - Rq: different RingParam");
    std::vector<uint64_t> c(x.param.n);
    for (size_t i = 0; i < c.size(); i++)
        c[i] = x.coeffs_v[i] >= y.coeffs_v[i] ? x.coeffs_v[i] - y.coeffs_v[i] : (x.param.q + x.coeffs_v[i]) - y.coeffs_v[i];
    return Rq(x.param, std::move(c));
}
inline Rq zero(const RingParam &p) { return Rq(p, std::vector<uint64_t>(p.n, 0)); }

namespace detail {
// rows of equal-param polynomials packed row-major, the layout of include/fhe_ntt.h
inline std::vector<uint64_t> pack(const std::vector<Rq> &rows, const RingParam &p) {
    std::vector<uint64_t> out;
    out.reserve(rows.size() * p.n);
    for (const Rq &r : rows) {
        if (r.param != p) throw Panic(FHE_E_PARAM_MISMATCH, "rows with different RingParam");
        out.insert(out.end(), r.coeffs_v.begin(), r.coeffs_v.end());
    }
    return out;
}
inline std::vector<Rq> unpack(const std::vector<uint64_t> &w, const RingParam &p) {
    std::vector<Rq> rows;
    for (size_t i = 0; i + p.n <= w.size(); i += p.n)
        rows.emplace_back(p, std::vector<uint64_t>(w.begin() + i, w.begin() + i + p.n));
    return rows;
}
}  // namespace detail

struct TR {  // tuple_ring.rs:17-21
    size_t k;
    std::vector<Rq> r;
    static TR zero(size_t k, const RingParam &p) { return TR{k, std::vector<Rq>(k, gfhe::zero(p))}; }
    const RingParam &param() const { return r.at(0).param; }
};

// tuple_ring.rs:117-134: sum_i A_i * B_i — one library call (k products, one inverse transform)
inline Rq operator*(const TR &a, const TR &b) {
    if (a.k != b.k) throw Panic(FHE_E_INVALID, "TR * TR: different k");  // debug_assert_eq!(self.k, other.k)
    const RingParam &p = a.param();
    std::vector<uint64_t> pa = detail::pack(a.r, p), pb = detail::pack(b.r, p), c(p.n);
    arith::check(fhe_tr_dot(arith::plan_of(p), pa.data(), pb.data(), c.data(), (unsigned)a.k, 1));
    return Rq(p, std::move(c));
}
// tuple_ring.rs:137-155
inline TR operator*(const TR &a, const Rq &s) {
    const RingParam &p = a.param();
    std::vector<uint64_t> pa = detail::pack(a.r, p), out(a.k * p.n);
    arith::check(fhe_tr_mul_r(arith::plan_of(p), pa.data(), s.coeffs_v.data(), out.data(), (unsigned)a.k, 1));
    return TR{a.k, detail::unpack(out, p)};
}
inline TR operator+(const TR &a, const TR &b) {
    TR c{a.k, {}};
    for (size_t i = 0; i < a.k; i++) c.r.push_back(add(a.r[i], b.r[i]));
    return c;
}
inline TR operator-(const TR &a, const TR &b) {
    TR c{a.k, {}};
    for (size_t i = 0; i < a.k; i++) c.r.push_back(sub(a.r[i], b.r[i]));
    return c;
}

struct Param {  // glwe.rs:20-52
    double err_sigma;
    RingParam ring;
    size_t k;
    uint64_t t;
    RingParam pt() const { return RingParam{t, ring.n}; }
};

struct SecretKey { TR s; };           // glwe.rs:60
struct GLWE;
struct GLev { std::vector<GLWE> rows; };   // glev.rs:13
struct KSK { std::vector<GLev> levs; };    // glwe.rs:66

struct GLWE {  // glwe.rs:57: GLWE(TR<R>, R) = (mask a_0..a_{k-1}, body b)
    TR a;
    Rq b;
    // (a_0 .. a_{k-1}, b) packed as the library's [(k+1)][n]
    std::vector<uint64_t> packed() const {
        std::vector<uint64_t> w = detail::pack(a.r, b.param);
        w.insert(w.end(), b.coeffs_v.begin(), b.coeffs_v.end());
        return w;
    }
    static GLWE from_packed(const std::vector<uint64_t> &w, size_t k, const RingParam &p) {
        std::vector<Rq> rows = detail::unpack(w, p);
        Rq body = rows.back();
        rows.pop_back();
        return GLWE{TR{k, std::move(rows)}, std::move(body)};
    }
    // glwe.rs:175-180: b - d . sk
    Rq decrypt(const SecretKey &sk) const { return sub(b, a * sk.s); }
    // glwe.rs:126-137: (0, b) - sum_i ksk_i * decompose(a_i, beta, l) — one library call
    GLWE key_switch(const Param &param, uint32_t beta, uint32_t l, const KSK &ksk) const;
};
inline GLWE operator+(const GLWE &x, const GLWE &y) { return GLWE{x.a + y.a, add(x.b, y.b)}; }
inline GLWE operator-(const GLWE &x, const GLWE &y) { return GLWE{x.a - y.a, sub(x.b, y.b)}; }

// glwe.rs:263-280: every component times the plaintext polynomial
inline GLWE operator*(const GLWE &c, const Rq &s) {
    const RingParam &p = c.b.param;
    std::vector<uint64_t> w = c.packed(), out(w.size());
    arith::check(fhe_tr_mul_r(arith::plan_of(p), w.data(), s.coeffs_v.data(), out.data(), (unsigned)(c.a.k + 1), 1));
    return GLWE::from_packed(out, c.a.k, p);
}

namespace detail {
inline std::vector<uint64_t> pack(const GLev &g) {   // [l][(k+1)][n]
    std::vector<uint64_t> w;
    for (const GLWE &row : g.rows) {
        std::vector<uint64_t> r = row.packed();
        w.insert(w.end(), r.begin(), r.end());
    }
    return w;
}
}  // namespace detail

// glev.rs:68-80: sum_d glev[d] * v[d]
inline GLWE operator*(const GLev &g, const std::vector<Rq> &v) {
    if (g.rows.size() != v.size()) throw Panic(FHE_E_INVALID, "GLev * Vec<R>: lengths differ");   // zip_eq
    const RingParam &p = v.at(0).param;
    const size_t k = g.rows.at(0).a.k;
    std::vector<uint64_t> pg = detail::pack(g), pv = detail::pack(v, p), out((k + 1) * p.n);
    arith::check(fhe_glev_mul(arith::plan_of(p), (unsigned)k, (unsigned)v.size(), pg.data(), pv.data(), out.data(), 1));
    return GLWE::from_packed(out, k, p);
}

inline GLWE GLWE::key_switch(const Param &param, uint32_t beta, uint32_t l, const KSK &ksk) const {
    if (ksk.levs.size() != a.k) throw Panic(FHE_E_INVALID, "key_switch: KSK has != k GLevs");   // zip_eq, glwe.rs:132
    std::vector<uint64_t> pk;   // [k][l][(k+1)][n]
    for (const GLev &g : ksk.levs) {
        if (g.rows.size() != l) throw Panic(FHE_E_INVALID, "key_switch: GLev has != l levels");
        std::vector<uint64_t> w = detail::pack(g);
        pk.insert(pk.end(), w.begin(), w.end());
    }
    std::vector<uint64_t> w = packed(), out(w.size());
    arith::check(fhe_glwe_key_switch(arith::plan_of(param.ring), (unsigned)a.k, beta, l, w.data(), pk.data(), out.data(), 1));
    return from_packed(out, a.k, param.ring);
}

}  // namespace gfhe
