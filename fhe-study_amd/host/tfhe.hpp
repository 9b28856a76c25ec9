// tfhe.hpp — C++ host mirror of the reference's torus products over the
// C ABI of libfhe_ntt.so:
//
//   reference (Rust)                                          here (C++)
//   arith::T64(u64), Tn{param, coeffs}  torus.rs:13, ring_torus.rs:20-24   tfhe::Tn (u64 words, wrapping)
//   Tn * Tn  (naive_poly_mul mod 2^64)  ring_torus.rs:251-298  operator*(Tn, Tn)      fhe_tn_mul
//   TGLWE(GLWE<Tn>)                     tfhe/src/tglwe.rs:33   tfhe::TGLWE{a, b}
//   TGLev(Vec<TGLWE>)                   tfhe/src/tggsw.rs:65   tfhe::TGLev
//   TGGSW(Vec<TGLev>, TGLev)            tggsw.rs:12-14         tfhe::TGGSW
//   TGLWE * Tn (plaintext product)      tglwe.rs:182-194       operator*(TGLWE, Tn)    fhe_tglwe_mul_tn
//   TGLev * Vec<Tn> -> TGLWE            tggsw.rs:139-149       operator*(TGLev, vector<Tn>)  fhe_tglev_mul
//   TGGSW * TGLWE (external product)    tggsw.rs:45-62         operator*(TGGSW, TGLWE) fhe_tggsw_external_product
//
// Additions are the reference's wrapping u64 adds (torus.rs:80-104) and stay on the host.
#pragma once
#include <cstdint>
#include <vector>

#include "arith.hpp"

namespace tfhe {

using arith::Panic;

struct Tn {
    std::vector<uint64_t> coeffs;   // T64.0 of each coefficient
    size_t n() const { return coeffs.size(); }
    bool operator==(const Tn &o) const { return coeffs == o.coeffs; }
};
inline Tn operator+(const Tn &x, const Tn &y) {
    Tn r{std::vector<uint64_t>(x.n())};
    for (size_t i = 0; i < r.n(); i++) r.coeffs[i] = x.coeffs[i] + y.coeffs[i];   // wrapping_add
    return r;
}
inline Tn operator-(const Tn &x, const Tn &y) {
    Tn r{std::vector<uint64_t>(x.n())};
    for (size_t i = 0; i < r.n(); i++) r.coeffs[i] = x.coeffs[i] - y.coeffs[i];   // wrapping_sub
    return r;
}
inline Tn operator-(const Tn &x) {
    Tn r{std::vector<uint64_t>(x.n())};
    for (size_t i = 0; i < r.n(); i++) r.coeffs[i] = (uint64_t)0 - x.coeffs[i];
    return r;
}
// Tn * &u64, ring_torus.rs:300-310 (coefficient-wise wrapping multiply)
inline Tn operator*(const Tn &x, uint64_t s) {
    Tn r{std::vector<uint64_t>(x.n())};
    for (size_t i = 0; i < r.n(); i++) r.coeffs[i] = x.coeffs[i] * s;
    return r;
}
// ring_torus.rs:251-298: negacyclic product mod 2^64
inline Tn operator*(const Tn &x, const Tn &y) {
    if (x.n() != y.n()) throw Panic(FHE_E_PARAM_MISMATCH, "Tn * Tn: different n");
    Tn r{std::vector<uint64_t>(x.n())};
    arith::check(fhe_tn_mul(x.n(), x.coeffs.data(), y.coeffs.data(), r.coeffs.data(), 1));
    return r;
}

struct TGLWE {   // (a_0 .. a_{k-1}, b)
    std::vector<Tn> a;
    Tn b;
    size_t k() const { return a.size(); }
    std::vector<uint64_t> packed() const {
        std::vector<uint64_t> w;
        for (const Tn &t : a) w.insert(w.end(), t.coeffs.begin(), t.coeffs.end());
        w.insert(w.end(), b.coeffs.begin(), b.coeffs.end());
        return w;
    }
};
struct TGLev { std::vector<TGLWE> rows; };          // l TGLWEs
struct TGGSW { std::vector<TGLev> a; TGLev b; };    // k TGLevs for the mask, one for the body

namespace detail {
inline TGLWE unpack(const std::vector<uint64_t> &w, size_t k, size_t n) {
    TGLWE r;
    for (size_t i = 0; i < k; i++) r.a.push_back(Tn{std::vector<uint64_t>(w.begin() + i * n, w.begin() + (i + 1) * n)});
    r.b = Tn{std::vector<uint64_t>(w.begin() + k * n, w.end())};
    return r;
}
}  // namespace detail

// tglwe.rs:182-194: plaintext multiplication, every component times the Tn
inline TGLWE operator*(const TGLWE &c, const Tn &p) {
    if (c.b.n() != p.n()) throw Panic(FHE_E_PARAM_MISMATCH, "TGLWE * Tn: different n");   // debug_assert_eq!(param)
    std::vector<uint64_t> w = c.packed(), out(w.size());
    arith::check(fhe_tglwe_mul_tn(p.n(), (unsigned)c.k(), w.data(), p.coeffs.data(), out.data(), 1));
    return detail::unpack(out, c.k(), p.n());
}

// tggsw.rs:139-149: sum_d tglev[d] * v[d]
inline TGLWE operator*(const TGLev &g, const std::vector<Tn> &v) {
    if (g.rows.size() != v.size()) throw Panic(FHE_E_INVALID, "TGLev * Vec<Tn>: lengths differ");   // assert_eq!, :143
    const size_t k = g.rows.at(0).k(), n = v.at(0).n(), l = v.size();
    std::vector<uint64_t> pg, pv;
    for (const TGLWE &row : g.rows) { std::vector<uint64_t> w = row.packed(); pg.insert(pg.end(), w.begin(), w.end()); }
    for (const Tn &t : v) pv.insert(pv.end(), t.coeffs.begin(), t.coeffs.end());
    std::vector<uint64_t> out((k + 1) * n);
    arith::check(fhe_tglev_mul(n, (unsigned)k, (unsigned)l, pg.data(), pv.data(), out.data(), 1));
    return detail::unpack(out, k, n);
}

// tggsw.rs:45-62, with beta = 2 and l = 64 as hard-coded there
inline TGLWE operator*(const TGGSW &g, const TGLWE &c) {
    const size_t k = c.k(), n = c.b.n(), l = g.b.rows.size();
    if (g.a.size() != k) throw Panic(FHE_E_INVALID, "TGGSW * TGLWE: lengths differ");   // assert_eq!, tggsw.rs:55
    std::vector<uint64_t> pg;   // [(k+1)][l][(k+1)][n]
    auto push_lev = [&](const TGLev &lev) {
        if (lev.rows.size() != l) throw Panic(FHE_E_INVALID, "TGGSW: TGLevs of different length");
        for (const TGLWE &row : lev.rows) {
            std::vector<uint64_t> w = row.packed();
            pg.insert(pg.end(), w.begin(), w.end());
        }
    };
    for (const TGLev &lev : g.a) push_lev(lev);
    push_lev(g.b);
    std::vector<uint64_t> pc = c.packed(), out(pc.size());
    arith::check(fhe_tggsw_external_product(n, (unsigned)k, (unsigned)l, pg.data(), pc.data(), out.data(), 1));
    return detail::unpack(out, k, n);
}

}  // namespace tfhe
