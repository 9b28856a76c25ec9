// bfv.hpp — C++ host mirror of the reference's BFV ciphertext product,
// over the C ABI of libfhe_ntt.so:
//
//   reference (Rust, bfv/src/lib.rs)                      here (C++)
//   Param{ring, t, p}                     :20-33          bfv::Param
//   RLK(Rq, Rq)   (mod p*q)               :43             bfv::RLK
//   RLWE(Rq, Rq)                          :47             bfv::RLWE
//   RLWE::tensor(t, a, b) -> (c0,c1,c2)   :59-85          RLWE::tensor          fhe_bfv_tensor
//   RLWE::mul(t, rlk, a, b)               :87-90          RLWE::mul             fhe_bfv_mul
//   tmp_naive_mul(a, b)                   :92-97          bfv::tmp_naive_mul    fhe_r_naive_mul + fold
//
// The reference forms these products by schoolbook convolution over Z with `as i64` wrapping
// and f64 scale-and-round; the library returns the same words through exact multi-prime NTTs.
#pragma once
#include <cmath>
#include <tuple>

#include "arith.hpp"

namespace bfv {

using arith::Panic;
using arith::RingParam;
using arith::Rq;

struct Param {
    RingParam ring;
    uint64_t t, p;
    RingParam pt() const { return RingParam{t, ring.n}; }
};
struct RLK { Rq r0, r1; };

// Zq::from_f64 (zq.rs:32-39) on an exactly representable integer-valued double
inline uint64_t zq_from_f64(uint64_t q, double e) {
    double r = std::round(e);
    long long v = r >= 9223372036854775807.0 ? INT64_MAX : r <= -9223372036854775808.0 ? INT64_MIN : (long long)r;   // `as i64` saturates
    if (v < 0 || (uint64_t)v >= q) {
        long long m = v % (long long)q;
        return (uint64_t)(m < 0 ? m + (long long)q : m);
    }
    return (uint64_t)v;
}

// tmp_naive_mul, lib.rs:92-97: Rq::from_vec_i64(naive_mul(a.to_r(), b.to_r())) — the convolution on
// the GPU (fhe_r_naive_mul), `*c as f64` + from_f64 + the X^n+1 fold (ring_nq.rs:164-170,132-141) here
inline Rq tmp_naive_mul(const Rq &a, const Rq &b) {
    if (a.param != b.param) throw Panic(FHE_E_PARAM_MISMATCH, "tmp_naive_mul: different RingParam");
    const size_t n = a.param.n;
    const uint64_t q = a.param.q;
    std::vector<int64_t> ia(a.coeffs_v.begin(), a.coeffs_v.end()), ib(b.coeffs_v.begin(), b.coeffs_v.end()), conv(2 * n);
    arith::check(fhe_r_naive_mul(n, ia.data(), ib.data(), conv.data(), 1));
    std::vector<uint64_t> c(n);
    for (size_t i = 0; i < n; i++) c[i] = zq_from_f64(q, (double)conv[i]);
    for (size_t i = n; i < 2 * n - 1; i++) {
        const uint64_t hi = zq_from_f64(q, (double)conv[i]);
        c[i - n] = c[i - n] >= hi ? c[i - n] - hi : (q + c[i - n]) - hi;
    }
    return Rq(a.param, std::move(c));
}

struct RLWE {
    Rq c0, c1;

    // lib.rs:59-85
    static std::tuple<Rq, Rq, Rq> tensor(uint64_t t, const RLWE &a, const RLWE &b) {
        const RingParam &p = a.c0.param;
        const size_t n = p.n;
        std::vector<uint64_t> ab;
        for (const Rq *x : {&a.c0, &a.c1, &b.c0, &b.c1}) {
            if (x->param != p) throw Panic(FHE_E_PARAM_MISMATCH, "RLWE::tensor: different RingParam");
            ab.insert(ab.end(), x->coeffs_v.begin(), x->coeffs_v.end());
        }
        std::vector<uint64_t> c(3 * n);
        arith::check(fhe_bfv_tensor(p.q, n, t, ab.data(), c.data(), 1));
        auto part = [&](size_t i) { return Rq(p, std::vector<uint64_t>(c.begin() + i * n, c.begin() + (i + 1) * n)); };
        return {part(0), part(1), part(2)};
    }

    // lib.rs:87-90: relinearize_204(rlk, tensor(t, a, b))
    static RLWE mul(uint64_t t, const RLK &rlk, const RLWE &a, const RLWE &b) {
        const RingParam &p = a.c0.param;
        const size_t n = p.n;
        std::vector<uint64_t> ab, k;
        for (const Rq *x : {&a.c0, &a.c1, &b.c0, &b.c1}) {
            if (x->param != p) throw Panic(FHE_E_PARAM_MISMATCH, "RLWE::mul: different RingParam");
            ab.insert(ab.end(), x->coeffs_v.begin(), x->coeffs_v.end());
        }
        for (const Rq *x : {&rlk.r0, &rlk.r1}) k.insert(k.end(), x->coeffs_v.begin(), x->coeffs_v.end());
        std::vector<uint64_t> out(2 * n);
        arith::check(fhe_bfv_mul(p.q, n, t, rlk.r0.param.q, k.data(), ab.data(), out.data(), 1));
        return RLWE{Rq(p, std::vector<uint64_t>(out.begin(), out.begin() + n)),
                    Rq(p, std::vector<uint64_t>(out.begin() + n, out.end()))};
    }
};

}  // namespace bfv
