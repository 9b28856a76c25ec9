// arith.hpp — C++ host mirror of the reference's `arith::{RingParam, Zq, Rq, NTT}`
// surface for the NTT path, over the C ABI of libfhe_ntt.so (include/fhe_ntt.h).
//
// The reference is Rust (arnaucube/fhe-study, crate `arith`); no Rust toolchain
// exists in this image, so the host side above the C ABI is written in C++ with the
// reference's names, argument meaning and error behaviour:
//
//   reference (Rust)                                      here (C++)
//   RingParam{q,n}              arith/src/ring.rs:6-10     arith::RingParam
//   Zq{q,v}                     arith/src/zq.rs:6-10       arith::Zq
//   Rq::from_vec_u64            ring_nq.rs:160-163         Rq::from_vec_u64
//   Rq::coeffs                  ring_nq.rs:144-146         Rq::coeffs
//   Rq::compute_evals           ring_nq.rs:147-150         Rq::compute_evals
//   NTT::ntt / NTT::intt        ntt.rs:44-73 / 78-110      NTT::ntt / NTT::intt
//   impl Mul for Rq / &Rq       ring_nq.rs:490-503         operator*
//   Rq::mul(&mut self,&mut rhs) ring_nq.rs:294-296,564     Rq::mul (mul_mut)
//   PartialEq for Rq            ring_nq.rs:401-405         operator==
//   panic!/assert!              ntt.rs:116-130, ring_nq.rs:565,587   throws arith::Panic
//
// This header only needs include/fhe_ntt.h: no HIP, no torch.  Every transform runs
// on the GPU through the library; there is no host fallback.
#pragma once
#include <cstdint>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/fhe_ntt.h"

namespace arith {

// what a Rust `panic!` is to the reference's callers
struct Panic : std::runtime_error {
    int code;
    Panic(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

inline void check(int rc) {
    if (rc != FHE_OK) throw Panic(rc, std::string("fhe_ntt: ") + fhe_last_error());
}

struct RingParam {
    uint64_t q;
    size_t n;
    bool operator==(const RingParam &o) const { return q == o.q && n == o.n; }
    bool operator!=(const RingParam &o) const { return !(*this == o); }
};

struct Zq {
    uint64_t q, v;
    bool operator==(const Zq &o) const { return q == o.q && v == o.v; }
};

inline const fhe_ntt_plan *plan_of(const RingParam &p) {
    const fhe_ntt_plan *pl = nullptr;
    check(fhe_ntt_plan_get(p.q, p.n, &pl));  // memoised like CACHE, ntt.rs:18
    return pl;
}

class Rq {
  public:
    RingParam param;
    std::vector<uint64_t> coeffs_v;                  // the `v` of each Zq, canonical
    std::optional<std::vector<uint64_t>> evals_v;    // cached NTT image (ring_nq.rs:24-26)

    Rq(const RingParam &p, std::vector<uint64_t> c,
       std::optional<std::vector<uint64_t>> e = std::nullopt)
        : param(p), coeffs_v(std::move(c)), evals_v(std::move(e)) {
        if (coeffs_v.size() != p.n) throw Panic(FHE_E_INVALID, "coefficient vector length != n");
    }

    // ring_nq.rs:160-163 (+ Zq::from_u64 zq.rs:21-31, + X^n+1 fold ring_nq.rs:132-141)
    static Rq from_vec_u64(const RingParam &p, const std::vector<uint64_t> &coeffs) {
        if (coeffs.size() < p.n) throw Panic(FHE_E_INVALID, "fewer than n coefficients");
        std::vector<uint64_t> c(coeffs.size());
        for (size_t i = 0; i < coeffs.size(); i++) c[i] = coeffs[i] % p.q;
        for (size_t i = p.n; i < c.size(); i++) {
            uint64_t &lo = c[i - p.n];
            lo = lo >= c[i] ? lo - c[i] : (p.q + lo) - c[i];  // Zq::sub, zq.rs:259-276
        }
        c.resize(p.n);
        return Rq(p, std::move(c));
    }

    std::vector<Zq> coeffs() const {  // ring_nq.rs:144-146
        std::vector<Zq> r(coeffs_v.size());
        for (size_t i = 0; i < r.size(); i++) r[i] = Zq{param.q, coeffs_v[i]};
        return r;
    }

    void compute_evals();   // ring_nq.rs:147-150
    Rq mul(Rq &rhs);        // Rq::mul(&mut self, &mut rhs) = mul_mut, ring_nq.rs:294-296,564-583

    // ring_nq.rs:401-405: coefficients and param; evals do not take part
    bool operator==(const Rq &o) const { return param == o.param && coeffs_v == o.coeffs_v; }
    bool operator!=(const Rq &o) const { return !(*this == o); }
};

struct NTT {
    // ntt.rs:44-73: returns an Rq whose coefficients are the NTT image, evals = None
    static Rq ntt(const Rq &a) {
        std::vector<uint64_t> out(a.param.n);
        check(fhe_ntt_forward(plan_of(a.param), a.coeffs_v.data(), out.data(), 1));
        return Rq(a.param, std::move(out));
    }
    // ntt.rs:78-110
    static Rq intt(const Rq &a) {
        std::vector<uint64_t> out(a.param.n);
        check(fhe_ntt_inverse(plan_of(a.param), a.coeffs_v.data(), out.data(), 1));
        return Rq(a.param, std::move(out));
    }
};

inline void Rq::compute_evals() { evals_v = NTT::ntt(*this).coeffs_v; }

namespace detail {
inline Rq mul_impl(const Rq &lhs, const Rq &rhs, std::vector<uint64_t> *a_evals,
                   std::vector<uint64_t> *b_evals) {
    if (lhs.param != rhs.param)  // assert_eq!(lhs.param, rhs.param), ring_nq.rs:565,587
        throw Panic(FHE_E_PARAM_MISMATCH, "assertion `left == right` failed: lhs.param != rhs.param");
    const size_t n = lhs.param.n;
    const bool ae = lhs.evals_v.has_value(), be = rhs.evals_v.has_value();
    std::vector<uint64_t> c(n), c_evals(n);
    if (a_evals) a_evals->resize(n);
    if (b_evals) b_evals->resize(n);
    check(fhe_rq_mul(plan_of(lhs.param), ae ? lhs.evals_v->data() : lhs.coeffs_v.data(), ae,
                     be ? rhs.evals_v->data() : rhs.coeffs_v.data(), be, c.data(), c_evals.data(),
                     a_evals ? a_evals->data() : nullptr, b_evals ? b_evals->data() : nullptr, 1));
    return Rq(lhs.param, std::move(c), std::move(c_evals));  // the product carries its evals (:606)
}
}  // namespace detail

// `mul`, ring_nq.rs:586-607 and the Mul impls :490-503
inline Rq operator*(const Rq &lhs, const Rq &rhs) { return detail::mul_impl(lhs, rhs, nullptr, nullptr); }

// `mul_mut`, ring_nq.rs:564-583: also stores the operands' evals back into them
inline Rq Rq::mul(Rq &rhs) {
    std::vector<uint64_t> ae, be;
    Rq c = detail::mul_impl(*this, rhs, &ae, &be);
    if (!evals_v) evals_v = std::move(ae);
    if (!rhs.evals_v) rhs.evals_v = std::move(be);
    return c;
}

}  // namespace arith
