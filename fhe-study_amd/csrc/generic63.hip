// generic63.hip — NTT::ntt / intt / pointwise product for moduli 2^62 <= q < 2^63 (arith/src/ntt.rs:44-110,
// ring_nq.rs:601-604) in PLAIN kernels.  The reference's Zq works for every q below 2^63 (its `self.v + rhs.v`, zq.rs:225,
// is the limit); the lazy butterflies of ntt_kernels.hip need 4q < 2^64, so this range runs STRICT butterflies — every
// value canonical between stages, three conditional subtractions per butterfly (zq_device.hpp: ct_bfly63 / gs_bfly63).
// Since round 5 the transforms and products of this range at n >= 16 run in the two-pass / fused kernels of
// ntt_kernels.hip with those butterflies (AR = 3).  What stays here: the pointwise product, n < 16, and — under
// FHE_G63_PLAIN=1 — the whole range as it ran until round 5: up to four stages per launch on 2^R coefficients per thread, in
// place in global memory, ceil(log2 n / 4) launches per transform (0.86 M NTT/s at n = 2^16).  The two forms are compared
// word for word by tests/test_round5.py::test_strict_moduli_two_builds_of_the_same_words.
#include "ntt_kernels.hpp"

namespace fhe {

namespace {

// add63 / sub63 / mul63: zq_device.hpp (strict arithmetic, every value canonical)
__device__ __forceinline__ u64 mul63(u64 y, const Tw &w, const Mod &m) { return mul63(y, w.w, w.wp, m); }

// Stages s0 .. s0+R-1 of the forward transform (ntt.rs:49-70: m = 2^s, t = n / 2m, S = roots[m + i]) on the 2^R
// coefficients j = hi * (tl << R) + k * tl + lo, k = 0 .. 2^R - 1, tl = n >> (s0 + R): stage s0 + i pairs the k that
// differ in bit R-1-i, and the block index j / 2t of such a pair is (hi << i) + (k >> (R - i)).
template <int R>
__global__ __launch_bounds__(256) void g63_fwd_kernel(const u64 *in, u64 *out, const Tw *__restrict__ tw,
                                                      Mod m, u32 log_n, u32 s0, u64 threads) {
    const u64 g = (u64)blockIdx.x * 256 + threadIdx.x;
    if (g >= threads) return;
    const u32 lt = log_n - s0 - R;                      // log2 tl
    const u64 per = 1ull << (log_n - R);                // threads per polynomial
    const u64 poly = g >> (log_n - R), idx = g & (per - 1);
    const u64 lo = idx & ((1ull << lt) - 1), hi = idx >> lt;
    const u64 base = (poly << log_n) + (hi << (lt + R)) + lo;
    u64 v[1 << R];
#pragma unroll
    for (int k = 0; k < (1 << R); k++) v[k] = in[base + ((u64)k << lt)];
#pragma unroll
    for (int i = 0; i < R; i++) {
        const int half = 1 << (R - 1 - i);
#pragma unroll
        for (int k = 0; k < (1 << R); k++) {
            if (k & half) continue;
            const Tw w = tw[(1ull << (s0 + i)) + (hi << i) + (u64)(k >> (R - i))];
            const u64 U = v[k], V = mul63(v[k + half], w, m);
            v[k] = add63(U, V, m);
            v[k + half] = sub63(U, V, m);
        }
    }
#pragma unroll
    for (int k = 0; k < (1 << R); k++) out[base + ((u64)k << lt)] = v[k];
}

// Stages s0+R-1 down to s0 of the inverse transform (ntt.rs:83-98: S = roots_inv[m + i], r[j+t] = (U - V) * S), the
// same index algebra; SCALE: the group holds stage 0, the values leave multiplied by n^-1 (ntt.rs:100-102).
// MUL: the input is the pointwise product in .* in2 (ring_nq.rs:601-604), also stored to `evals` when not null.
template <int R, bool SCALE, bool MUL>
__global__ __launch_bounds__(256) void g63_inv_kernel(const u64 *in, const u64 *in2, u64 *evals,      // may alias one another
                                                      u64 *out, const Tw *__restrict__ tw, Mod m, Tw ninv, u32 log_n, u32 s0,
                                                      u64 threads) {
    const u64 g = (u64)blockIdx.x * 256 + threadIdx.x;
    if (g >= threads) return;
    const u32 lt = log_n - s0 - R;
    const u64 per = 1ull << (log_n - R);
    const u64 poly = g >> (log_n - R), idx = g & (per - 1);
    const u64 lo = idx & ((1ull << lt) - 1), hi = idx >> lt;
    const u64 base = (poly << log_n) + (hi << (lt + R)) + lo;
    u64 v[1 << R];
#pragma unroll
    for (int k = 0; k < (1 << R); k++) {
        const u64 at = base + ((u64)k << lt);
        v[k] = in[at];
        if (MUL) {
            v[k] = mul_mod_var63(v[k], in2[at], m);
            if (evals) evals[at] = v[k];
        }
    }
#pragma unroll
    for (int i = R - 1; i >= 0; i--) {
        const int half = 1 << (R - 1 - i);
#pragma unroll
        for (int k = 0; k < (1 << R); k++) {
            if (k & half) continue;
            const Tw w = tw[(1ull << (s0 + i)) + (hi << i) + (u64)(k >> (R - i))];
            const u64 U = v[k], V = v[k + half];
            v[k] = add63(U, V, m);
            v[k + half] = mul63(sub63(U, V, m), w, m);
        }
    }
#pragma unroll
    for (int k = 0; k < (1 << R); k++) out[base + ((u64)k << lt)] = SCALE ? mul63(v[k], ninv, m) : v[k];
}

__global__ __launch_bounds__(256) void g63_pointwise_kernel(const u64 *x, const u64 *y, u64 *z,
                                                            u64 count, Mod m) {
    const u64 stride = (u64)gridDim.x * 256;
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) z[i] = mul_mod_var63(x[i], y[i], m);
}

template <int R>
hipError_t fwd_group(const DevicePlan &p, const u64 *in, u64 *out, u64 batch, u32 s0, hipStream_t st) {
    const u64 threads = batch << (p.log_n - R);
    KernelTimer kt("g63_fwd", R, st);
    g63_fwd_kernel<R><<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st>>>(in, out, p.tw_fwd, p.mod, p.log_n, s0, threads);
    return hipGetLastError();
}
template <int R, bool SCALE, bool MUL>
hipError_t inv_group(const DevicePlan &p, const u64 *in, const u64 *in2, u64 *evals, u64 *out, u64 batch, u32 s0, hipStream_t st) {
    const u64 threads = batch << (p.log_n - R);
    KernelTimer kt("g63_inv", R, st);
    g63_inv_kernel<R, SCALE, MUL><<<dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st>>>(in, in2, evals, out, p.tw_inv, p.mod, p.ninv,
                                                                                                   p.log_n, s0, threads);
    return hipGetLastError();
}
template <bool SCALE, bool MUL>
hipError_t inv_group_r(int r, const DevicePlan &p, const u64 *in, const u64 *in2, u64 *evals, u64 *out, u64 batch, u32 s0, hipStream_t st) {
    switch (r) {
        case 1: return inv_group<1, SCALE, MUL>(p, in, in2, evals, out, batch, s0, st);
        case 2: return inv_group<2, SCALE, MUL>(p, in, in2, evals, out, batch, s0, st);
        case 3: return inv_group<3, SCALE, MUL>(p, in, in2, evals, out, batch, s0, st);
        default: return inv_group<4, SCALE, MUL>(p, in, in2, evals, out, batch, s0, st);
    }
}

}  // namespace

// the grid of a group is batch * n / 2^R / 256 workgroups (R >= 1): batch * n <= 2^38 keeps it below 2^31
static inline bool g63_fits(const DevicePlan &p, u64 batch) { return batch <= (1ull << 38 >> p.log_n); }

hipError_t launch_g63_forward(const DevicePlan &p, const u64 *in, u64 *out, u64 batch, hipStream_t st) {
    if (batch == 0) return hipSuccess;
    if (!g63_fits(p, batch)) return hipErrorInvalidValue;
    const u64 *src = in;
    for (u32 s0 = 0; s0 < p.log_n;) {                   // the short group first: the later ones are the coalesced ones
        const int r = s0 == 0 && (p.log_n & 3u) ? (int)(p.log_n & 3u) : 4;
        hipError_t e;
        switch (r) {
            case 1: e = fwd_group<1>(p, src, out, batch, s0, st); break;
            case 2: e = fwd_group<2>(p, src, out, batch, s0, st); break;
            case 3: e = fwd_group<3>(p, src, out, batch, s0, st); break;
            default: e = fwd_group<4>(p, src, out, batch, s0, st); break;
        }
        if (e != hipSuccess) return e;
        src = out;
        s0 += (u32)r;
    }
    return hipSuccess;
}

hipError_t launch_g63_inverse(const DevicePlan &p, const u64 *in, const u64 *in2, u64 *evals_out, u64 *out, u64 batch, hipStream_t st) {
    if (batch == 0) return hipSuccess;
    if (!g63_fits(p, batch)) return hipErrorInvalidValue;
    // groups from the last stages down; the group that holds stage 0 (the short one, if any) scales by n^-1
    const u32 L = p.log_n;
    const u64 *src = in;
    bool first = true;
    for (u32 top = L; top > 0;) {                        // stages [top - r, top)
        const int r = top >= 4 ? 4 : (int)top;
        const u32 s0 = top - (u32)r;
        const bool scale = s0 == 0, mul = first && in2 != nullptr;
        hipError_t e;
        if (scale && mul) e = inv_group_r<true, true>(r, p, src, in2, evals_out, out, batch, s0, st);
        else if (scale) e = inv_group_r<true, false>(r, p, src, nullptr, nullptr, out, batch, s0, st);
        else if (mul) e = inv_group_r<false, true>(r, p, src, in2, evals_out, out, batch, s0, st);
        else e = inv_group_r<false, false>(r, p, src, nullptr, nullptr, out, batch, s0, st);
        if (e != hipSuccess) return e;
        src = out;
        first = false;
        top = s0;
    }
    return hipSuccess;
}

hipError_t launch_g63_pointwise(const DevicePlan &p, const u64 *x, const u64 *y, u64 *z, u64 count, hipStream_t st) {
    if (count == 0) return hipSuccess;
    const u64 want = (count + 255) / 256;
    KernelTimer kt("g63_pointwise", 0, st);
    g63_pointwise_kernel<<<dim3((unsigned)(want < 16384 ? want : 16384)), dim3(256), 0, st>>>(x, y, z, count, p.mod);
    return hipGetLastError();
}

}  // namespace fhe
