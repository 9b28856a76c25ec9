// mac_kernel.hpp — multiply-accumulate over rows in the NTT domain, shared by the external
// product (zring.hip) and the gfhe batch surfaces (glue.hip):
//
//   out[b][c][j] = sum_{t<T} G[g_b][t][c][j] * D[b][t][j]   (mod q)
//
// G is shared by the batch (gstride = 0: a key) or per batch element (gstride = T*nc*n words).
// With T = (k+1)*l, nc = k+1 this is `TGLev * Vec<Tn>` summed over the k+1 TGLevs
// (tfhe/src/tggsw.rs:57-59,145); with T = l it is GLev * Vec<R> (gfhe/src/glev.rs:68-80); with
// T = k, nc = 1, per-element G it is TR . TR (arith/src/tuple_ring.rs:117-134).
//
// One thread owns coefficients (j, j+1) of output rows (c0, c0+1): every D word is loaded once,
// as 16 bytes, for both rows; sums run in 128-bit accumulators reduced once per kMacChunk terms
// (zq_device.hpp).  All operands must be canonical (< q).  Q63 (2^62 <= q < 2^63, with CHUNK = kMacChunk63 = 2): the
// strict reduction, every partial result canonical — 4q does not fit a word there.
#pragma once
#include "zq_device.hpp"

namespace fhe {

template <int CHUNK = (int)kMacChunk, bool Q63 = false>
__global__ __launch_bounds__(256) void mac_rows_kernel(const u64 *__restrict__ G, const u64 *__restrict__ D,
                                                       u64 *__restrict__ out, u64 batch, u32 n, u32 T, u32 nc,
                                                       u64 gstride, Mod m) {
    const u32 cch = (nc + 1) / 2, nh = n / 2;
    const u64 total = batch * cch * nh, stride = (u64)gridDim.x * 256;
    for (u64 idx = (u64)blockIdx.x * 256 + threadIdx.x; idx < total; idx += stride) {
        const u32 jp = (u32)(idx % nh);
        const u64 bc = idx / nh;
        const u32 c0 = 2 * (u32)(bc % cch);
        const u64 b = bc / cch;
        const bool two = c0 + 1 < nc;
        const ulonglong2 *g = reinterpret_cast<const ulonglong2 *>(G + b * gstride + (u64)c0 * n) + jp;
        const ulonglong2 *d = reinterpret_cast<const ulonglong2 *>(D + b * T * n) + jp;
        const u64 gt = (u64)nc * nh;   // ulonglong2 per term of G
        MacAcc a00, a01, a10, a11;
        for (u32 t0 = 0; t0 < T; t0 += CHUNK) {
#pragma unroll
            for (int u = 0; u < CHUNK; u++) {
                const u32 t = t0 + u;
                if (t < T) {
                    const ulonglong2 dv = d[(u64)t * nh];
                    const ulonglong2 g0 = g[t * gt];
                    a00.mac(g0.x, dv.x);
                    a01.mac(g0.y, dv.y);
                    if (two) {
                        const ulonglong2 g1 = g[t * gt + nh];
                        a10.mac(g1.x, dv.x);
                        a11.mac(g1.y, dv.y);
                    }
                }
            }
            if constexpr (Q63) {
                static_assert(!Q63 || CHUNK <= (int)kMacChunk63, "q < 2^63: two terms per fold");
                a00.fold63(m); a01.fold63(m);
                if (two) { a10.fold63(m); a11.fold63(m); }
            } else {
                a00.fold(m); a01.fold(m);
                if (two) { a10.fold(m); a11.fold(m); }
            }
        }
        ulonglong2 *o = reinterpret_cast<ulonglong2 *>(out + (b * nc + c0) * n) + jp;
        o[0] = ulonglong2{(u64)a00.v, (u64)a01.v};          // after fold() the accumulator IS the canonical sum
        if (two) o[nh] = ulonglong2{(u64)a10.v, (u64)a11.v};
    }
}

// the launch, by the plan's modulus range
static inline void launch_mac_rows(bool q63, unsigned grid, hipStream_t st, const u64 *G, const u64 *D, u64 *out, u64 batch, u32 n, u32 T, u32 nc,
                                   u64 gstride, const Mod &m) {
    if (q63) hipLaunchKernelGGL((mac_rows_kernel<(int)kMacChunk63, true>), dim3(grid), dim3(256), 0, st, G, D, out, batch, n, T, nc, gstride, m);
    else hipLaunchKernelGGL((mac_rows_kernel<>), dim3(grid), dim3(256), 0, st, G, D, out, batch, n, T, nc, gstride, m);
}

// workgroups for mac_rows_kernel over `batch` elements of nc rows of n words
static inline u64 mac_rows_threads(u64 batch, u32 nc, u64 n) { return batch * ((nc + 1) / 2) * (n / 2); }

}  // namespace fhe
