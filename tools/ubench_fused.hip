// tools/ubench_fused.hip — UPPER BOUND of a single-HBM-pass N = 2^16 forward transform
// (diagnostic; results quoted in DESIGN.md §5).  It answers, before any inter-workgroup
// machinery is built: if the second HBM round trip of the two-pass transform disappeared,
// how fast could the kernel go on this chip?
//
// The kernels below are NOT transforms (the twiddle indices of the later rounds are
// arbitrary valid entries): they have the instruction mix, LDS traffic, occupancy and HBM
// traffic of one, which is what bounds it.
//   mode 0  the production strided pass as it is          (8 stages, 16N bytes of HBM)
//   mode 1  load tile, FOUR rounds (16 stages) with LDS exchanges, store: what a fused
//           transform costs when the exchange between the halves is free
//   mode 2  mode 1 + the tile written to and read back from an L2-sized scratch between
//           rounds 2 and 3 (plain stores, sc1 loads): the exchange through the XCD's L2,
//           without the synchronisation
//   mode 3  16 stages, no HBM at all (tile synthesised in registers, result kept in LDS):
//           the VALU/LDS ceiling at the clock the chip then holds
//   ref     the production two-pass transform through launch_ntt_forward
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../fhe-study_amd/csrc/ntt_kernels.hip"

namespace fhe {
KernelTimer::KernelTimer(const char *, int, hipStream_t st) : slot_(nullptr), st_(st) {}
KernelTimer::~KernelTimer() {}

__device__ __forceinline__ u64 ld_sc1(const u64 *p) {
    u64 v;
    asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int MODE>
__global__ __launch_bounds__((StridedCfg<8, 32>::TH)) void fused_bound_kernel(PassArgs a, u64 *scratch, u32 scratch_slots) {
    using C = StridedCfg<8, 32>;
    constexpr int CW = 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64 *lds = reinterpret_cast<u64 *>(smem_raw);
    const u32 tid = threadIdx.x, c = tid % CW, tf = tid / CW;
    const u32 lb = a.log_n - 8;
    const u32 lcg = lb - 5;
    const u32 cg = blockIdx.x & ((1u << lcg) - 1u);
    const u64 poly = (u64)(blockIdx.x >> lcg);
    const u64 ubase = (poly << a.log_n) + (u64)cg * CW;
    const u64 *__restrict__ pin = a.in + ubase;
    u64 *__restrict__ pout = a.out + ubase;
    const Mod &m = a.mod;
    Tw *ltw = reinterpret_cast<Tw *>(smem_raw + C::DATA_BYTES);
    const Tw *tw = ltw;

    u64 v[16];
    if (MODE == 3) {
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = __umul64hi(splitmix64(blockIdx.x * 8192ull + tid * 16 + k), m.q);
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = ld_at<u64>(pin, ((field_of<4>(tf, k) << lb) + c) * 8u);
    }
    for (u32 li = tid; li < (u32)C::F; li += C::TH) ltw[li] = a.tw[li];

    constexpr int B0 = 2, B1 = fwd_bound_out(4, B0);
    round_fwd<4, true, B0, false, false>(v, a.tw, 1u, m);
    exchange_strided<CW, 4, 0, true>(v, lds, c, tf);
    round_fwd<4, true, B1, false, MODE == 0>(v, tw, (1u << 4) + tf, m);
    if (MODE != 0) {
        if (MODE == 2) {
            // through the XCD's L2: slot shared by the workgroups with equal blockIdx % slots
            u64 *s = scratch + (size_t)(blockIdx.x % scratch_slots) * (C::F * CW);
#pragma unroll
            for (int k = 0; k < 16; k++) s[field_of<0>(tf, k) * CW + c] = v[k];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 16; k++) v[k] = ld_sc1(s + field_of<4>(tf, k) * CW + c);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            exchange_strided<CW, 0, 4, false>(v, lds, c, tf);
        }
        round_fwd<4, true, 6, false, false>(v, tw, (1u << 4) + (tf ^ 5u), m);
        exchange_strided<CW, 4, 0, MODE == 2>(v, lds, c, tf);
        round_fwd<4, true, 6, true, false>(v, tw, (1u << 4) + (tf ^ 9u), m);
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = canon4(v[k], m);
    }
    if (MODE == 3) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++) lds[field_of<0>(tf, k) * CW + c] = v[k];
        __syncthreads();
        if (lds[tid] == 0x1234567ull) pout[0] = 1;   // keeps the work alive, practically never taken
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++) st_at(pout, ((field_of<0>(tf, k) << lb) + c) * 8u, v[k]);
    }
}
}  // namespace fhe

using namespace fhe;
typedef unsigned __int128 u128;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

int main(int argc, char **argv) {
    const u64 batch = argc > 1 ? strtoull(argv[1], 0, 10) : 16384;
    const u32 log_n = 16;
    const u64 n = 1ull << log_n, q = 2305843009211596801ull;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs, nominal %.0f MHz; batch %llu polynomials of 2^16\n", prop.name, prop.multiProcessorCount,
           prop.clockRate * 1e-3, (unsigned long long)batch);

    // a table of valid twiddles (powers of a fixed element: not the plan's table, the kernels are cost models)
    std::vector<Tw> h(n);
    u64 w = 1, g = 1681162619342215248ull;
    for (u64 i = 0; i < n; i++) {
        h[i].w = w;
        h[i].wp = (u64)(((u128)w << 64) / q);
        w = (u64)(((u128)w * g) % q);
    }
    Tw *dtw;
    CK(hipMalloc(&dtw, n * sizeof(Tw)));
    CK(hipMemcpy(dtw, h.data(), n * sizeof(Tw), hipMemcpyHostToDevice));
    u64 *x, *y, *scratch;
    CK(hipMalloc(&x, batch * n * 8));
    CK(hipMalloc(&y, batch * n * 8));
    const u32 slots = 256;   // 256 x 64 KiB = 16 MiB: 2 MiB per XCD under round-robin placement
    CK(hipMalloc(&scratch, (size_t)slots * 65536));
    CK(launch_fill_synthetic(x, batch * n, q, 1, 0, 0));
    CK(hipDeviceSynchronize());

    DevicePlan p{};
    p.tw_fwd = dtw; p.tw_inv = dtw; p.log_n = log_n; p.wide = true;
    p.mod.q = q; p.mod.q2 = 2 * q; p.mod.nq = 0 - q; p.mod.neg2q = 0 - 2 * q; p.mod.neg4q = 0 - 4 * q;
    p.mod.q2p1 = 2 * q + 1; p.mod.r64 = (u64)((((u128)1) << 64) % q);
    p.mod.r64p = (u64)(((u128)p.mod.r64 << 64) / q); p.mod.onep = (u64)((((u128)1) << 64) / q);
    PassArgs a{};
    a.in = x; a.out = y; a.tw = dtw; a.mod = p.mod; a.batch = batch; a.log_n = log_n;

    using C = StridedCfg<8, 32>;
    const unsigned grid = (unsigned)(batch * 8);
    CK(hipFuncSetAttribute((const void *)fused_bound_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
    CK(hipFuncSetAttribute((const void *)fused_bound_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
    CK(hipFuncSetAttribute((const void *)fused_bound_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
    CK(hipFuncSetAttribute((const void *)fused_bound_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto run = [&](const char *name, int mode) {
        float best = 1e30f, sum = 0;
        const int reps = 6;
        for (int r = 0; r < reps; r++) {
            CK(hipEventRecord(e0));
            switch (mode) {
                case 0: hipLaunchKernelGGL(fused_bound_kernel<0>, dim3(grid), dim3(C::TH), C::LDS_BYTES, 0, a, scratch, slots); break;
                case 1: hipLaunchKernelGGL(fused_bound_kernel<1>, dim3(grid), dim3(C::TH), C::LDS_BYTES, 0, a, scratch, slots); break;
                case 2: hipLaunchKernelGGL(fused_bound_kernel<2>, dim3(grid), dim3(C::TH), C::LDS_BYTES, 0, a, scratch, slots); break;
                case 3: hipLaunchKernelGGL(fused_bound_kernel<3>, dim3(grid), dim3(C::TH), C::LDS_BYTES, 0, a, scratch, slots); break;
                default: CK(launch_ntt_forward(p, x, y, batch, 0, 0)); break;
            }
            CK(hipGetLastError());
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2) { sum += ms; if (ms < best) best = ms; }
        }
        const double avg = sum / (reps - 2);
        printf("%-44s avg %7.3f ms  best %7.3f ms  -> %6.3f M NTT/s if this were the whole transform (%.1f %% of 8 TB/s)\n", name, avg, best,
               batch / avg * 1e-3, batch / (avg * 1e-3) * 1048576.0 / 8e12 * 100.0);
        fflush(stdout);
    };
    for (int rep = 0; rep < 2; rep++) {
        run("ref: production two-pass transform", 9);
        run("mode 0: strided pass alone (8 stages)", 0);
        run("mode 1: 16 stages, one HBM pass, free exchange", 1);
        run("mode 2: mode 1 + L2 scratch round trip", 2);
        run("mode 3: 16 stages, no HBM", 3);
    }
    return 0;
}
