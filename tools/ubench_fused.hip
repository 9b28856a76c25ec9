// tools/ubench_fused.hip — UPPER BOUND of a single-HBM-pass N = 2^16 forward transform
// (diagnostic; results quoted in DESIGN.md §5).  It answers, before any inter-workgroup
// machinery is built: if the second HBM round trip of the two-pass transform disappeared,
// how fast could the kernel go on this chip?
//
// Round 5: rebuilt on the PRODUCTION arithmetic (the 13-instruction pseudo-Mersenne butterflies,
// round_fwd_pm; rounds 2-4 quoted the 21-instruction Shoup form) and run at 16, 8 and 4 waves per
// CU (workgroups of 512 or 256 threads; the occupancy is set by padding the LDS allocation).
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
// Build (in-tree, the binary travels to the GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -w '-DCONTIG_CASES(X)=X(8)' -o tools/ubench_fused \
//         tools/ubench_fused.hip fhe-study_amd/csrc/generic63.hip fhe-study_amd/csrc/ntt_kernels_q62.hip
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

template <int MODE, int CW>
__global__ __launch_bounds__((StridedCfg<8, CW>::TH)) void fused_bound_kernel(PassArgs a, u64 *scratch, u32 scratch_slots) {
    using C = StridedCfg<8, CW>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u64 *lds = reinterpret_cast<u64 *>(smem_raw);
    const u32 tid = threadIdx.x, c = tid % CW, tf = tid / CW;
    const u32 lb = a.log_n - 8;
    const u32 lcg = lb - __builtin_ctz(CW);
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
        ld16<true>(v, pin, 4 + lb, ((field_of<4>(tf, 0) << lb) + c) * 8u);
    }
    for (u32 li = tid; li < (u32)C::F; li += C::TH) ltw[li] = a.tw[li];

    constexpr int P0 = kPmOne, P1 = pm_fwd_bound_out(4, P0), P2 = pm_fwd_bound_out(4, P1), P3 = pm_fwd_bound_out(4, P2);
    round_fwd_pm<4, P0, true>(v, a.tw, 1u, m);
    exchange_strided<CW, 4, 0, true>(v, lds, c, tf);
    round_fwd_pm<4, P1, false>(v, tw, (1u << 4) + tf, m);
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
        round_fwd_pm<4, P2, false>(v, tw, (1u << 4) + (tf ^ 5u), m);
        exchange_strided<CW, 4, 0, MODE == 2>(v, lds, c, tf);
        round_fwd_pm<4, P3, false>(v, tw, (1u << 4) + (tf ^ 9u), m);
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = pm_canon(v[k], m);
    }
    if (MODE == 3) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++) lds[field_of<0>(tf, k) * CW + c] = v[k];
        __syncthreads();
        if (lds[tid] == 0x1234567ull) pout[0] = 1;   // keeps the work alive, practically never taken
    } else {
        st16<true>(pout, lb, ((field_of<0>(tf, 0) << lb) + c) * 8u, v);
    }
}
}  // namespace fhe

using namespace fhe;
typedef unsigned __int128 u128;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int MODE, int CW>
static void launch_mode(const PassArgs &a, u64 *scratch, u32 slots, unsigned grid, size_t lds) {
    using C = StridedCfg<8, CW>;
    static bool set = false;
    if (!set) { CK(hipFuncSetAttribute((const void *)fused_bound_kernel<MODE, CW>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set = true; }
    hipLaunchKernelGGL((fused_bound_kernel<MODE, CW>), dim3(grid), dim3(C::TH), lds, 0, a, scratch, slots);
}

int main(int argc, char **argv) {
    const u64 batch = argc > 1 ? strtoull(argv[1], 0, 10) : 16384;
    const u32 log_n = 16;
    const u64 n = 1ull << log_n, q = 2305843009211596801ull;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs, nominal %.0f MHz; batch %llu polynomials of 2^16; pseudo-Mersenne butterflies (13 instructions)\n", prop.name, prop.multiProcessorCount,
           prop.clockRate * 1e-3, (unsigned long long)batch);

    // a table of valid twiddles {w, w 2^32 mod q} (powers of a fixed element: not the plan's table, the kernels are cost models)
    std::vector<Tw> h(n);
    u64 w = 1, g = 1681162619342215248ull;
    for (u64 i = 0; i < n; i++) {
        h[i].w = w;
        h[i].wp = (u64)((((u128)w) << 32) % q);
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
    p.tw_fwd = dtw; p.tw_inv = dtw; p.tw_fwd_pm = dtw; p.tw_inv_pm = dtw; p.log_n = log_n; p.wide = true; p.arith = kArPMersenne;
    p.mod.q = q; p.mod.q2 = 2 * q; p.mod.nq = 0 - q; p.mod.neg2q = 0 - 2 * q; p.mod.neg4q = 0 - 4 * q;
    p.mod.q2p1 = 2 * q + 1; p.mod.r64 = (u64)((((u128)1) << 64) % q);
    p.mod.r64p = (u64)(((u128)p.mod.r64 << 64) / q); p.mod.onep = (u64)((((u128)1) << 64) / q);
    const u32 k = 61; const u64 delta = (1ull << k) - q;
    p.mod.q3p1 = 3 * q + 1; p.mod.pm_k = k; p.mod.pm_delta = (u32)delta; p.mod.pm_c2 = (u32)(2 * delta);
    p.mod.pm_sh = k - 31; p.mod.pm_mask = (1u << (k - 31)) - 1u; p.mod.pm_rsh = k - 32; p.mod.pm_rmask = (1u << (k - 32)) - 1u;
    PassArgs a{};
    a.in = x; a.out = y; a.tw = dtw; a.mod = p.mod; a.batch = batch; a.log_n = log_n;

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    // cw: columns per workgroup (32: 512 threads, 16: 256 threads); wpc: waves per CU wanted (LDS padded so that only that many fit)
    auto run = [&](const char *name, int mode, int cw, int wpc) {
        const size_t need = cw == 32 ? StridedCfg<8, 32>::LDS_BYTES : StridedCfg<8, 16>::LDS_BYTES;
        const int waves_per_wg = cw == 32 ? 8 : 4;
        const int wgs = wpc / waves_per_wg;
        size_t lds = need;
        if (mode != 9) {
            if (wgs < 1 || (size_t)(160 * 1024) / wgs < need) return;
            lds = (size_t)(160 * 1024) / wgs;                 // exactly `wgs` workgroups fit a CU's 160 KiB
        }
        const unsigned grid = (unsigned)(batch * (256 / cw));
        float best = 1e30f, sum = 0;
        const int reps = 6;
        for (int r = 0; r < reps; r++) {
            CK(hipEventRecord(e0));
#define L(M_) if (cw == 32) launch_mode<M_, 32>(a, scratch, slots, grid, lds); else launch_mode<M_, 16>(a, scratch, slots, grid, lds);
            switch (mode) {
                case 0: L(0) break;
                case 1: L(1) break;
                case 2: L(2) break;
                case 3: L(3) break;
                default: CK(launch_ntt_forward(p, x, y, batch, 0, 0)); break;
            }
#undef L
            CK(hipGetLastError());
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2) { sum += ms; if (ms < best) best = ms; }
        }
        const double avg = sum / (reps - 2);
        printf("%-46s %3d thr %2d waves/CU  avg %7.3f ms  best %7.3f ms  -> %6.3f M NTT/s if this were the whole transform (%.1f %% of 8 TB/s)\n", name,
               cw * 16, mode == 9 ? 16 : wpc, avg, best, batch / avg * 1e-3, batch / (avg * 1e-3) * 1048576.0 / 8e12 * 100.0);
        fflush(stdout);
    };
    for (int rep = 0; rep < 2; rep++) {
        run("ref: production two-pass transform", 9, 32, 16);
        for (int cw : {32, 16})
            for (int wpc : {16, 8, 4}) {
                run("mode 0: strided pass alone (8 stages)", 0, cw, wpc);
                run("mode 1: 16 stages, one HBM pass, free exchange", 1, cw, wpc);
                run("mode 2: mode 1 + L2 scratch round trip", 2, cw, wpc);
                run("mode 3: 16 stages, no HBM", 3, cw, wpc);
            }
    }
    return 0;
}
