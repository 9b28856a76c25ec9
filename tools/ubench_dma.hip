// tools/ubench_dma.hip — the persistent, LDS-DMA fed variants of the two n = 2^16 forward passes
// (round 1 experiment, measured SLOWER than the occupancy-driven production kernels: 9.16 ms vs
// 8.36 ms per 16384 polynomials then; the double buffer halves the resident waves, which costs more
// than the prefetch gains — DESIGN.md §5).  Kept out of the library (they used to sit in
// ntt_kernels.hip behind FHE_NTT_DMA=1) and buildable on their own for A/B runs:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ubench_dma tools/ubench_dma.hip && ./tools/ubench_dma
// The output of both pass pairs is compared word for word (same tables, same inputs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../fhe-study_amd/csrc/ntt_kernels.hip"

namespace fhe {
KernelTimer::KernelTimer(const char *, int, hipStream_t st) : slot_(nullptr), st_(st) {}
KernelTimer::~KernelTimer() {}

// one LDS-DMA wave-instruction: lane l copies 16 B from gsrc to LDS byte lds_dst + 16*l.
// m0 is compiler-reserved: saved and restored inside the statement (guide §5.7).
__device__ __forceinline__ void dma16(const void *gsrc, u32 lds_dst_uniform) {
    u32 keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst_uniform)
        : "memory");
}
__device__ __forceinline__ void wait_vmem_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }


// ---------------------------------------------------------------------------
// PERSISTENT, LDS-DMA fed variants of the two passes (the n = 2^16 path).
//
// Why: a pass is fabric-bound (measured copy rate 5.4 TB/s read+write, tools/
// ubench_mem.hip), and a workgroup that loads, computes and stores in sequence
// leaves HBM idle while it computes — LDS caps a CU at ~18 waves of this shape, too
// few to cover that by occupancy alone (PMC: 34 % of wave time parked in waits).
// Here a workgroup lives for many work items and the NEXT item's tile is fetched by
// `global_load_lds_dwordx4` (16 B per lane, no VGPRs, no VALU) into the second of two
// LDS buffers while the current item is in its butterflies; the results of item i-1
// drain to HBM at the same time.  Per item the wave executes one `s_waitcnt vmcnt(0)`
// at a point where everything outstanding was issued a full item ago.
//
//   iteration i (buffer `cur` holds item i, complete and visible):
//     issue DMA(item i+1) -> buf[1-cur]      (last read one iteration ago, before E)
//     registers <- buf[cur] (round-0 window);  barrier
//     round 0;  scatter -> buf[cur];  barrier;  gather (next window);  round 1 ...
//     s_waitcnt vmcnt(0)   (DMA(i+1) and the stores of item i-1: both old)
//     global stores of item i;  barrier (E);  cur ^= 1
// ---------------------------------------------------------------------------
typedef __attribute__((address_space(3))) unsigned char lds_byte;

__device__ __forceinline__ u32 xrow(u32 f) { return f ^ ((f >> 4) & 1u); }

template <int LA, int CW>
struct StridedDmaCfg {
    static constexpr int F = 1 << LA;
    static constexpr int TPF = F / 16;
    static constexpr int TH = TPF * CW;            // 256 for LA = 8, CW = 16
    static constexpr int WAVES = TH / 64;
    static constexpr int NR = (LA + 3) / 4;
    static constexpr int R0 = LA - 4 * (NR - 1);
    static constexpr int A0 = LA - 4;
    static constexpr int TILE_BYTES = F * CW * 8;
    static constexpr int CHUNKS = TILE_BYTES / 1024;          // wave-instructions per tile
    static constexpr int CHUNKS_PER_WAVE = CHUNKS / WAVES;
    static constexpr int LANES_PER_ROW = CW * 8 / 16;         // 16-byte pieces per row segment
    static constexpr int ROWS_PER_CHUNK = 64 / LANES_PER_ROW;
    static constexpr size_t LDS_BYTES = 2 * (size_t)TILE_BYTES + (size_t)F * sizeof(Tw);
    static constexpr int a_of(int j) { return j == 0 ? A0 : LA - R0 - 4 * j; }
    static constexpr int ls0_of(int j) { return j == 0 ? 0 : R0 + 4 * (j - 1); }
    static_assert(NR == 2, "two rounds");
    static_assert(CHUNKS % WAVES == 0 && 64 % LANES_PER_ROW == 0, "tile must split into whole wave chunks");
};

template <int LA, int CW, bool WIDE>
__global__ __launch_bounds__((StridedDmaCfg<LA, CW>::TH)) void ntt_fwd_strided_dma_kernel(PassArgs a) {
    using C = StridedDmaCfg<LA, CW>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Tw *ltw = reinterpret_cast<Tw *>(smem_raw + 2 * C::TILE_BYTES);
    const u32 lds_base = (u32)(uintptr_t)(lds_byte *)smem_raw;
    const u32 tid = threadIdx.x, c = tid % CW, tf = tid / CW;
    const u32 lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u32 lb = a.log_n - LA;               // log2 of the row length
    const u32 lcg = lb - __builtin_ctz(CW);    // log2(column groups per polynomial)
    const u64 items = a.batch << lcg;
    const Mod &m = a.mod;
    for (u32 li = tid; li < (u32)C::F; li += C::TH) ltw[li] = a.tw[li];
    const Tw *tw = ltw;

    auto base_of = [&](u64 item) -> u64 {       // element index of (row 0, column 0 of the group)
        const u64 cg = item & ((1ull << lcg) - 1ull), poly = item >> lcg;
        return (poly << a.log_n) + cg * CW;
    };
    // lane's share of a tile fetch: row (lane / LANES_PER_ROW) of each chunk, 16-byte piece lane % LANES_PER_ROW
    const u32 lrow = lane / C::LANES_PER_ROW, lpiece = lane % C::LANES_PER_ROW;
    auto fetch = [&](u64 item, u32 buf) {
        const unsigned char *src = reinterpret_cast<const unsigned char *>(a.in + base_of(item)) + lpiece * 16;
#pragma unroll
        for (int i = 0; i < C::CHUNKS_PER_WAVE; i++) {
            const u32 chunk = wave * C::CHUNKS_PER_WAVE + i;
            const u64 row = (u64)chunk * C::ROWS_PER_CHUNK + lrow;
            dma16(src + ((row << lb) << 3), lds_base + buf * C::TILE_BYTES + chunk * 1024);
        }
    };

    u64 item = blockIdx.x;
    u32 cur = 0;
    if (item < items) fetch(item, 0);
    wait_vmem_all();
    __syncthreads();   // tile 0 and the twiddles are in LDS
    for (; item < items; item += gridDim.x) {
        const u64 nitem = item + gridDim.x;
        if (nitem < items) fetch(nitem, cur ^ 1u);
        u64 *lds = reinterpret_cast<u64 *>(smem_raw + cur * C::TILE_BYTES);
        u64 v[16];
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = lds[field_of<C::A0>(tf, k) * CW + c];
        __syncthreads();
        round_fwd<C::R0, WIDE>(v, tw, 1u, m);
        {
            constexpr int A = C::a_of(1), LS = C::ls0_of(1);
            // rows are swizzled (row ^= bit 4 of row) in the exchange so that the gather,
            // whose two row values per 32-lane group differ by 16, covers all 64 banks
#pragma unroll
            for (int k = 0; k < 16; k++) lds[xrow(field_of<C::A0>(tf, k)) * CW + c] = v[k];
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 16; k++) v[k] = lds[xrow(field_of<A>(tf, k)) * CW + c];
            round_fwd<4, WIDE>(v, tw, (1u << LS) + (tf >> A), m);
        }
        wait_vmem_all();
        constexpr int ALAST = C::a_of(C::NR - 1);
        const u64 base = base_of(item) + c;
#pragma unroll
        for (int k = 0; k < 16; k++) a.out[base + ((u64)field_of<ALAST>(tf, k) << lb)] = v[k];  // lazy
        __syncthreads();
        cur ^= 1u;
    }
}

// contiguous pass, LP = 8: work item = 16 polynomials x one 256-coefficient block `blk`
// (fixed for the life of the workgroup, so its 255 twiddles sit in LDS).
// DMA layout: unit w (2 KiB) lands at byte w*2048 with its 16-byte pieces rotated by 8*w
// (piece p of the unit at slot (p + 8*w) mod 128), so that the round-0 gather — 16 lanes
// per unit, 4 units per wave — spreads over all 64 banks.
template <int LP>
struct ContigDmaCfg {
    static constexpr int M = 1 << LP;              // 256
    static constexpr int TPB = M / 16;             // 16 threads per unit
    static constexpr int TH = 256;
    static constexpr int W = TH / TPB;             // 16 units per item
    static constexpr int WAVES = TH / 64;
    static constexpr int UNIT_BYTES = M * 8;       // 2048
    static constexpr int TILE = W * M;
    static constexpr int BUF_BYTES = (TILE + TILE / 16) * 8;   // padded layout must fit too
    static constexpr int PIECES = UNIT_BYTES / 16; // 128 pieces per unit
    static constexpr int CHUNKS = W * UNIT_BYTES / 1024;       // 32 wave-instructions per item
    static constexpr int CHUNKS_PER_WAVE = CHUNKS / WAVES;
    static constexpr size_t LDS_BYTES = 2 * (size_t)BUF_BYTES + (size_t)M * sizeof(Tw);
    static_assert(LP == 8, "two radix-16 rounds");
};

template <int LP, bool FINAL, bool WIDE>
__global__ __launch_bounds__(ContigDmaCfg<LP>::TH) void ntt_fwd_contig_dma_kernel(PassArgs a) {
    using C = ContigDmaCfg<LP>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Tw *ltw = reinterpret_cast<Tw *>(smem_raw + 2 * C::BUF_BYTES);
    const u32 lds_base = (u32)(uintptr_t)(lds_byte *)smem_raw;
    const u32 tid = threadIdx.x, w = tid / C::TPB, tf = tid % C::TPB;
    const u32 lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const u32 s0 = a.log_n - LP;
    const u32 blk = blockIdx.x & ((1u << s0) - 1u);
    const u32 chunk0 = blockIdx.x >> s0, nchunks = gridDim.x >> s0;
    const u64 n = 1ull << a.log_n;
    const u64 groups = (a.batch + C::W - 1) / C::W;
    const Mod &m = a.mod;
    stage_twiddles<C::M, C::TH>(ltw, a.tw, s0, blk, tid);
    const Tw *tw = ltw;

    // chunk = 1 KiB = half a unit: unit u = chunk/2, half h = chunk%2; LDS slot s = h*64 + lane
    // holds piece (s - 8*u) mod 128 of the unit
    auto fetch = [&](u64 pg, u32 buf) {
#pragma unroll
        for (int i = 0; i < C::CHUNKS_PER_WAVE; i++) {
            const u32 chunk = wave * C::CHUNKS_PER_WAVE + i;
            const u32 u = chunk >> 1, h = chunk & 1u;
            u64 poly = pg * C::W + u;
            if (poly >= a.batch) poly = a.batch - 1;   // ragged group: fetch something valid, never stored
            const u32 piece = (h * 64 + lane - 8 * u) & (C::PIECES - 1);
            const unsigned char *src =
                reinterpret_cast<const unsigned char *>(a.in + poly * n + (u64)blk * C::M) + piece * 16;
            dma16(src, lds_base + buf * C::BUF_BYTES + chunk * 1024);
        }
    };
    // element f of unit u in the DMA layout (u64 index inside the buffer)
    auto dma_slot = [&](u32 u, u32 f) -> u32 {
        const u32 piece = f >> 1;
        return u * C::M + (((piece + 8 * u) & (C::PIECES - 1)) << 1) + (f & 1u);
    };

    u64 pg = chunk0;
    u32 cur = 0;
    if (pg < groups) fetch(pg, 0);
    wait_vmem_all();
    __syncthreads();
    for (; pg < groups; pg += nchunks) {
        const u64 npg = pg + nchunks;
        if (npg < groups) fetch(npg, cur ^ 1u);
        u64 *lds = reinterpret_cast<u64 *>(smem_raw + cur * C::BUF_BYTES);
        u64 v[16];
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = lds[dma_slot(w, field_of<LP - 4>(tf, k))];
        __syncthreads();
        round_fwd<4, WIDE>(v, tw, 1u, m);                       // local stages 0..3: H = 0
#pragma unroll
        for (int k = 0; k < 16; k++) lds[pad16(w * C::M + field_of<LP - 4>(tf, k))] = v[k];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = lds[pad16(w * C::M + field_of<0>(tf, k))];
        round_fwd<4, WIDE>(v, tw, (1u << 4) + tf, m);           // local stages 4..7: H = tf
        // each thread rewrites exactly the slots it has just read: no barrier needed before
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const u64 x = FINAL ? (WIDE ? canon8(v[k], m) : canon4(v[k], m)) : v[k];
            lds[pad16(w * C::M + field_of<0>(tf, k))] = x;
        }
        __syncthreads();
        wait_vmem_all();
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const u32 e = i * C::TH + tid;
            const u32 wu = e >> LP, f = e & (C::M - 1);
            const u64 p = pg * C::W + wu;
            if (p < a.batch) a.out[p * n + (u64)blk * C::M + f] = lds[pad16(e)];
        }
        __syncthreads();
        cur ^= 1u;
    }
}


// persistent DMA kernels: exactly the workgroups a CU can hold (2 per CU by LDS), each
// walking a strided share of the items
constexpr u64 kPersistentWorkgroups = 256 * 2;

template <bool WIDE>
static hipError_t launch_fwd_strided_dma(const PassArgs &a, hipStream_t st) {
    using C = StridedDmaCfg<8, 16>;
    const u64 ncg = (1ull << (a.log_n - 8)) / 16;
    const u64 items = ncg * a.batch;
    const u64 grid = items < kPersistentWorkgroups ? items : kPersistentWorkgroups;
    if (grid == 0) return hipSuccess;
    if (hipError_t e = allow_big_lds((const void *)ntt_fwd_strided_dma_kernel<8, 16, WIDE>, C::LDS_BYTES)) return e;
    KernelTimer kt("ntt_fwd_strided_dma", 8, st);
    hipLaunchKernelGGL((ntt_fwd_strided_dma_kernel<8, 16, WIDE>), dim3((unsigned)grid), dim3(C::TH),
                       C::LDS_BYTES, st, a);
    return post_launch();
}

template <bool FINAL, bool WIDE>
static hipError_t launch_fwd_contig_dma(const PassArgs &a, hipStream_t st) {
    using C = ContigDmaCfg<8>;
    const u64 nb = 1ull << (a.log_n - 8);
    const u64 groups = (a.batch + C::W - 1) / C::W;
    u64 nchunks = kPersistentWorkgroups / nb;
    if (nchunks < 1) nchunks = 1;
    if (nchunks > groups) nchunks = groups;
    const u64 grid = nb * nchunks;
    if (grid == 0) return hipSuccess;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    if (hipError_t e = allow_big_lds((const void *)ntt_fwd_contig_dma_kernel<8, FINAL, WIDE>, C::LDS_BYTES)) return e;
    KernelTimer kt(FINAL ? "ntt_fwd_contig_dma_final" : "ntt_fwd_contig_dma", 8, st);
    hipLaunchKernelGGL((ntt_fwd_contig_dma_kernel<8, FINAL, WIDE>), dim3((unsigned)grid), dim3(C::TH),
                       C::LDS_BYTES, st, a);
    return post_launch();
}


}  // namespace fhe

using namespace fhe;
typedef unsigned __int128 u128;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

int main(int argc, char **argv) {
    const u64 batch = argc > 1 ? strtoull(argv[1], 0, 10) : 4096;
    const u32 log_n = 16;
    const u64 n = 1ull << log_n, q = 2305843009211596801ull;
    std::vector<Tw> h(n);
    u64 w = 1, g = 1681162619342215248ull;
    for (u64 i = 0; i < n; i++) {   // any valid table: both variants read the same one
        h[i].w = w;
        h[i].wp = (u64)(((u128)w << 64) / q);
        w = (u64)(((u128)w * g) % q);
    }
    Tw *dtw;
    CK(hipMalloc(&dtw, n * sizeof(Tw)));
    CK(hipMemcpy(dtw, h.data(), n * sizeof(Tw), hipMemcpyHostToDevice));
    u64 *x, *y0, *y1;
    CK(hipMalloc(&x, batch * n * 8));
    CK(hipMalloc(&y0, batch * n * 8));
    CK(hipMalloc(&y1, batch * n * 8));
    CK(launch_fill_synthetic(x, batch * n, q, 1, 0, 0));
    DevicePlan p{};
    p.tw_fwd = dtw; p.tw_inv = dtw; p.log_n = log_n; p.wide = true;
    p.mod.q = q; p.mod.q2 = 2 * q; p.mod.nq = 0 - q; p.mod.neg2q = 0 - 2 * q; p.mod.neg4q = 0 - 4 * q;
    p.mod.q2p1 = 2 * q + 1; p.mod.r64 = (u64)((((u128)1) << 64) % q);
    p.mod.r64p = (u64)(((u128)p.mod.r64 << 64) / q); p.mod.onep = (u64)((((u128)1) << 64) / q);
    PassArgs a{};
    a.tw = dtw; a.mod = p.mod; a.batch = batch; a.log_n = log_n;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int variant = 0; variant < 2; variant++) {
        u64 *y = variant ? y1 : y0;
        float best = 1e30f;
        for (int r = 0; r < 5; r++) {
            CK(hipEventRecord(e0));
            if (variant == 0) {
                CK(launch_ntt_forward(p, x, y, batch, 0, 0));
            } else {
                a.in = x; a.out = y;
                CK(launch_fwd_strided_dma<true>(a, 0));
                a.in = y;
                CK((launch_fwd_contig_dma<true, true>(a, 0)));
            }
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (r && ms < best) best = ms;
        }
        printf("%-34s %8.3f ms per %llu polynomials of 2^16\n", variant ? "persistent LDS-DMA passes" : "production passes", best,
               (unsigned long long)batch);
    }
    std::vector<u64> h0(batch * n), h1(batch * n);
    CK(hipMemcpy(h0.data(), y0, batch * n * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h1.data(), y1, batch * n * 8, hipMemcpyDeviceToHost));
    u64 bad = 0;
    for (u64 i = 0; i < batch * n; i++) bad += h0[i] != h1[i];
    printf("word-for-word comparison: %llu mismatches\n", (unsigned long long)bad);
    return bad != 0;
}
