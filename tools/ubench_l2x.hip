// tools/ubench_l2x.hip — can an XCD's 4 MiB L2 carry the exchange between the two halves of a 2^16-point transform while
// the polynomials stream through it?  (diagnostic for DESIGN.md section 5, round 3: "what >= 40 % would take".)
//
// Persistent workgroups copy in -> out in chunks of 16 words per thread (the passes' shape).  MODE 1 adds a round trip
// of every chunk through a scratch slot in global memory between the loads and the stores: written by the workgroup,
// barrier, read back by OTHER threads of the same workgroup (same CU, hence same XCD and same L2).  The scratch
// footprint per XCD = resident workgroups per XCD x slots x chunk bytes is swept from 0.5 to 16 MiB; the question is at
// which footprint the round trip stops being free (scratch lines evicted by the streaming traffic and re-fetched).
// Run under `rocprofv3 --pmc FETCH_SIZE WRITE_SIZE` to see the scratch bytes that left the L2.
//
//   hipcc --offload-arch=gfx950 -O3 -o ubench_l2x tools/ubench_l2x.hip && ./ubench_l2x
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned u32;

template <int TH, int MODE, bool NT>
__global__ __launch_bounds__(TH) void k_l2x(const u64* __restrict__ in, u64* __restrict__ out, u64* scratch, u32 nchunks, u32 slots, int spin) {
  const u32 tid = threadIdx.x;
  u32 it = 0;
  for (u32 c = blockIdx.x; c < nchunks; c += gridDim.x, it++) {
    const u64 base = (u64)c * (16 * TH);
    u64 v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = NT ? __builtin_nontemporal_load(in + base + k * TH + tid) : in[base + k * TH + tid];
    for (int i = 0; i < spin; i++) {
#pragma unroll
      for (int k = 0; k < 16; k++) v[k] = v[k] * 0x9E3779B97F4A7C15ull + (u64)i;
    }
    if (MODE == 1) {
      u64* s = scratch + ((u64)blockIdx.x * slots + it % slots) * (16 * TH);
#pragma unroll
      for (int k = 0; k < 16; k++) s[k * TH + tid] = v[k];
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 16; k++) v[k] = s[k * TH + (TH - 1 - tid)];
      if (slots == 1) __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 16; k++) {
      if (NT) __builtin_nontemporal_store(v[k], out + base + k * TH + tid);
      else out[base + k * TH + tid] = v[k];
    }
  }
}

template <typename F> static float timeit(F f) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int r = 0; r < 4; r++) { hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (r && ms < best) best = ms; }
  return best;
}

int main(int argc, char** argv) {
  const u64 words = 1ull << 29, bytes = words * 8;                  // 4 GiB in, 4 GiB out
  u64 *a, *b, *s; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&s, 512ull << 20);
  {  // in[i] = i
    u64* h = (u64*)malloc(1 << 24);
    for (u64 off = 0; off < bytes; off += 1 << 24) { for (u64 i = 0; i < (1 << 21); i++) h[i] = off / 8 + i; hipMemcpy((char*)a + off, h, 1 << 24, hipMemcpyHostToDevice); }
    free(h);
  }
  hipMemset(b, 0, bytes); hipMemset(s, 0, 512ull << 20);
  printf("%-64s %9s %8s\n", "variant", "ms", "TB/s r+w");
  auto check = [&](int TH, int mode) {
    u64 h[4096];
    int bad = 0;
    for (u64 c : {0ull, 12345ull, words / (16 * TH) - 1}) {
      hipMemcpy(h, b + c * 16 * TH, sizeof(u64) * 2 * TH, hipMemcpyDeviceToHost);
      for (int i = 0; i < 2 * TH; i++) { u64 t = i % TH, k = i / TH; u64 want = c * 16 * TH + k * TH + (mode ? TH - 1 - t : t); bad += h[i] != want; }
    }
    return bad;
  };
#define RUN(TH, MODE, NT, KPC, SLOTS, SPIN) { \
    const u32 grid = 256 * (KPC), nch = (u32)(words / (16 * TH)); \
    const double fp = 32.0 * (KPC) * (SLOTS) * 16 * TH * 8 / 1048576.0; \
    if ((u64)grid * (SLOTS) * 16 * TH * 8 <= (512ull << 20)) { \
      hipMemset(b, 0, 1 << 20); \
      float ms = timeit([&] { k_l2x<TH, MODE, NT><<<grid, TH>>>(a, b, s, nch, SLOTS, SPIN); }); \
      char nm[128]; snprintf(nm, 128, "%s %4d thr x %2d wg/CU (%2d waves/CU) slots %d %s spin %d: %5.2f MiB/XCD", MODE ? "scratch" : "copy   ", TH, KPC, (KPC) * TH / 64, SLOTS, NT ? "nt" : "  ", SPIN, MODE ? fp : 0.0); \
      printf("%-64s %9.3f %8.2f  %s\n", nm, ms, 2.0 * bytes / ms * 1e-9, check(TH, MODE) ? "MISMATCH" : "ok"); fflush(stdout); } }
  for (int spin : {0}) {
    RUN(256, 0, false, 2, 1, spin) RUN(256, 0, false, 4, 1, spin) RUN(256, 0, false, 8, 1, spin) RUN(256, 0, true, 4, 1, spin) RUN(256, 0, true, 8, 1, spin)
    RUN(512, 0, false, 2, 1, spin) RUN(512, 0, false, 4, 1, spin) RUN(1024, 0, false, 1, 1, spin) RUN(1024, 0, false, 2, 1, spin)
    RUN(256, 1, false, 1, 1, spin) RUN(256, 1, false, 2, 1, spin) RUN(256, 1, false, 4, 1, spin) RUN(256, 1, false, 8, 1, spin)
    RUN(256, 1, true, 1, 1, spin) RUN(256, 1, true, 2, 1, spin) RUN(256, 1, true, 4, 1, spin) RUN(256, 1, true, 8, 1, spin)
    RUN(256, 1, false, 2, 2, spin) RUN(256, 1, false, 4, 2, spin) RUN(256, 1, true, 2, 2, spin) RUN(256, 1, true, 4, 2, spin)
    RUN(512, 1, false, 1, 1, spin) RUN(512, 1, false, 2, 1, spin) RUN(512, 1, false, 4, 1, spin) RUN(512, 1, true, 2, 1, spin) RUN(512, 1, true, 4, 1, spin)
    RUN(1024, 1, false, 1, 1, spin) RUN(1024, 1, false, 2, 1, spin) RUN(1024, 1, true, 1, 1, spin) RUN(1024, 1, true, 2, 1, spin)
    RUN(256, 1, true, 8, 4, spin) RUN(1024, 1, true, 2, 4, spin)
  }
  return 0;
}
