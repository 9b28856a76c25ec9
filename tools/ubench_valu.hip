// tools/ubench_valu.hip — gfx950 VALU issue-rate microbenchmark for the integer
// instructions a 64-bit modular butterfly is made of.  Diagnostic only (not part
// of the product); results are recorded in DESIGN.md.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_valu ubench_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>
typedef unsigned long long u64;
typedef unsigned int u32;

#define CHECK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

// Each kernel: 8 independent chains, 8 instr per chain-iteration body => 64 instr/iter
#define DEF_KERNEL32(NAME, ASM)                                                  \
__global__ void NAME(u32* out, u32 a, u32 b, int iters){                          \
  u32 r0=threadIdx.x, r1=r0+1, r2=r0+2, r3=r0+3, r4=r0+4, r5=r0+5, r6=r0+6, r7=r0+7; \
  u32 s0=a, s1=b;                                                                 \
  for(int i=0;i<iters;i++){                                                       \
    _Pragma("unroll") for(int u=0;u<8;u++){                                       \
      asm volatile(ASM : "+v"(r0) : "v"(s0), "v"(s1));                            \
      asm volatile(ASM : "+v"(r1) : "v"(s0), "v"(s1));                            \
      asm volatile(ASM : "+v"(r2) : "v"(s0), "v"(s1));                            \
      asm volatile(ASM : "+v"(r3) : "v"(s0), "v"(s1));                            \
      asm volatile(ASM : "+v"(r4) : "v"(s0), "v"(s1));                            \
      asm volatile(ASM : "+v"(r5) : "v"(s0), "v"(s1));                            \
      asm volatile(ASM : "+v"(r6) : "v"(s0), "v"(s1));                            \
      asm volatile(ASM : "+v"(r7) : "v"(s0), "v"(s1));                            \
    }                                                                             \
  }                                                                               \
  out[blockIdx.x*blockDim.x+threadIdx.x]=r0^r1^r2^r3^r4^r5^r6^r7;                 \
}

DEF_KERNEL32(k_mul_lo,   "v_mul_lo_u32 %0, %0, %1")
DEF_KERNEL32(k_mul_hi,   "v_mul_hi_u32 %0, %0, %1")
DEF_KERNEL32(k_add,      "v_add_u32 %0, %0, %1")
DEF_KERNEL32(k_add3,     "v_add3_u32 %0, %0, %1, %2")
DEF_KERNEL32(k_mul24,    "v_mul_u32_u24 %0, %0, %1")
DEF_KERNEL32(k_mulhi24,  "v_mul_hi_u32_u24 %0, %0, %1")
DEF_KERNEL32(k_mad24,    "v_mad_u32_u24 %0, %0, %1, %2")
DEF_KERNEL32(k_alignbit, "v_alignbit_b32 %0, %0, %1, 11")
DEF_KERNEL32(k_xor,      "v_xor_b32 %0, %0, %1")
DEF_KERNEL32(k_fma32,    "v_fma_f32 %0, %0, %1, %2")
DEF_KERNEL32(k_addco,    "v_add_co_u32 %0, vcc, %0, %1")
DEF_KERNEL32(k_cndmask,  "v_cndmask_b32 %0, %0, %1, vcc")
DEF_KERNEL32(k_mad_u32_u16, "v_mad_u32_u16 %0, %0, %1, %2")
DEF_KERNEL32(k_mullo16,  "v_mul_lo_u16 %0, %0, %1")
DEF_KERNEL32(k_dot4,     "v_dot4_u32_u8 %0, %0, %1, %2")

#define DEF_KERNEL64(NAME, ASM)                                                  \
__global__ void NAME(u32* out, u32 a, u32 b, int iters){                          \
  u64 r0=threadIdx.x, r1=r0+1, r2=r0+2, r3=r0+3, r4=r0+4, r5=r0+5, r6=r0+6, r7=r0+7; \
  u32 s0=a, s1=b; u64 t0=((u64)a<<32)|b;                                          \
  for(int i=0;i<iters;i++){                                                       \
    _Pragma("unroll") for(int u=0;u<8;u++){                                       \
      asm volatile(ASM : "+v"(r0) : "v"(s0), "v"(s1), "v"(t0));                   \
      asm volatile(ASM : "+v"(r1) : "v"(s0), "v"(s1), "v"(t0));                   \
      asm volatile(ASM : "+v"(r2) : "v"(s0), "v"(s1), "v"(t0));                   \
      asm volatile(ASM : "+v"(r3) : "v"(s0), "v"(s1), "v"(t0));                   \
      asm volatile(ASM : "+v"(r4) : "v"(s0), "v"(s1), "v"(t0));                   \
      asm volatile(ASM : "+v"(r5) : "v"(s0), "v"(s1), "v"(t0));                   \
      asm volatile(ASM : "+v"(r6) : "v"(s0), "v"(s1), "v"(t0));                   \
      asm volatile(ASM : "+v"(r7) : "v"(s0), "v"(s1), "v"(t0));                   \
    }                                                                             \
  }                                                                               \
  out[blockIdx.x*blockDim.x+threadIdx.x]=(u32)(r0^r1^r2^r3^r4^r5^r6^r7);          \
}
DEF_KERNEL64(k_mad64,     "v_mad_u64_u32 %0, vcc, %1, %2, %0")
DEF_KERNEL64(k_lshl_add64,"v_lshl_add_u64 %0, %0, 0, %3")
DEF_KERNEL64(k_lshl64,    "v_lshlrev_b64 %0, 3, %0")
DEF_KERNEL64(k_fma64,     "v_fma_f64 %0, %0, %3, %3")
DEF_KERNEL64(k_mul64f,    "v_mul_f64 %0, %0, %3")
DEF_KERNEL64(k_add64f,    "v_add_f64 %0, %0, %3")
DEF_KERNEL64(k_cmp64,     "v_cmp_le_u64 vcc, %0, %3")

// compiled butterflies (what hipcc makes of the C code), ILP = 4 butterflies per thread
__device__ __forceinline__ u64 mulhi_approx(u64 a, u64 b){
  u32 a0=(u32)a, a1=(u32)(a>>32), b0=(u32)b, b1=(u32)(b>>32);
  u64 hh = (u64)a1*b1;
  u32 c1 = __umulhi(a0,b1);
  u32 c2 = __umulhi(a1,b0);
  return hh + c1 + c2;
}
template<int MODE>
__global__ void k_bfly(u64* p, u64 q, u64 w, u64 wp, int iters){
  u64 x[4], y[4];
  #pragma unroll
  for(int k=0;k<4;k++){ x[k]=p[threadIdx.x+64*k]%q; y[k]=p[threadIdx.x+64*k+256]%q; }
  const u64 q2 = 2*q, q4 = 4*q;
  for(int i=0;i<iters;i++){
    #pragma unroll
    for(int k=0;k<4;k++){
      if(MODE==0){ // Harvey exact Shoup, values in [0,4q)
        u64 xr = x[k] >= q2 ? x[k]-q2 : x[k];
        u64 qh = __umul64hi(y[k], wp);
        u64 t = y[k]*w - qh*q;
        x[k] = xr + t; y[k] = xr - t + q2;
      } else if(MODE==1){ // approx Shoup, values in [0,8q), q<2^61
        u64 xr = x[k] >= q4 ? x[k]-q4 : x[k];
        u64 qh = mulhi_approx(y[k], wp);
        u64 t = y[k]*w - qh*q;
        x[k] = xr + t; y[k] = xr - t + q4;
      } else if(MODE==2){ // reference-style: u128 % (what a naive GPU port would do)
        unsigned __int128 pr = (unsigned __int128)y[k]*w;
        u64 t = (u64)(pr % q);
        u64 s = x[k]+t; if(s>=q) s-=q;
        u64 d = x[k]>=t ? x[k]-t : q+x[k]-t;
        x[k]=s; y[k]=d;
      } else if(MODE==3){ // approx Shoup with q = 2^61-2^21+1 folded as shifts
        u64 xr = x[k] >= q4 ? x[k]-q4 : x[k];
        u64 qh = mulhi_approx(y[k], wp);
        u64 t = y[k]*w - ((qh<<61) - (qh<<21) + qh);
        x[k] = xr + t; y[k] = xr - t + q4;
      }
    }
  }
  #pragma unroll
  for(int k=0;k<4;k++){ p[threadIdx.x+64*k]=x[k]; p[threadIdx.x+64*k+256]=y[k]; }
}

template<typename F>
static float time_kernel(F launch, int reps=3){
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  float best=1e30f;
  for(int r=0;r<reps;r++){
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms,e0,e1); if(ms<best) best=ms;
  }
  hipEventDestroy(e0); hipEventDestroy(e1);
  return best;
}

int main(){
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop,0));
  int cus = prop.multiProcessorCount; double clk = prop.clockRate*1e3;
  printf("device %s CUs=%d clock=%.0f MHz\n", prop.name, cus, clk/1e6);
  u32* out; CHECK(hipMalloc(&out, (size_t)cus*32*1024*4));
  u64* p; CHECK(hipMalloc(&p, 4096*8)); CHECK(hipMemset(p, 0x5a, 4096*8));
  const int iters=2000;
  struct K { const char* name; void(*fn)(u32*,u32,u32,int); };
  K ks[] = {
    {"v_mul_lo_u32",k_mul_lo},{"v_mul_hi_u32",k_mul_hi},{"v_mad_u64_u32",k_mad64},
    {"v_add_u32",k_add},{"v_add3_u32",k_add3},{"v_add_co_u32",k_addco},{"v_cndmask_b32",k_cndmask},
    {"v_lshl_add_u64",k_lshl_add64},{"v_lshlrev_b64",k_lshl64},{"v_cmp_le_u64",k_cmp64},
    {"v_alignbit_b32",k_alignbit},{"v_xor_b32",k_xor},
    {"v_mul_u32_u24",k_mul24},{"v_mul_hi_u32_u24",k_mulhi24},{"v_mad_u32_u24",k_mad24},
    {"v_mad_u32_u16",k_mad_u32_u16},{"v_mul_lo_u16",k_mullo16},{"v_dot4_u32_u8",k_dot4},
    {"v_fma_f32",k_fma32},{"v_fma_f64",k_fma64},{"v_mul_f64",k_mul64f},{"v_add_f64",k_add64f},
  };
  for(int wpS : {1,2,4}){  // waves per SIMD
    printf("--- %d wave(s) per SIMD (block=%d threads, 1 block/CU) ---\n", wpS, 256*wpS);
    for(auto& k: ks){
      int threads=256*wpS; int blocks=cus;
      float ms=time_kernel([&]{ hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(threads), 0, 0, out, 12345u, 6789u, iters); });
      double instr_per_wave = (double)iters*64;
      // cycles per instr per SIMD (issue): time*clk / (instr_per_wave * waves_per_simd)
      double cyc = ms*1e-3*clk/(instr_per_wave*wpS);
      printf("%-20s %8.3f ms  %6.2f cyc/instr/SIMD (at nominal clock)\n", k.name, ms, cyc);
    }
  }
  // butterflies
  u64 q=2305843009211596801ull, w=1681162619342215248ull; u64 wp=(u64)((((unsigned __int128)w)<<64)/q);
  const char* names[]={"bfly exact-shoup","bfly approx-shoup","bfly u128 %","bfly approx+solinas-q"};
  for(int wpS : {1,2,4,8}){
    for(int mode=0;mode<4;mode++){
      int threads=64*4; // 4 waves per block; blocks per CU = wpS
      int blocks=cus*wpS; int it = mode==2? 200: 2000;
      float ms;
      if(mode==0) ms=time_kernel([&]{ hipLaunchKernelGGL(k_bfly<0>, dim3(blocks), dim3(threads),0,0,p,q,w,wp,it); });
      else if(mode==1) ms=time_kernel([&]{ hipLaunchKernelGGL(k_bfly<1>, dim3(blocks), dim3(threads),0,0,p,q,w,wp,it); });
      else if(mode==2) ms=time_kernel([&]{ hipLaunchKernelGGL(k_bfly<2>, dim3(blocks), dim3(threads),0,0,p,q,w,wp,it); });
      else ms=time_kernel([&]{ hipLaunchKernelGGL(k_bfly<3>, dim3(blocks), dim3(threads),0,0,p,q,w,wp,it); });
      double bf = (double)blocks*threads*4*it;
      double cyc = ms*1e-3*clk/((double)it*4*wpS);
      printf("%-24s wpS=%d %8.3f ms  %8.1f Gbfly/s  %6.1f cyc/bfly-wave/SIMD\n", names[mode], wpS, ms, bf/ms*1e-6, cyc);
    }
  }
  hipFree(out); hipFree(p);
  return 0;
}
