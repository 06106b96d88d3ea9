// Microbenchmark: integer / fp64 VALU issue rates on gfx950, to size the 64-bit modmul.
// Build: hipcc -O3 --offload-arch=gfx950 tools/mulrate.hip -o tools/mulrate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

constexpr int ITERS = 4096;
constexpr int CH = 8; // independent chains per thread

template<int OP> __global__ void __launch_bounds__(256) k(uint64_t* out, uint64_t seed) {
  uint64_t a[CH]; 
  uint32_t tid = blockIdx.x*blockDim.x+threadIdx.x;
  for (int c=0;c<CH;++c) a[c] = seed*(tid+1) + c*0x9e3779b97f4a7c15ull;
  uint32_t m = (uint32_t)seed | 1u;
  double dm = (double)(seed&0xffff)*1e-9+1.0000001;
  for (int i=0;i<ITERS;++i) {
#pragma unroll
    for (int c=0;c<CH;++c) {
      if constexpr (OP==0) { // v_mad_u64_u32
        uint64_t r; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[c]) : "v"((uint32_t)a[c]), "v"(m) : "vcc");
      } else if constexpr (OP==1) { // v_mul_lo_u32
        uint32_t x=(uint32_t)a[c]; asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(m)); a[c]=x;
      } else if constexpr (OP==2) { // v_mul_hi_u32
        uint32_t x=(uint32_t)a[c]; asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x) : "v"(m)); a[c]=x|1;
      } else if constexpr (OP==3) { // v_fma_f64
        double d = __longlong_as_double(a[c]); asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d) : "v"(dm)); a[c]=__double_as_longlong(d);
      } else if constexpr (OP==4) { // 32-bit add
        uint32_t x=(uint32_t)a[c]; asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(m)); a[c]=x;
      } else if constexpr (OP==5) { // v_mul_u32_u24
        uint32_t x=(uint32_t)a[c]; asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(m)); a[c]=x;
      } else if constexpr (OP==6) { // 64-bit add (2 insts)
        a[c] += seed;
      } else if constexpr (OP==7) { // v_mad_u32_u24
        uint32_t x=(uint32_t)a[c]; asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(x) : "v"(m)); a[c]=x;
      } else if constexpr (OP==8) { // v_mul_hi_u32_u24
        uint32_t x=(uint32_t)a[c]; asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(x) : "v"(m)); a[c]=x|0x10001;
      } else if constexpr (OP==9) { // v_mad_u64_u32 with sgpr multiplier
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[c]) : "v"((uint32_t)a[c]), "s"(m) : "vcc");
      } else if constexpr (OP==10) { // v_mul_f64
        double d = __longlong_as_double(a[c]); asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d) : "v"(dm)); a[c]=__double_as_longlong(d);
      } else if constexpr (OP==11) { // v_lshlrev_b64
        asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(a[c]));
      } else if constexpr (OP==12) { // v_cmp_lt_u64 + cndmask x2
        uint64_t t = a[c] - seed; a[c] = (a[c] >= seed) ? t : a[c]; a[c] += m;
      }
    }
  }
  uint64_t s=0; for (int c=0;c<CH;++c) s^=a[c];
  out[tid]=s;
}

template<int OP> int run(const char* name, int instr_per_iter) {
  int blocks = 256*8, threads=256; // 8 blocks/CU x 4 waves = 32 waves/CU
  uint64_t* d; CK(hipMalloc(&d, (size_t)blocks*threads*8));
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 0x123456789abcdefull);
  CK(hipDeviceSynchronize());
  float best=1e30f;
  for (int r=0;r<3;++r){
    CK(hipEventRecord(e0)); 
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(threads), 0, 0, d, 0x123456789abcdefull+r);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms,e0,e1)); if(ms<best)best=ms;
  }
  double laneops = (double)blocks*threads*ITERS*CH*instr_per_iter;
  double rate = laneops/(best*1e-3);
  // cycles per wave-instruction per SIMD at 2.4 GHz: 1024 SIMDs
  double waveinstr = laneops/64.0;
  double cyc = (best*1e-3*2.4e9)*1024.0/waveinstr;
  printf("%-28s %8.3f ms  %8.2f Tlaneop/s  ~%5.2f cyc/wave-instr/SIMD (@2.4GHz)\n", name, best, rate*1e-12, cyc);
  CK(hipFree(d));
  return 0;
}

int main(){
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p,0));
  printf("device %s CUs=%d clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  run<4>("v_add_u32",1);
  run<0>("v_mad_u64_u32",1);
  run<9>("v_mad_u64_u32 (sgpr)",1);
  run<1>("v_mul_lo_u32",1);
  run<2>("v_mul_hi_u32",1);
  run<5>("v_mul_u32_u24",1);
  run<7>("v_mad_u32_u24",1);
  run<8>("v_mul_hi_u32_u24",1);
  run<3>("v_fma_f64",1);
  run<10>("v_mul_f64",1);
  run<6>("add64 (2 inst)",2);
  run<11>("v_lshlrev_b64",1);
  run<12>("condsub64+add",1);
  return 0;
}
