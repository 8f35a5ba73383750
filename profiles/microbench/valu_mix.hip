// Diagnostic micro-benchmark (not part of the product): issue cost on gfx950 of the VALU instructions lk_track_kernel is
// made of, at 1 / 2 / 4 / 8 wavefronts per SIMD, as independent streams (8 chains) and as one dependent chain.
// Prints shader cycles per wave-instruction per SIMD (s_memtime deltas of every wave, averaged, times waves per SIMD
// divided by instructions issued per wave).  The LK kernel's roofline.valu_issue_frac in bench.py is priced with these.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

enum Op { ADD, DOT2C, PERM, ALIGNBYTE, MUL24, MAD24, DPP_ADD, DPP_ADD_NOP, READLANE, PK_ADD_U16, PK_MUL_U16, ASHR, ADD3, CVT_I2F, RNDNE, MUL_F32,
          LKPIX, NOPS };
static const char* kName[] = {"v_add_u32", "v_dot2c_i32_i16", "v_perm_b32", "v_alignbyte_b32", "v_mul_i32_i24", "v_mad_i32_i24",
                              "v_add_u32_dpp quad_perm", "v_add_u32_dpp + s_nop 1 (dependent form)", "v_readlane_b32", "v_pk_add_u16",
                              "v_pk_mul_lo_u16", "v_ashrrev_i32", "v_add3_u32", "v_cvt_f32_i32", "v_rndne_f32", "v_mul_f32",
                              "LK pixel: 2 dot2c + ashr + 2 mul24 + 2 add", "s_nop 0"};

#define ITER 512
#define UNR 8

template <int OP, int CH>
__global__ __launch_bounds__(256) void bench(unsigned* __restrict__ out, long long* __restrict__ cyc, unsigned seed) {
  unsigned x[8];
#pragma unroll
  for (int i = 0; i < 8; i++) x[i] = seed * (threadIdx.x + 1) + i * 0x9E3779B9u;
  unsigned y = seed ^ 0x01020304u, sel = 0x05040100u, sh = threadIdx.x & 3;
  unsigned sg = 0;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int u = 0; u < UNR; u++) {
#pragma unroll
      for (int c = 0; c < CH; c++) {
        unsigned& v = x[c];
        if (OP == ADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v) : "v"(y));
        if (OP == DOT2C) asm volatile("v_dot2c_i32_i16 %0, %1, %2" : "+v"(v) : "v"(y), "v"(sel));
        if (OP == PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(v) : "v"(y), "v"(sel));
        if (OP == ALIGNBYTE) asm volatile("v_alignbyte_b32 %0, %0, %1, %2" : "+v"(v) : "v"(y), "v"(sh));
        if (OP == MUL24) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(v) : "v"(y));
        if (OP == MAD24) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(v) : "v"(y), "v"(sel));
        if (OP == DPP_ADD) asm volatile("v_add_u32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(v) : "v"(y));
        if (OP == DPP_ADD_NOP) asm volatile("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(v));
        if (OP == READLANE) { unsigned s; asm volatile("v_readlane_b32 %0, %1, 63" : "=s"(s) : "v"(v)); sg += s; }
        if (OP == PK_ADD_U16) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(v) : "v"(y));
        if (OP == PK_MUL_U16) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(v) : "v"(y));
        if (OP == ASHR) asm volatile("v_ashrrev_i32 %0, 9, %0" : "+v"(v));
        if (OP == ADD3) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(v) : "v"(y), "v"(sel));
        if (OP == CVT_I2F) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(v));
        if (OP == RNDNE) asm volatile("v_rndne_f32 %0, %0" : "+v"(v));
        if (OP == MUL_F32) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v) : "v"(y));
        if (OP == LKPIX) {
          unsigned a, b, d;
          asm volatile("v_mov_b32 %0, %3\n\tv_dot2c_i32_i16 %0, %4, %5\n\tv_dot2c_i32_i16 %0, %5, %4\n\ts_nop 0\n\tv_ashrrev_i32 %0, 9, %0\n\t"
                       "v_mul_i32_i24 %1, %0, %4\n\tv_mul_i32_i24 %2, %0, %5\n\tv_add_u32 %3, %3, %1\n\tv_add_u32 %3, %3, %2"
                       : "=&v"(a), "=&v"(b), "=&v"(d), "+v"(v) : "v"(y), "v"(sel));
        }
        if (OP == NOPS) asm volatile("s_nop 0");
      }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  unsigned r = sg;
#pragma unroll
  for (int i = 0; i < 8; i++) r ^= x[i];
  out[blockIdx.x * 256 + threadIdx.x] = r;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP, int CH>
static void run(unsigned* d_out, long long* d_cyc, std::vector<long long>& h) {
  const int per_it = (OP == LKPIX ? 8 : 1);   // instructions issued per chain step (LKPIX: 8 VALU + 1 s_nop)
  printf("%-44s chains %d :", kName[OP], CH);
  for (int wps : {1, 2, 4, 8}) {
    const int blocks = 256 * wps;   // 256 CUs x wps workgroups of 4 wavefronts (one per SIMD)
    hipLaunchKernelGGL((bench<OP, CH>), dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, 12345u);
    hipEvent_t ea, eb;
    CK(hipEventCreate(&ea)); CK(hipEventCreate(&eb));
    CK(hipEventRecord(ea));
    hipLaunchKernelGGL((bench<OP, CH>), dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, 12345u);
    CK(hipEventRecord(eb));
    CK(hipDeviceSynchronize());
    float wall_ms = 0;
    CK(hipEventElapsedTime(&wall_ms, ea, eb));
    CK(hipMemcpy(h.data(), d_cyc, (size_t)blocks * 4 * sizeof(long long), hipMemcpyDeviceToHost));
    double s = 0;
    for (int i = 0; i < blocks * 4; i++) s += (double)h[i];
    const double per_wave = s / (blocks * 4);
    const double ninst = (double)ITER * UNR * CH * per_it;
    // all wps waves of a SIMD run concurrently for about per_wave cycles and issue wps * ninst instructions
    // wall clock: every SIMD issues wps * ninst instructions during the launch -> ns per wave-instruction per SIMD
    printf("  %dw %6.2f (SIMD %5.2f tick, %5.2f ns)", wps, per_wave / ninst, per_wave / ninst / wps, wall_ms * 1e6 / (ninst * wps));
  }
  printf("\n");
}

int main() {
  unsigned* d_out; long long* d_cyc;
  CK(hipMalloc(&d_out, 256 * 8 * 256 * sizeof(unsigned)));
  CK(hipMalloc(&d_cyc, 256 * 8 * 4 * sizeof(long long)));
  std::vector<long long> h(256 * 8 * 4);
  printf("cycles per instruction as one wave sees them, and (SIMD) the same divided by the waves sharing the SIMD = issue cost per wave-instruction\n");
#define RUN(OP) run<OP, 8>(d_out, d_cyc, h); run<OP, 1>(d_out, d_cyc, h);
  RUN(ADD) RUN(DOT2C) RUN(PERM) RUN(ALIGNBYTE) RUN(MUL24) RUN(MAD24) RUN(DPP_ADD) RUN(DPP_ADD_NOP) RUN(READLANE) RUN(PK_ADD_U16) RUN(PK_MUL_U16)
  RUN(ASHR) RUN(ADD3) RUN(CVT_I2F) RUN(RNDNE) RUN(MUL_F32) RUN(LKPIX) RUN(NOPS)
  return 0;
}
