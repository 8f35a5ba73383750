// Diagnostic micro-benchmark (not part of the product): LDS read cost by access width / alignment on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
// MODE 0: u8, lane-consecutive bytes; 1: u16 aligned (2*lane); 2: u16 at odd address (2*lane+1); 3: u16 at 5*lane (mixed);
//      4: b32 aligned (4*lane); 5: u8 stride 5 bytes; 6: b32 stride 5 dwords
template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters) {
  __shared__ unsigned s[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) s[i] = i * 2654435761u;
  __syncthreads();
  const unsigned char* b = (const unsigned char*)s;
  int lane = threadIdx.x;
  unsigned acc = 0;
  int off = 0;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      int o = (off + u * 64) & 8191;
      if (MODE == 0) acc += b[o + lane];
      if (MODE == 1) acc += *(const unsigned short*)(b + o + 2 * lane);
      if (MODE == 2) acc += *(const unsigned short*)(b + o + 2 * lane + 1);
      if (MODE == 3) acc += *(const unsigned short*)(b + o + 5 * lane);
      if (MODE == 4) acc += *(const unsigned*)(b + o + 4 * lane);
      if (MODE == 5) acc += b[o + 5 * lane];
      if (MODE == 6) acc += *(const unsigned*)(b + ((o + 20 * lane) & 16380));
    }
    off += acc & 4;  // keeps the loads in the loop
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int MODE>
static void run(const char* name, unsigned* out) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int iters = 2000, blocks = 256 * 8;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  // wave-instructions per CU: blocks/256 CUs * 4 waves * iters * 8
  double winstr = (double)blocks / 256 * 4 * iters * 8;
  printf("%-34s %8.1f us   %6.1f cycles per wave-instruction per CU (2.4 GHz)\n", name, ms * 1e3, ms * 1e-3 * 2.4e9 / winstr);
}
int main() {
  unsigned* out; CK(hipMalloc(&out, 256 * 8 * 256 * 4));
  run<4>("b32 aligned, consecutive", out);
  run<6>("b32 aligned, stride 5 dwords", out);
  run<0>("u8 consecutive bytes", out);
  run<5>("u8 stride 5 bytes", out);
  run<1>("u16 aligned consecutive", out);
  run<2>("u16 odd address", out);
  run<3>("u16 stride 5 bytes (mixed alignment)", out);
  return 0;
}
