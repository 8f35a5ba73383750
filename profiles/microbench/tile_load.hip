// Diagnostic micro-benchmark (not part of the product): cost structure of "dword tile -> LDS -> store" image kernels on
// MI355X.  Variants: tile rows, byte vs dword loads, XCD-aware block order, stores per thread.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct TG { int gx, gy, gz; };
template <int ROWS, int HALO, bool XCD, int MODE>
__global__ __launch_bounds__(256) void tile_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, int w, int h, int pitch,
                                                   size_t slot, TG g) {
  constexpr int SP = 72, SR = ROWS + 2 * HALO;
  __shared__ unsigned s[SR * SP / 4];
  unsigned n = (unsigned)g.gx * g.gy * g.gz, b = blockIdx.x, t = b;
  if (XCD) { unsigned per = (n + 7) / 8; t = (b & 7) * per + (b >> 3); }
  if (t >= n) return;
  unsigned row = t / g.gx;
  int bx = t - row * g.gx, bz = row / g.gy, by = row - bz * g.gy;
  const unsigned char* sp = src + (size_t)bz * slot;
  unsigned char* dp = dst + (size_t)bz * slot;
  int x0 = bx * 64, y0 = by * ROWS, tid = threadIdx.x;
  if (MODE == 0) {  // dword loads
    for (int i = tid; i < SR * (SP / 4); i += 256) {
      int ty = i / (SP / 4), k = i - ty * (SP / 4);
      int gy = min(max(y0 - HALO + ty, 0), h - 1), gx = min(max(x0 - 4 + 4 * k, 0), pitch - 4);
      s[i] = *(const unsigned*)(sp + (size_t)gy * pitch + gx);
    }
  } else {          // byte loads
    unsigned char* sb = (unsigned char*)s;
    for (int i = tid; i < SR * SP; i += 256) {
      int ty = i / SP, tx = i - ty * SP;
      int gy = min(max(y0 - HALO + ty, 0), h - 1), gx = min(max(x0 - 4 + tx, 0), w - 1);
      sb[i] = sp[(size_t)gy * pitch + gx];
    }
  }
  __syncthreads();
  // one dword store per thread per 16 rows
  int r = tid >> 4, c4 = (tid & 15) * 4;
  for (int q = 0; q < ROWS / 16; q++) {
    int y = y0 + r + 16 * q;
    unsigned v = s[(r + 16 * q + HALO) * (SP / 4) + 1 + (tid & 15)] + s[(r + 16 * q) * (SP / 4) + (tid & 15)];
    if (y < h && x0 + c4 < pitch) *(unsigned*)(dp + (size_t)y * pitch + x0 + c4) = v;
  }
}

template <int ROWS, int HALO, bool XCD, int MODE>
static void run(const char* name, const unsigned char* src, unsigned char* dst, int w, int h, int pitch, size_t slot, int B) {
  TG g{(w + 63) / 64, (h + ROWS - 1) / ROWS, B};
  unsigned n = (unsigned)g.gx * g.gy * g.gz, blocks = ((n + 7) / 8) * 8;
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int it = 0; it < 2; it++) hipLaunchKernelGGL((tile_kernel<ROWS, HALO, XCD, MODE>), dim3(blocks), dim3(256), 0, 0, src, dst, w, h, pitch, slot, g);
  CK(hipEventRecord(a));
  const int K = 5;
  for (int it = 0; it < K; it++) hipLaunchKernelGGL((tile_kernel<ROWS, HALO, XCD, MODE>), dim3(blocks), dim3(256), 0, 0, src, dst, w, h, pitch, slot, g);
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  double bytes = 2.0 * w * h * B;
  printf("%-44s rows %2d halo %d xcd %d mode %d : %8.1f us  (%6.2f TB/s read+write)\n", name, ROWS, HALO, (int)XCD, MODE, ms / K * 1e3, bytes / (ms / K * 1e-3) / 1e12);
}

__global__ void copy_kernel(const uint4* s, uint4* d, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) d[i] = s[i];
}

int main() {
  const int w = 1280, h = 720, pitch = 1280, B = 256;
  size_t slot = (size_t)pitch * h;
  unsigned char *src, *dst;
  CK(hipMalloc(&src, slot * B)); CK(hipMalloc(&dst, slot * B));
  CK(hipMemset(src, 1, slot * B)); CK(hipMemset(dst, 0, slot * B));
  {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(copy_kernel, dim3(16384), dim3(256), 0, 0, (const uint4*)src, (uint4*)dst, slot * B / 16);
    CK(hipEventRecord(a));
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL(copy_kernel, dim3(16384), dim3(256), 0, 0, (const uint4*)src, (uint4*)dst, slot * B / 16);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("linear copy: %8.1f us (%6.2f TB/s read+write)\n", ms / 5 * 1e3, 2.0 * slot * B / (ms / 5 * 1e-3) / 1e12);
  }
  run<16, 3, false, 0>("dword tile 16 rows", src, dst, w, h, pitch, slot, B);
  run<16, 3, true, 0>("dword tile 16 rows xcd", src, dst, w, h, pitch, slot, B);
  run<32, 3, false, 0>("dword tile 32 rows", src, dst, w, h, pitch, slot, B);
  run<32, 3, true, 0>("dword tile 32 rows xcd", src, dst, w, h, pitch, slot, B);
  run<16, 0, true, 0>("dword tile 16 rows no halo xcd", src, dst, w, h, pitch, slot, B);
  run<64, 3, true, 0>("dword tile 64 rows xcd", src, dst, w, h, pitch, slot, B);
  run<16, 4, false, 1>("byte tile 16 rows", src, dst, w, h, pitch, slot, B);
  run<16, 4, true, 1>("byte tile 16 rows xcd", src, dst, w, h, pitch, slot, B);
  return 0;
}
