// Diagnostic micro-benchmark (not part of the product): calibration of rocprofv3's FETCH_SIZE on gfx950 for the access
// shapes of this repository, on kernels whose byte counts are known ("calibrate FETCH_SIZE on a known copy").
// Every kernel reads a 1.9 GB buffer (well past the 256 MB Infinity Cache) exactly once per launch:
//   stream16     : linear 16 B / lane streaming read (the guide's x2 case)
//   rows_full128 : LK-like placement - a wave reads a 24-row window, rows one image pitch apart - but every row is a whole
//                  128-byte line (8 lanes x 16 B): all fetched bytes are used
//   rows_48of128 : the LK window shape: 3 lanes x 16 B = 48 bytes of each 128-byte line, windows on disjoint lines
//   rows_32of64  : 2 lanes x 16 B = 32 bytes of each 64-byte half line
// Run plain for the times, and under rocprofv3 --pmc FETCH_SIZE (own pass) for the counter; compare per kernel:
// requested bytes, 64 B sectors touched, 128 B lines touched, FETCH_SIZE, and the time against the stream16 time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void stream16(const uint4* __restrict__ s, size_t n, unsigned* __restrict__ sink) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { uint4 v = s[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) sink[0] = acc;
}

// one wave per window: ROWS rows, LANES16 lanes of 16 B per row, window w at line (w % lines_per_row), row band (w / lines_per_row)
template <int LANES16, int LINE>
__global__ __launch_bounds__(256) void rows_kernel(const unsigned char* __restrict__ img, int pitch, int nbands, int rows, unsigned* __restrict__ sink) {
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lines_per_row = pitch / LINE;
  const size_t total = (size_t)lines_per_row * nbands;
  unsigned acc = 0;
  for (size_t w = wave; w < total; w += (size_t)gridDim.x * 4) {
    const int band = (int)(w / lines_per_row), line = (int)(w - (size_t)band * lines_per_row);
    const unsigned char* base = img + ((size_t)band * rows) * pitch + (size_t)line * LINE;
    for (int i = lane; i < rows * LANES16; i += 64) {
      const int r = i / LANES16, k = i - r * LANES16;
      const uint4 v = *(const uint4*)(base + (size_t)r * pitch + 16 * k);
      acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <class F>
static float time_ms(F f) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f();
  CK(hipEventRecord(a));
  for (int i = 0; i < 3; i++) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / 3;
}

int main() {
  const int pitch = 8192, rows = 24, nbands = 9600;           // 8192 x 230400 bytes = 1.887 GB
  const size_t bytes = (size_t)pitch * rows * nbands;
  unsigned char* d; unsigned* sink;
  CK(hipMalloc(&d, bytes)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(d, 1, bytes));
  const double GB = 1e9;
  float t;
  t = time_ms([&] { hipLaunchKernelGGL(stream16, dim3(8192), dim3(256), 0, 0, (const uint4*)d, bytes / 16, sink); });
  printf("stream16      : requested %.3f GB  sectors64 %.3f GB  lines128 %.3f GB  time %.3f ms (%.2f TB/s of requested)\n", bytes / GB, bytes / GB, bytes / GB, t, bytes / (t * 1e-3) / 1e12);
  t = time_ms([&] { hipLaunchKernelGGL((rows_kernel<8, 128>), dim3(8192), dim3(256), 0, 0, d, pitch, nbands, rows, sink); });
  printf("rows_full128  : requested %.3f GB  sectors64 %.3f GB  lines128 %.3f GB  time %.3f ms (%.2f TB/s of requested)\n", bytes / GB, bytes / GB, bytes / GB, t, bytes / (t * 1e-3) / 1e12);
  t = time_ms([&] { hipLaunchKernelGGL((rows_kernel<3, 128>), dim3(8192), dim3(256), 0, 0, d, pitch, nbands, rows, sink); });
  printf("rows_48of128  : requested %.3f GB  sectors64 %.3f GB  lines128 %.3f GB  time %.3f ms (%.2f TB/s of requested)\n", bytes * 48.0 / 128 / GB, bytes * 64.0 / 128 / GB, bytes / GB, t, bytes * 48.0 / 128 / (t * 1e-3) / 1e12);
  t = time_ms([&] { hipLaunchKernelGGL((rows_kernel<2, 64>), dim3(8192), dim3(256), 0, 0, d, pitch, nbands, rows, sink); });
  printf("rows_32of64   : requested %.3f GB  sectors64 %.3f GB  lines128 %.3f GB  time %.3f ms (%.2f TB/s of requested)\n", bytes * 32.0 / 64 / GB, bytes / GB, bytes / GB, t, bytes * 32.0 / 64 / (t * 1e-3) / 1e12);
  return 0;
}
