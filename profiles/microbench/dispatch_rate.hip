// Diagnostic micro-benchmark (not part of the product): what does a workgroup that exits at once cost on MI355X?
// Decides between "launch the worst-case grid and let surplus workgroups return" and grid-stride work loops for the
// device-driven (no host count) launches of the frame-batch step.
//   early_exit   : grid of N workgroups, each reads a device count and returns (all surplus)
//   stride_loop  : fixed grid (CUs x k), each workgroup loops over `work` items of a trivial body
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int LDS_BYTES>
__global__ void early_exit(const int* __restrict__ n, int* __restrict__ sink) {
  __shared__ unsigned s[LDS_BYTES / 4 > 0 ? LDS_BYTES / 4 : 1];
  if ((int)blockIdx.x >= *n) return;
  s[threadIdx.x & 7] = threadIdx.x;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(sink, (int)s[3]);
}

__global__ void stride_loop(const int* __restrict__ n, int* __restrict__ sink) {
  const int work = *n;
  int acc = 0;
  for (int b = blockIdx.x; b < work; b += gridDim.x) { acc += b; __syncthreads(); }
  if (threadIdx.x == 0 && acc == 0x7fffffff) atomicAdd(sink, acc);
}

template <class F>
static float time_us(F f, int reps) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); f();
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; i++) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms * 1e3f / reps;
}

int main() {
  int *d_n, *d_sink;
  CK(hipMalloc(&d_n, 4)); CK(hipMalloc(&d_sink, 4));
  CK(hipMemset(d_n, 0, 4)); CK(hipMemset(d_sink, 0, 4));
  const int grids[] = {1024, 16384, 65536, 262144, 1048576, 4194304};
  for (int threads : {64, 256}) {
    for (int g : grids) {
      float t0 = time_us([&] { hipLaunchKernelGGL(early_exit<0>, dim3(g), dim3(threads), 0, 0, d_n, d_sink); }, 10);
      float t1 = time_us([&] { hipLaunchKernelGGL(early_exit<16384>, dim3(g), dim3(threads), 0, 0, d_n, d_sink); }, 10);
      printf("early_exit threads %3d grid %8d : no-LDS %9.1f us (%6.2f ns/wg)   16KB-LDS %9.1f us (%6.2f ns/wg)\n", threads, g, t0,
             t0 * 1e3 / g, t1, t1 * 1e3 / g);
    }
  }
  // a launch whose grid is sized on the host to a handful of workgroups, for reference
  float tl = time_us([&] { hipLaunchKernelGGL(early_exit<0>, dim3(8), dim3(256), 0, 0, d_n, d_sink); }, 50);
  printf("minimal launch (8 workgroups): %.2f us per launch back to back\n", tl);
  int h = 1 << 20;
  CK(hipMemcpy(d_n, &h, 4, hipMemcpyHostToDevice));
  for (int g : {256, 1024, 2048, 4096}) {
    float t = time_us([&] { hipLaunchKernelGGL(stride_loop, dim3(g), dim3(256), 0, 0, d_n, d_sink); }, 10);
    printf("stride_loop grid %5d over 1M items: %9.1f us (%6.2f ns/item)\n", g, t, t * 1e3 / h);
  }
  return 0;
}
