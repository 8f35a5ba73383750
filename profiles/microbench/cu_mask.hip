// Does hipExtStreamCreateWithCUMask confine a stream's kernels to the masked CUs on this platform, and which mask bit is which
// (XCD, CU)?  Every workgroup records the XCC id and the HW_ID (SE / CU) it ran on.
//   hipcc --offload-arch=gfx950 -O3 cu_mask.hip -o cu_mask && ./cu_mask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void where_kernel(unsigned* out, int spin) {
  unsigned hwid, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) {}
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hwid; out[2 * blockIdx.x + 1] = xcc; }
}

static int census(hipStream_t st, unsigned* d, int nblk, const char* name) {
  std::vector<unsigned> h(2 * nblk);
  hipLaunchKernelGGL(where_kernel, dim3(nblk), dim3(64), 0, st, d, 2000);
  if (hipStreamSynchronize(st) != hipSuccess) { printf("%s: launch failed\n", name); return -1; }
  (void)hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  std::set<unsigned> cus, xccs;
  for (int b = 0; b < nblk; b++) {
    const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xF;
    const unsigned cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
    cus.insert(xcc << 16 | se << 8 | sh << 4 | cu);
    xccs.insert(xcc);
  }
  printf("%-28s distinct (xcc, se, sh, cu) = %zu over %zu XCCs\n", name, cus.size(), xccs.size());
  return (int)cus.size();
}

int main() {
  unsigned* d;
  const int nblk = 8192;
  CK(hipMalloc(&d, 2 * nblk * 4));
  hipStream_t s0;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  census(s0, d, nblk, "no mask");
  for (int keep : {192, 128, 64, 32}) {
    // mask words: bit i of the 256-bit mask; try "first `keep` bits" and "every XCD keeps keep/8 CUs" (bit = cu * 8 + xcd)
    for (int mode = 0; mode < 2; mode++) {
      unsigned mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int i = 0; i < 256; i++) {
        bool on = mode == 0 ? i < keep : (i / 8) < keep / 8;
        if (on) mask[i / 32] |= 1u << (i % 32);
      }
      hipStream_t s;
      hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask);
      if (e != hipSuccess) { printf("hipExtStreamCreateWithCUMask(%d, mode %d): %s\n", keep, mode, hipGetErrorString(e)); continue; }
      char name[64];
      snprintf(name, sizeof name, "mask %d CUs, mode %d", keep, mode);
      census(s, d, nblk, name);
      (void)hipStreamDestroy(s);
    }
  }
  return 0;
}
