#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL, int BANK> __device__ __forceinline__ int dpp(int old, int src) {
  return __builtin_amdgcn_update_dpp(old, src, CTRL, 0xF, BANK, false);
}
__device__ __forceinline__ int x1(int v) { return dpp<0xB1, 0xF>(0, v); }
__device__ __forceinline__ int x2(int v) { return dpp<0x4E, 0xF>(0, v); }
__device__ __forceinline__ int x8(int v) { return dpp<0x128, 0xF>(0, v); }
__device__ __forceinline__ int x4(int v) { int r = dpp<0x124, 0xA>(0, v); return dpp<0x12C, 0x5>(r, v); }
__global__ void k(int* out) {
  int l = threadIdx.x;
  out[l] = x1(l); out[64 + l] = x2(l); out[128 + l] = x4(l); out[192 + l] = x8(l);
}
int main() {
  int* d; hipMalloc(&d, 256 * 4); k<<<1, 64>>>(d); int h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
  int masks[4] = {1, 2, 4, 8};
  for (int m = 0; m < 4; ++m) { int bad = 0; for (int l = 0; l < 64; ++l) bad += h[m * 64 + l] != (l ^ masks[m]); printf("xor %d: %s", masks[m], bad ? "WRONG:" : "ok"); if (bad) for (int l = 0; l < 16; ++l) printf(" %d", h[m * 64 + l]); printf("\n"); }
  return 0;
}
