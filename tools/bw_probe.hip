// Bandwidth probe for the tiled propagator design: how fast can a 2-read/2-write pass over 16 MiB complex128
// vectors run when the vectors are Infinity-Cache resident, in the contiguous ("A") and the strided-run ("B") layout?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)

// each WG handles one tile of TILE double2 elements. layout A: contiguous. layout B: RUN-element runs with stride.
template<int NIN, int NOUT, bool LAYOUT_B>
__global__ __launch_bounds__(256) void k_pass(const double2* __restrict__ a, const double2* __restrict__ b,
    double2* __restrict__ c, double2* __restrict__ d, int tile_log2, int n_log2) {
  const unsigned tile = 1u << tile_log2;
  const unsigned t = blockIdx.x;
  for (unsigned i = threadIdx.x; i < tile; i += 256) {
    size_t idx;
    if (!LAYOUT_B) idx = (size_t)t * tile + i;
    else { // low 4 bits contiguous (16 elements = 256 B), tile index in bits 4..(4+mid-1), high bits from i>>4
      const unsigned mid_bits = n_log2 - tile_log2; // bits fixed by the tile index, placed at bit 4
      idx = (i & 15u) | ((size_t)t << 4) | ((size_t)(i >> 4) << (4 + mid_bits));
    }
    double2 x = a[idx];
    if (NIN > 1) { double2 y = b[idx]; x.x += y.x; x.y += y.y; }
    c[idx] = x;
    if (NOUT > 1) { d[idx] = make_double2(x.y, x.x); }
  }
}

template<int NIN, int NOUT, bool LB>
float run(double2* bufs[4], int n_log2, int tile_log2, int iters, hipStream_t s) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  unsigned ntiles = 1u << (n_log2 - tile_log2);
  for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k_pass<NIN,NOUT,LB>), dim3(ntiles), dim3(256), 0, s, bufs[0], bufs[1], bufs[2], bufs[3], tile_log2, n_log2);
  hipEventRecord(e0, s);
  for (int it = 0; it < iters; ++it) {
    if (it & 1) hipLaunchKernelGGL((k_pass<NIN,NOUT,LB>), dim3(ntiles), dim3(256), 0, s, bufs[2], bufs[3], bufs[0], bufs[1], tile_log2, n_log2);
    else hipLaunchKernelGGL((k_pass<NIN,NOUT,LB>), dim3(ntiles), dim3(256), 0, s, bufs[0], bufs[1], bufs[2], bufs[3], tile_log2, n_log2);
  }
  hipEventRecord(e1, s); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / iters;
}

int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  for (int n_log2 : {20, 24}) {
    size_t n = (size_t)1 << n_log2;
    double2* bufs[4];
    for (int i = 0; i < 4; ++i) { CK(hipMalloc(&bufs[i], n * sizeof(double2))); CK(hipMemset(bufs[i], 0, n * sizeof(double2))); }
    int iters = n_log2 == 20 ? 400 : 40;
    double mb = n * 16.0 / 1e6;
    for (int tl : {10, 11, 12, 13}) {
      float t11a = run<1,1,false>(bufs, n_log2, tl, iters, s);
      float t22a = run<2,2,false>(bufs, n_log2, tl, iters, s);
      float t22b = run<2,2,true>(bufs, n_log2, tl, iters, s);
      float t11b = run<1,1,true>(bufs, n_log2, tl, iters, s);
      printf("N=%d tile=2^%d: 1R1W A %.2f us (%.0f GB/s) | 2R2W A %.2f us (%.0f GB/s) | 2R2W B %.2f us (%.0f GB/s) | 1R1W B %.2f us (%.0f GB/s)\n",
        n_log2, tl, t11a, 2*mb/t11a*1e3, t22a, 4*mb/t22a*1e3, t22b, 4*mb/t22b*1e3, t11b, 2*mb/t11b*1e3);
    }
    for (int i = 0; i < 4; ++i) hipFree(bufs[i]);
  }
  // empty-kernel launch cadence
  {
    double2* b[4]; for (int i=0;i<4;++i) CK(hipMalloc(&b[i], 4096*16));
    float t = run<1,1,false>(b, 8, 8, 2000, s);
    printf("tiny kernel back-to-back: %.2f us per launch\n", t);
  }
  return 0;
}
