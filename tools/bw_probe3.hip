// Probe 3: does sub-tile pipelining inside one workgroup (loads of all sub-tiles issued up front, each sub-tile
// barrier'd + stored on its own) recover the read/write overlap that a single load-all / barrier / store-all tile loses?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)

template<int NT, int S>   // S sub-tiles, each thread owns one element per sub-tile per vector
__global__ __launch_bounds__(NT) void k_pipe(const double2* __restrict__ a, const double2* __restrict__ b, double2* __restrict__ c, double2* __restrict__ d) {
  __shared__ double2 sh[NT];
  const size_t base = (size_t)blockIdx.x * NT * S;
  double2 x[S], y[S];
#pragma unroll
  for (int k = 0; k < S; ++k) { x[k] = a[base + k*NT + threadIdx.x]; y[k] = b[base + k*NT + threadIdx.x]; }
#pragma unroll
  for (int k = 0; k < S; ++k) {
    sh[threadIdx.x] = x[k];
    __syncthreads();
    double2 p = sh[threadIdx.x ^ 1];
    x[k].x += 1e-30 * p.y;
    c[base + k*NT + threadIdx.x] = x[k];
    __syncthreads();
    sh[threadIdx.x] = x[k];
    __syncthreads();
    p = sh[threadIdx.x ^ 2];
    y[k].x += 1e-30 * p.x;
    d[base + k*NT + threadIdx.x] = y[k];
    __syncthreads();
  }
}

template<int NT, int S>
float run(double2* bufs[4], size_t n, int iters, hipStream_t s) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  unsigned nb = n / (NT * S);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_pipe<NT,S>), dim3(nb), dim3(NT), 0, s, bufs[0], bufs[1], bufs[2], bufs[3]);
  (void)hipEventRecord(e0, s);
  for (int it = 0; it < iters; ++it) {
    if (it & 1) hipLaunchKernelGGL((k_pipe<NT,S>), dim3(nb), dim3(NT), 0, s, bufs[2], bufs[3], bufs[0], bufs[1]);
    else hipLaunchKernelGGL((k_pipe<NT,S>), dim3(nb), dim3(NT), 0, s, bufs[0], bufs[1], bufs[2], bufs[3]);
  }
  (void)hipEventRecord(e1, s); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / iters;
}

int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  size_t n = (size_t)1 << 20;
  double2* bufs[4];
  for (int i = 0; i < 4; ++i) { CK(hipMalloc(&bufs[i], n * sizeof(double2))); CK(hipMemset(bufs[i], 0, n * sizeof(double2))); }
  const int it = 400;
  printf("2R2W, 256 WGs x 1024 thr, sub-tiles per WG:  S=1(x4 WGs) %.2f us | S=2(x2) %.2f | S=4 %.2f us\n", run<1024,1>(bufs,n,it,s), run<1024,2>(bufs,n,it,s), run<1024,4>(bufs,n,it,s));
  printf("2R2W, 512 thr:  S=2 (1024 WGs) %.2f | S=4 (512 WGs) %.2f | S=8 (256 WGs) %.2f us\n", run<512,2>(bufs,n,it,s), run<512,4>(bufs,n,it,s), run<512,8>(bufs,n,it,s));
  printf("2R2W, 256 thr:  S=4 (1024 WGs) %.2f | S=8 (512 WGs) %.2f | S=16 (256 WGs) %.2f us\n", run<256,4>(bufs,n,it,s), run<256,8>(bufs,n,it,s), run<256,16>(bufs,n,it,s));
  return 0;
}
