// Probe 2: read-only / write-only / phase-separated passes over Infinity-Cache-resident vectors (N=20, 16 MiB each).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)

template<int NT, int R, int MODE>   // MODE 0: read 2 vectors only; 1: write 2 only; 2: read 2, barrier, write 2; 3: streaming (read 2 write 2 per element)
__global__ __launch_bounds__(NT) void k(const double2* __restrict__ a, const double2* __restrict__ b, double2* __restrict__ c, double2* __restrict__ d, int flag) {
  __shared__ double2 sh[64];
  const size_t base = (size_t)blockIdx.x * NT * R;
  double2 x[R], y[R];
  if (MODE != 1) {
#pragma unroll
    for (int r = 0; r < R; ++r) { x[r] = a[base + r*NT + threadIdx.x]; y[r] = b[base + r*NT + threadIdx.x]; }
  } else {
#pragma unroll
    for (int r = 0; r < R; ++r) { x[r] = make_double2(threadIdx.x, r); y[r] = make_double2(r, blockIdx.x); }
  }
  if (MODE == 2) { if (threadIdx.x < 64) sh[threadIdx.x] = x[0]; __syncthreads(); x[0].x += sh[(threadIdx.x+1)&63].y; }
  if (MODE == 3) {
#pragma unroll
    for (int r = 0; r < R; ++r) { c[base + r*NT + threadIdx.x] = x[r]; d[base + r*NT + threadIdx.x] = y[r]; }
    return;
  }
  if (MODE == 0) {
    double s = 0; 
#pragma unroll
    for (int r = 0; r < R; ++r) s += x[r].x + y[r].y;
    c[(size_t)blockIdx.x * NT + threadIdx.x] = make_double2(s, s);  // one 16-B store per thread keeps the loads alive
  } else {
#pragma unroll
    for (int r = 0; r < R; ++r) { c[base + r*NT + threadIdx.x] = x[r]; d[base + r*NT + threadIdx.x] = y[r]; }
  }
}

template<int NT, int R, int MODE>
float run(double2* bufs[4], size_t n, int iters, hipStream_t s) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  unsigned nb = n / (NT * R);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<NT,R,MODE>), dim3(nb), dim3(NT), 0, s, bufs[0], bufs[1], bufs[2], bufs[3], 0);
  (void)hipEventRecord(e0, s);
  for (int it = 0; it < iters; ++it) {
    if (it & 1) hipLaunchKernelGGL((k<NT,R,MODE>), dim3(nb), dim3(NT), 0, s, bufs[2], bufs[3], bufs[0], bufs[1], 0);
    else hipLaunchKernelGGL((k<NT,R,MODE>), dim3(nb), dim3(NT), 0, s, bufs[0], bufs[1], bufs[2], bufs[3], 0);
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
  printf("256 WGs x 1024 thr x 4 elem (1 WG/CU):  read2 %.2f us | write2 %.2f us | read2-barrier-write2 %.2f us | no-barrier %.2f\n",
     run<1024,4,0>(bufs,n,it,s), run<1024,4,1>(bufs,n,it,s), run<1024,4,2>(bufs,n,it,s), run<1024,4,3>(bufs,n,it,s));
  printf("256 WGs x 512 thr x 8 elem:             read2 %.2f us | write2 %.2f us | read2-barrier-write2 %.2f us\n",
     run<512,8,0>(bufs,n,it,s), run<512,8,1>(bufs,n,it,s), run<512,8,2>(bufs,n,it,s));
  printf("1024 WGs x 256 thr x 4 elem (4 WG/CU):  read2 %.2f us | write2 %.2f us | read2-barrier-write2 %.2f us\n",
     run<256,4,0>(bufs,n,it,s), run<256,4,1>(bufs,n,it,s), run<256,4,2>(bufs,n,it,s));
  printf("2048 WGs x 256 thr x 2 elem (8 WG/CU):  read2 %.2f us | write2 %.2f us | read2-barrier-write2 %.2f us\n",
     run<256,2,0>(bufs,n,it,s), run<256,2,1>(bufs,n,it,s), run<256,2,2>(bufs,n,it,s));
  printf("512 WGs x 512 thr x 4 elem (2 WG/CU):   read2 %.2f us | write2 %.2f us | read2-barrier-write2 %.2f us\n",
     run<512,4,0>(bufs,n,it,s), run<512,4,1>(bufs,n,it,s), run<512,4,2>(bufs,n,it,s));
  return 0;
}
