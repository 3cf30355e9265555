// Probe 4: which ingredient of the real chain kernel costs the read/write overlap that probe 3 shows?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)
struct Big { const double2* a; const double2* b; double2* c; double2* d; const double* tab; int lo, hs, hb; unsigned pad[180]; };

template<int NT, int S, bool NTSTORE, bool TABLE, bool IDX>
__global__ __launch_bounds__(NT) void k_pipe(Big g) {
  extern __shared__ __attribute__((aligned(16))) double2 sh[];
  const size_t base = (size_t)blockIdx.x * NT * S;
  double2 x[S], y[S]; double tv[S]; size_t off[S];
#pragma unroll
  for (int k = 0; k < S; ++k) {
    unsigned i = k*NT + threadIdx.x;
    if (IDX) { unsigned lomask=(1u<<g.lo)-1u; i = (i & lomask) | ((i >> g.lo) << g.hs); }
    off[k] = base + i;
    x[k] = g.a[off[k]]; y[k] = g.b[off[k]];
    if (TABLE) tv[k] = g.tab[k*NT + threadIdx.x];
  }
#pragma unroll
  for (int k = 0; k < S; ++k) {
    double2* tile = sh + k*NT;
    tile[threadIdx.x] = x[k];
    __syncthreads();
    double2 p = tile[threadIdx.x ^ 1];
    x[k].x += 1e-30 * p.y;
    if (TABLE) x[k].y += 1e-30*tv[k];
    if (NTSTORE) { __builtin_nontemporal_store(x[k].x, &g.c[off[k]].x); __builtin_nontemporal_store(x[k].y, &g.c[off[k]].y); } else g.c[off[k]] = x[k];
    __syncthreads();
    tile[threadIdx.x] = x[k];
    __syncthreads();
    p = tile[threadIdx.x ^ 2];
    y[k].x += 1e-30 * p.x;
    if (NTSTORE) { __builtin_nontemporal_store(y[k].x, &g.d[off[k]].x); __builtin_nontemporal_store(y[k].y, &g.d[off[k]].y); } else g.d[off[k]] = y[k];
  }
}

template<int NT, int S, bool NTSTORE, bool TABLE, bool IDX>
float run(double2* bufs[4], double* tab, size_t n, int iters, size_t lds, hipStream_t s) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  unsigned nb = n / (NT * S);
  auto kern = k_pipe<NT,S,NTSTORE,TABLE,IDX>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds));
  Big g{}; g.tab = tab; g.lo = 12; g.hs = 12; g.hb = 0;
  for (int it = -3; it < iters; ++it) {
    if (it == 0) (void)hipEventRecord(e0, s);
    if (it & 1) { g.a=bufs[2]; g.b=bufs[3]; g.c=bufs[0]; g.d=bufs[1]; } else { g.a=bufs[0]; g.b=bufs[1]; g.c=bufs[2]; g.d=bufs[3]; }
    hipLaunchKernelGGL(kern, dim3(nb), dim3(NT), lds, s, g);
  }
  (void)hipEventRecord(e1, s); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f / iters;
}

int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  size_t n = (size_t)1 << 20;
  double2* bufs[4]; double* tab;
  for (int i = 0; i < 4; ++i) { CK(hipMalloc(&bufs[i], n * sizeof(double2))); CK(hipMemset(bufs[i], 0, n * sizeof(double2))); }
  CK(hipMalloc(&tab, 4096*8)); CK(hipMemset(tab, 0, 4096*8));
  const int it = 400;
  printf("base (16 KiB LDS)            : %.2f us\n", run<1024,4,false,false,false>(bufs,tab,n,it,4*1024*16,s));
  printf("64 KiB dynamic LDS           : %.2f us\n", run<1024,4,false,false,false>(bufs,tab,n,it,65536+256,s));
  printf("+ nt stores                  : %.2f us\n", run<1024,4,true,false,false>(bufs,tab,n,it,65536+256,s));
  printf("+ table loads                : %.2f us\n", run<1024,4,true,true,false>(bufs,tab,n,it,65536+256,s));
  printf("+ runtime index math         : %.2f us\n", run<1024,4,true,true,true>(bufs,tab,n,it,65536+256,s));
  printf("S=2 all                      : %.2f us\n", run<1024,2,true,true,true>(bufs,tab,n,it,65536+256,s));
  printf("plain stores, table, idx     : %.2f us\n", run<1024,4,false,true,true>(bufs,tab,n,it,65536+256,s));
  return 0;
}
