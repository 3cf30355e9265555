// Probe 5: what does a grid-wide barrier cost on this chip, and does a PERSISTENT pass loop (one co-resident workgroup per
// CU, flag barrier between passes, explicit release/acquire) beat one launch per pass for the 2R+2W tile-exchange pattern
// of the chained factor passes (16 MiB vectors, 256 tiles of 2^12, 1024 threads)?
//   hipcc --offload-arch=gfx950 -O3 -o bw_probe5 bw_probe5.hip && ./bw_probe5
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)

constexpr int NT = 1024, R = 4, TILE = NT * R;
constexpr unsigned kSpinLimit = 1u << 22;  // bounded spin: a stuck barrier ends the kernel instead of hanging the GPU

// layout A: tile = contiguous 2^12 run; layout B: 256-byte runs (16 amplitudes) strided by 2^12 amplitudes (N = 20)
__device__ __forceinline__ size_t addr(int layout, unsigned t, unsigned i) {
  if (layout == 0) return (size_t)t * TILE + i;
  return ((size_t)(i >> 4) << 12) | ((size_t)t << 4) | (i & 15u);
}

__device__ __forceinline__ void pass_body(int layout, unsigned t, const double2* a, const double2* b, double2* c, double2* d, double2* sh) {
  double2 x[R], y[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const size_t g = addr(layout, t, r * NT + threadIdx.x);
    x[r].x = __builtin_nontemporal_load(&a[g].x); x[r].y = __builtin_nontemporal_load(&a[g].y);
    y[r].x = __builtin_nontemporal_load(&b[g].x); y[r].y = __builtin_nontemporal_load(&b[g].y);
  }
#pragma unroll
  for (int r = 0; r < R; ++r) sh[r * NT + threadIdx.x] = x[r];
  __syncthreads();
#pragma unroll
  for (int r = 0; r < R; ++r) { double2 p = sh[(r * NT + threadIdx.x) ^ 64]; y[r].x += 1e-30 * p.y; }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const size_t g = addr(layout, t, r * NT + threadIdx.x);
    __builtin_nontemporal_store(y[r].x, &c[g].x); __builtin_nontemporal_store(y[r].y, &c[g].y);
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < R; ++r) sh[r * NT + threadIdx.x] = y[r];
  __syncthreads();
#pragma unroll
  for (int r = 0; r < R; ++r) { double2 p = sh[(r * NT + threadIdx.x) ^ 128]; x[r].x = y[r].x + 1e-30 * p.x; x[r].y = y[r].y; }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const size_t g = addr(layout, t, r * NT + threadIdx.x);
    __builtin_nontemporal_store(x[r].x, &d[g].x); __builtin_nontemporal_store(x[r].y, &d[g].y);
  }
}

__global__ __launch_bounds__(NT) void k_pass(int layout, const double2* a, const double2* b, double2* c, double2* d) {
  extern __shared__ double2 sh[];
  pass_body(layout, blockIdx.x, a, b, c, d, sh);
}

// flag barrier: workgroup w publishes epoch e in flags[w * 16] (own 64-byte line); everyone waits until all flags >= e
__device__ __forceinline__ bool grid_barrier(unsigned* flags, unsigned nwg, unsigned epoch, unsigned* fail) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // stores of this workgroup reach memory
    __hip_atomic_store(&flags[blockIdx.x * 16], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  bool ok = true;
  if (threadIdx.x < 64) {
    unsigned spins = 0;
    for (;;) {
      bool all = true;
      for (unsigned w = threadIdx.x; w < nwg; w += 64)
        all &= __hip_atomic_load(&flags[w * 16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= epoch;
      if (__all(all)) break;
      if (++spins > kSpinLimit) { ok = false; if (threadIdx.x == 0) *fail = epoch; break; }
      __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
  return ok;
}

template <int NBAR>
__global__ __launch_bounds__(NT) void k_persist(int passes, double2* b0, double2* b1, double2* b2, double2* b3, unsigned* flags, unsigned* fail, int do_work) {
  extern __shared__ double2 sh[];
  unsigned epoch = 0;
  for (int it = 0; it < passes; ++it) {
    if (do_work) {
      if (it & 1) pass_body(1, blockIdx.x, b2, b3, b0, b1, sh);
      else pass_body(0, blockIdx.x, b0, b1, b2, b3, sh);
    }
    for (int k = 0; k < NBAR; ++k)
      if (!grid_barrier(flags, gridDim.x, ++epoch, fail)) return;
  }
}

int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  const size_t n = (size_t)1 << 20;
  double2* bufs[4];
  for (int i = 0; i < 4; ++i) { CK(hipMalloc(&bufs[i], n * sizeof(double2))); CK(hipMemset(bufs[i], 0, n * sizeof(double2))); }
  unsigned *flags, *fail;
  CK(hipMalloc(&flags, 256 * 64)); CK(hipMalloc(&fail, 4));
  const unsigned nb = n / TILE;
  const size_t lds = TILE * sizeof(double2);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pass), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_persist<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_persist<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_persist<1>, NT, lds));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("CUs %d, co-resident workgroups per CU %d, grid %u\n", prop.multiProcessorCount, occ, nb);
  if ((unsigned)(occ * prop.multiProcessorCount) < nb) { printf("grid does not fit: skipping persistent runs\n"); return 0; }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int it = 1000;
  float ms;
  // (a) one launch per pass
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_pass, dim3(nb), dim3(NT), lds, s, 0, bufs[0], bufs[1], bufs[2], bufs[3]);
  CK(hipEventRecord(e0, s));
  for (int i = 0; i < it; ++i) {
    if (i & 1) hipLaunchKernelGGL(k_pass, dim3(nb), dim3(NT), lds, s, 1, bufs[2], bufs[3], bufs[0], bufs[1]);
    else hipLaunchKernelGGL(k_pass, dim3(nb), dim3(NT), lds, s, 0, bufs[0], bufs[1], bufs[2], bufs[3]);
  }
  CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  printf("launch per pass (alternating layouts): %.2f us per pass\n", ms * 1e3f / it);
  // (b) persistent
  for (int mode = 0; mode < 4; ++mode) {
    const int nbar = (mode & 1) ? 2 : 1, work = mode < 2;
    CK(hipMemsetAsync(flags, 0, 256 * 64, s)); CK(hipMemsetAsync(fail, 0, 4, s));
    void* args[] = {(void*)&it, &bufs[0], &bufs[1], &bufs[2], &bufs[3], &flags, &fail, (void*)&work};
    CK(hipEventRecord(e0, s));
    if (nbar == 1) CK(hipLaunchCooperativeKernel(reinterpret_cast<void*>(k_persist<1>), dim3(nb), dim3(NT), args, lds, s));
    else CK(hipLaunchCooperativeKernel(reinterpret_cast<void*>(k_persist<2>), dim3(nb), dim3(NT), args, lds, s));
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned f = 0; CK(hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost));
    printf("persistent, %d barrier(s) per pass, %s: %.2f us per pass%s\n", nbar, work ? "2R2W work" : "barriers only", ms * 1e3f / it, f ? "  [BARRIER TIMED OUT]" : "");
  }
  return 0;
}
