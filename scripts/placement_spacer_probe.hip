// Can two separate hipMalloc buffers be pushed into different HBM placement classes with a
// temporary spacer allocation between them?  A (8.6 GB), spacer (X GB), B (8.6 GB), spacer freed,
// then the read-modify-write rate over A and B together.
//   hipcc -O3 --offload-arch=gfx950 -o bin_scratch/spacer scripts/placement_spacer_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256, 2) void k_rmw(double2* __restrict__ a, double2* __restrict__ b, long chunks_per_half, double f) {
  for (long c = blockIdx.x; c < 2 * chunks_per_half; c += gridDim.x) {
    double2* p = (c & 1 ? b : a) + (c >> 1) * 4096 + threadIdx.x;
    double2 v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) { v[u].x = __builtin_nontemporal_load(&p[u * 256].x); v[u].y = __builtin_nontemporal_load(&p[u * 256].y); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 16; u++) { __builtin_nontemporal_store(v[u].x * f + 1.0, &p[u * 256].x); __builtin_nontemporal_store(v[u].y * f + 1.0, &p[u * 256].y); }
  }
}

int main(int argc, char** argv) {
  const size_t half = (size_t)8600 << 20;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int i = 1; i < argc; i++) {
    const double gap_gb = atof(argv[i]);
    char *a = nullptr, *sp = nullptr, *b = nullptr;
    CK(hipMalloc(&a, half));
    if (gap_gb > 0) CK(hipMalloc(&sp, (size_t)(gap_gb * 1e9)));
    CK(hipMalloc(&b, half));
    if (sp) CK(hipFree(sp));
    CK(hipMemset(a, 0, half));
    CK(hipMemset(b, 0, half));
    const long chunks = (long)(half / 65536);
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(k_rmw, dim3(8192), dim3(256), 0, 0, (double2*)a, (double2*)b, chunks, 0.5);
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k_rmw, dim3(8192), dim3(256), 0, 0, (double2*)a, (double2*)b, chunks, 0.5);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("spacer %6.1f GB: a=%p b=%p (VA distance %.1f GB)  %8.3f ms  %7.1f GB/s\n", gap_gb, (void*)a, (void*)b,
           ((double)((intptr_t)b - (intptr_t)a)) / 1e9, ms / 5, 4.0 * half / (ms / 5 * 1e-3) / 1e9);
    CK(hipFree(a));
    CK(hipFree(b));
  }
  return 0;
}
