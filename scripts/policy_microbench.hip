// Cache-policy exploration for the streaming read-modify-write of the rank-1
// update: buffer loads/stores with every combination of the sc0 / sc1 / nt bits.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double flushz(double v) { return fabs(v) < (double)1e-6f ? 0.0 : v; }

// aux bits on gfx940+: 1 = sc0, 2 = nt, 16 = sc1
template <int LAUX, int SAUX, int UNR, int BS>
__global__ __launch_bounds__(BS) void k_rmw(double* __restrict__ T, long lda, int n_vec, int ncols,
                                            const double* __restrict__ colp, const double* __restrict__ rowp) {
  for (int k = blockIdx.x; k < ncols; k += gridDim.x) {
    const double r0 = rowp[k];
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(T + (long)k * lda, 0, (int)(lda * 8), 0x27000);
    const v2d* cp = reinterpret_cast<const v2d*>(colp);
    int v = threadIdx.x;
    for (; v + (UNR - 1) * BS < n_vec; v += UNR * BS) {
      v4i t[UNR]; v2d c[UNR];
#pragma unroll
      for (int u = 0; u < UNR; u++) { t[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (v + u * BS) * 16, 0, LAUX); c[u] = cp[v + u * BS]; }
#pragma unroll
      for (int u = 0; u < UNR; u++) {
        v2d x = __builtin_bit_cast(v2d, t[u]);
        x.x = flushz(__dadd_rn(x.x, __dmul_rn(c[u].x, r0)));
        x.y = flushz(__dadd_rn(x.y, __dmul_rn(c[u].y, r0)));
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, x), rs, (v + u * BS) * 16, 0, SAUX);
      }
    }
  }
}
// read-only and write-only ceilings
__global__ __launch_bounds__(256) void k_read(const v2d* __restrict__ a, long n2, double* out) {
  double s = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long)gridDim.x * 256) { v2d v = __builtin_nontemporal_load(a + i); s += v.x + v.y; }
  if (s == 12345.678) out[0] = s;
}
__global__ __launch_bounds__(256) void k_write(v2d* __restrict__ a, long n2, double x) {
  v2d v = {x, x};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long)gridDim.x * 256) __builtin_nontemporal_store(v, a + i);
}
template <typename F> float time_it(F f, int reps) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0)); for (int i = 0; i < reps; i++) f(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipGetLastError()); return ms / reps;
}
int main(int argc, char** argv) {
  long m = argc > 1 ? atol(argv[1]) : 32768; int n = argc > 2 ? atoi(argv[2]) : 16384; int reps = 10;
  long lda = (m + 511) / 512 * 512; size_t bytes = sizeof(double) * lda * n;
  double *T, *colp, *rowp, *out;
  CK(hipMalloc(&T, bytes)); CK(hipMalloc(&colp, lda * 8)); CK(hipMalloc(&rowp, n * 8)); CK(hipMalloc(&out, 64));
  std::vector<double> h(lda), hr(n), hT(lda * 64);
  for (long i = 0; i < lda; i++) h[i] = i < m ? -0.5 - (i % 7) * 0.01 : 0.0;
  for (int j = 0; j < n; j++) hr[j] = 1e-9 * ((j % 5) - 2.5);
  for (size_t i = 0; i < hT.size(); i++) hT[i] = -0.1 - (double)(i * 2654435761u % 1000003) / 1000003.0;
  CK(hipMemcpy(colp, h.data(), lda * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(rowp, hr.data(), n * 8, hipMemcpyHostToDevice));
  for (int j = 0; j < n; j += 64) CK(hipMemcpy(T + (long)j * lda, hT.data(), sizeof(double) * lda * 64, hipMemcpyHostToDevice));  // random-ish, non-zero data
  const double gb = 16.0 * lda * n / 1e9; const int n_vec = lda / 2; long n2 = lda * n / 2;
  printf("m=%ld n=%d  %.2f GB tableau (non-zero data)\n", m, n, bytes / 1e9);
  auto report = [&](const char* name, float ms, double g) { printf("%-52s %8.3f ms  %7.1f GB/s  %.3f of 8 TB/s\n", name, ms, g / (ms * 1e-3), g / (ms * 1e-3) / 8000.0); fflush(stdout); };
  report("read-only nt (8 B/elem)", time_it([&] { hipLaunchKernelGGL(k_read, dim3(8192), dim3(256), 0, 0, (const v2d*)T, n2, out); }, reps), gb / 2);
#define RUN(L, S, U, B, G) do { char b[96]; snprintf(b, 96, "rmw load_aux=%d store_aux=%d unr=%d bs=%d grid=%d", L, S, U, B, G); \
    report(b, time_it([&] { hipLaunchKernelGGL((k_rmw<L, S, U, B>), dim3(G), dim3(B), 0, 0, T, lda, n_vec, n, colp, rowp); }, reps), gb); } while (0)
  RUN(0, 0, 8, 256, 2048);
  RUN(2, 2, 8, 256, 2048);
  RUN(2, 2, 16, 256, 2048);
  RUN(2, 2, 8, 512, 1024);
  RUN(1, 1, 8, 256, 2048);
  RUN(16, 16, 8, 256, 2048);
  RUN(17, 17, 8, 256, 2048);
  RUN(18, 18, 8, 256, 2048);
  RUN(19, 19, 8, 256, 2048);
  RUN(3, 3, 8, 256, 2048);
  RUN(2, 18, 8, 256, 2048);
  RUN(18, 2, 8, 256, 2048);
  RUN(2, 16, 8, 256, 2048);
  RUN(2, 17, 8, 256, 2048);
  RUN(2, 19, 8, 256, 2048);
  RUN(0, 2, 8, 256, 2048);
  RUN(0, 18, 8, 256, 2048);
  RUN(16, 2, 8, 256, 2048);
  report("write-only nt (8 B/elem)", time_it([&] { hipLaunchKernelGGL(k_write, dim3(8192), dim3(256), 0, 0, (v2d*)T, n2, 1.5); }, reps), gb / 2);
  return 0;
}
