// Stand-alone exploration of rank-1 update kernel shapes on a fully dense
// m x n fp64 block (no unit columns): which work decomposition gets closest to
// the HBM roofline.  Not part of the library; build with
//   hipcc -O3 -ffp-contract=off --offload-arch=gfx950 -o build/update_microbench scripts/update_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ double flushz(double v) { return fabs(v) < (double)1e-6f ? 0.0 : v; }
__device__ __forceinline__ double2 apply(double2 t, double2 c, double r) {
  double2 o;
  o.x = flushz(__dadd_rn(t.x, __dmul_rn(c.x, r)));
  o.y = flushz(__dadd_rn(t.y, __dmul_rn(c.y, r)));
  return o;
}

template <bool NT> __device__ __forceinline__ double2 ld(const double2* p) {
  if (NT) { double2 v; v.x = __builtin_nontemporal_load(&p->x); v.y = __builtin_nontemporal_load(&p->y); return v; }
  return *p;
}
template <bool NT> __device__ __forceinline__ void st(double2* p, double2 v) {
  if (NT) { __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y); }
  else *p = v;
}

// V0: block owns SLABS*512 rows, keeps col' in registers, walks columns (stride = blocks per row chunk), UNR columns per step
template <int SLABS, int UNR, bool NT>
__global__ __launch_bounds__(256) void k_rowchunk(double* __restrict__ T, long lda, int n_slabs, int ncols,
                                                  const double* __restrict__ colp, const double* __restrict__ rowp) {
  const int n_rc = (n_slabs + SLABS - 1) / SLABS;
  const int rc = blockIdx.x % n_rc, cstart = blockIdx.x / n_rc, cstride = gridDim.x / n_rc;
  const long row0 = (long)rc * SLABS * 512 + threadIdx.x * 2;
  double2 c[SLABS];
#pragma unroll
  for (int s = 0; s < SLABS; s++) c[s] = *reinterpret_cast<const double2*>(colp + row0 + s * 512);
  int k = cstart;
  for (; k + (UNR - 1) * cstride < ncols; k += UNR * cstride) {
    double2 t[UNR][SLABS];
    double r[UNR];
#pragma unroll
    for (int u = 0; u < UNR; u++) {
      r[u] = rowp[k + u * cstride];
      const double2* p = reinterpret_cast<const double2*>(T + (long)(k + u * cstride) * lda + row0);
#pragma unroll
      for (int s = 0; s < SLABS; s++) t[u][s] = ld<NT>(p + s * 256);
    }
#pragma unroll
    for (int u = 0; u < UNR; u++) {
      double2* p = reinterpret_cast<double2*>(T + (long)(k + u * cstride) * lda + row0);
#pragma unroll
      for (int s = 0; s < SLABS; s++) st<NT>(p + s * 256, apply(t[u][s], c[s], r[u]));
    }
  }
  for (; k < ncols; k += cstride) {
    const double r0 = rowp[k];
    double2* p = reinterpret_cast<double2*>(T + (long)k * lda + row0);
#pragma unroll
    for (int s = 0; s < SLABS; s++) st<NT>(p + s * 256, apply(ld<NT>(p + s * 256), c[s], r0));
  }
}

// V2: block walks whole columns; col' re-read from L2 for every column.
template <int UNR, bool NT>
__global__ __launch_bounds__(256) void k_fullcol(double* __restrict__ T, long lda, int n_slabs, int ncols,
                                                 const double* __restrict__ colp, const double* __restrict__ rowp) {
  for (int k = blockIdx.x; k < ncols; k += gridDim.x) {
    const double r0 = rowp[k];
    double2* p = reinterpret_cast<double2*>(T + (long)k * lda) + threadIdx.x;
    const double2* cp = reinterpret_cast<const double2*>(colp) + threadIdx.x;
    int s = 0;
    for (; s + UNR <= n_slabs; s += UNR) {
      double2 t[UNR], c[UNR];
#pragma unroll
      for (int u = 0; u < UNR; u++) { t[u] = ld<NT>(p + (s + u) * 256); c[u] = cp[(s + u) * 256]; }
#pragma unroll
      for (int u = 0; u < UNR; u++) st<NT>(p + (s + u) * 256, apply(t[u], c[u], r0));
    }
    for (; s < n_slabs; s++) st<NT>(p + s * 256, apply(ld<NT>(p + s * 256), cp[s * 256], r0));
  }
}

// V3: like V2 with separate nontemporal switches, block size BS and optional contiguous column ranges per block.
template <int BS, int UNR, bool NTL, bool NTS, bool CONTIG>
__global__ __launch_bounds__(BS) void k_fullcol2(double* __restrict__ T, long lda, int n_vec, int ncols,
                                                  const double* __restrict__ colp, const double* __restrict__ rowp) {
  // n_vec = number of double2 per column
  int k0, k1, kstep;
  if (CONTIG) { const int per = (ncols + gridDim.x - 1) / gridDim.x; k0 = blockIdx.x * per; k1 = min(ncols, k0 + per); kstep = 1; }
  else { k0 = blockIdx.x; k1 = ncols; kstep = gridDim.x; }
  for (int k = k0; k < k1; k += kstep) {
    const double r0 = rowp[k];
    double2* p = reinterpret_cast<double2*>(T + (long)k * lda);
    const double2* cp = reinterpret_cast<const double2*>(colp);
    int v = threadIdx.x;
    for (; v + (UNR - 1) * BS < n_vec; v += UNR * BS) {
      double2 t[UNR], c[UNR];
#pragma unroll
      for (int u = 0; u < UNR; u++) { t[u] = ld<NTL>(p + v + u * BS); c[u] = cp[v + u * BS]; }
#pragma unroll
      for (int u = 0; u < UNR; u++) st<NTS>(p + v + u * BS, apply(t[u], c[u], r0));
    }
    for (; v < n_vec; v += BS) st<NTS>(p + v, apply(ld<NTL>(p + v), cp[v], r0));
  }
}

// V4: linear sweep.  Work item = (column, batch of UNR*BS vectors); items are dealt round-robin so the
// resident workgroups always cover one contiguous window of memory.  All loads of a batch before its stores.
template <int BS, int UNR, bool NT>
__global__ __launch_bounds__(BS, 2) void k_linear(double* __restrict__ T, long lda, int n_vec, int ncols,
                                                   const double* __restrict__ colp, const double* __restrict__ rowp) {
  const int per_col = n_vec / (UNR * BS);
  const long n_items = (long)ncols * per_col;
  const double2* cp = reinterpret_cast<const double2*>(colp);
  for (long it = blockIdx.x; it < n_items; it += gridDim.x) {
    const int k = (int)(it / per_col), ch = (int)(it % per_col);
    const double r0 = rowp[k];
    double2* p = reinterpret_cast<double2*>(T + (long)k * lda) + (long)ch * UNR * BS + threadIdx.x;
    const double2* c0 = cp + (long)ch * UNR * BS + threadIdx.x;
    double2 t[UNR], c[UNR];
#pragma unroll
    for (int u = 0; u < UNR; u++) { t[u] = ld<NT>(p + u * BS); c[u] = c0[u * BS]; }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < UNR; u++) st<NT>(p + u * BS, apply(t[u], c[u], r0));
  }
}

// V5: full columns with all loads of a batch issued before its stores (the engine's kernel)
template <int BS, int UNR, bool NT>
__global__ __launch_bounds__(BS, 2) void k_fullcol_batched(double* __restrict__ T, long lda, int n_vec, int ncols,
                                                            const double* __restrict__ colp, const double* __restrict__ rowp) {
  const double2* cp = reinterpret_cast<const double2*>(colp);
  for (int k = blockIdx.x; k < ncols; k += gridDim.x) {
    const double r0 = rowp[k];
    double2* p = reinterpret_cast<double2*>(T + (long)k * lda);
    for (int v = threadIdx.x; v + (UNR - 1) * BS < n_vec; v += UNR * BS) {
      double2 t[UNR], c[UNR];
#pragma unroll
      for (int u = 0; u < UNR; u++) { t[u] = ld<NT>(p + v + u * BS); c[u] = cp[v + u * BS]; }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < UNR; u++) st<NT>(p + v + u * BS, apply(t[u], c[u], r0));
    }
  }
}

template <bool NT>
__global__ __launch_bounds__(256) void k_copy_nt(const double2* __restrict__ a, double2* __restrict__ b, long n2) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long)gridDim.x * 256) st<NT>(b + i, ld<NT>(a + i));
}

// reference points: out-of-place copy and in-place scale, 16 B per lane
__global__ __launch_bounds__(256) void k_copy(const double2* __restrict__ a, double2* __restrict__ b, long n2) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long)gridDim.x * 256) b[i] = a[i];
}
__global__ __launch_bounds__(256) void k_scale(double2* __restrict__ a, long n2, double s) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n2; i += (long)gridDim.x * 256) { double2 v = a[i]; v.x *= s; v.y *= s; a[i] = v; }
}

template <typename F> float time_it(F f, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  f(); f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; i++) f();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGetLastError());
  return ms / reps;
}

int main(int argc, char** argv) {
  long m = argc > 1 ? atol(argv[1]) : 32768;
  int n = argc > 2 ? atoi(argv[2]) : 16384;
  int reps = argc > 3 ? atoi(argv[3]) : 10;
  long lda = (m + 511) / 512 * 512;
  int n_slabs = lda / 512;
  size_t bytes = sizeof(double) * lda * n;
  double *T, *T2, *colp, *rowp;
  CK(hipMalloc(&T, bytes)); CK(hipMalloc(&T2, bytes)); CK(hipMalloc(&colp, lda * 8)); CK(hipMalloc(&rowp, n * 8));
  std::vector<double> h(lda), hr(n);
  for (long i = 0; i < lda; i++) h[i] = i < m ? -0.5 - (i % 7) * 0.01 : 0.0;
  for (int j = 0; j < n; j++) hr[j] = 1e-3 * ((j % 5) - 2.5);
  CK(hipMemcpy(colp, h.data(), lda * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(rowp, hr.data(), n * 8, hipMemcpyHostToDevice));
  CK(hipMemset(T, 0, bytes)); CK(hipMemset(T2, 0, bytes));
  hipLaunchKernelGGL(k_scale, dim3(4096), dim3(256), 0, 0, (double2*)T, (long)(lda * n / 2), 1.0);
  const double gb = 16.0 * lda * n / 1e9;
  printf("m=%ld n=%d lda=%ld  %.2f GB tableau, %.2f GB moved per pass\n", m, n, lda, bytes / 1e9, gb);
  auto report = [&](const char* name, float ms) { printf("%-44s %8.3f ms  %7.1f GB/s  %.3f of 8 TB/s\n", name, ms, gb / (ms * 1e-3), gb / (ms * 1e-3) / 8000.0); fflush(stdout); };
  long n2 = lda * n / 2;
  const int n_vec = lda / 2;
  for (int g : {2048, 8192})  { char b[64]; snprintf(b, 64, "copy out-of-place grid=%d", g); report(b, time_it([&] { hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, 0, (const double2*)T, (double2*)T2, n2); }, reps)); }
  for (int g : {2048, 8192})  { char b[64]; snprintf(b, 64, "copy nt grid=%d", g); report(b, time_it([&] { hipLaunchKernelGGL(k_copy_nt<true>, dim3(g), dim3(256), 0, 0, (const double2*)T, (double2*)T2, n2); }, reps)); }
  { report("hipMemcpyDtoD", time_it([&] { CK(hipMemcpyAsync(T2, T, bytes, hipMemcpyDeviceToDevice, 0)); }, reps)); }
#define RUN_V(KERN, BS, UN, NT, G) do { char b[112]; snprintf(b, 112, #KERN " bs=%d unr=%d nt=%d grid=%d", BS, UN, NT, G); \
    report(b, time_it([&] { hipLaunchKernelGGL((KERN<BS, UN, NT>), dim3(G), dim3(BS), 0, 0, T, lda, n_vec, n, colp, rowp); }, reps)); } while (0)
  RUN_V(k_fullcol_batched, 256, 16, true, 8192);
  RUN_V(k_fullcol_batched, 256, 16, true, 768);
  RUN_V(k_fullcol_batched, 256, 8, true, 8192);
  RUN_V(k_fullcol_batched, 512, 8, true, 4096);
  RUN_V(k_fullcol_batched, 512, 16, true, 4096);
  RUN_V(k_fullcol_batched, 1024, 8, true, 2048);
  RUN_V(k_linear, 256, 16, true, 768);
  RUN_V(k_linear, 256, 16, true, 2048);
  RUN_V(k_linear, 256, 16, true, 8192);
  RUN_V(k_linear, 256, 16, true, 65536);
  RUN_V(k_linear, 256, 8, true, 8192);
  RUN_V(k_linear, 512, 8, true, 4096);
  RUN_V(k_linear, 512, 16, true, 4096);
  RUN_V(k_linear, 1024, 8, true, 2048);
  RUN_V(k_linear, 1024, 16, true, 2048);
  RUN_V(k_linear, 256, 16, false, 8192);
  return 0;
}
