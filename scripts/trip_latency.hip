// What a short, latency-bound kernel costs on this GPU: a chain of launches on one stream in which every
// kernel reads what the kernel before it wrote (as the set-up kernels of a chain of pivots do), with
// 0..6 DEPENDENT round trips to that data per kernel.  Durations come from rocprofv3 --kernel-trace
// --stats; the program also prints the wall time per launch of each chain.
//   hipcc -O3 --offload-arch=gfx950 -o bin_scratch/trip_latency scripts/trip_latency.hip
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// Every thread follows `kTrips` indices: the next address depends on the value just loaded.  `in` was written
// by the launch before (by workgroups all over the chip), `out` is what the next launch reads.
template <int kTrips>
__global__ __launch_bounds__(256) void k_trips(const int* __restrict__ in, int* __restrict__ out, int n) {
  int i = blockIdx.x * 256 + threadIdx.x;
  int v = i;
#pragma unroll
  for (int t = 0; t < kTrips; t++) v = in[(v + t * 4099) % n];
  out[i] = (v + 1) % n;
}

// The same with one block-wide reduction through LDS per trip (two barriers), as the folds of the set-up kernels do.
template <int kTrips>
__global__ __launch_bounds__(256) void k_trips_fold(const int* __restrict__ in, int* __restrict__ out, int n) {
  __shared__ int s[4];
  int i = blockIdx.x * 256 + threadIdx.x;
  int v = i;
#pragma unroll
  for (int t = 0; t < kTrips; t++) {
    v = in[(v + t * 4099) % n];
    int w = v;
    for (int off = 32; off > 0; off >>= 1) w = max(w, __shfl_down(w, off, 64));
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = w;
    __syncthreads();
    v = (v + max(max(s[0], s[1]), max(s[2], s[3]))) % n;
    __syncthreads();
  }
  out[i] = (v + 1) % n;
}

template <typename F>
static void chain(const char* name, int blocks, int launches, F launch) {
  for (int k = 0; k < 20; k++) launch(k);
  CK(hipDeviceSynchronize());
  auto t0 = std::chrono::steady_clock::now();
  for (int k = 0; k < launches; k++) launch(k);
  CK(hipDeviceSynchronize());
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  printf("%-28s blocks %4d: %.2f us per launch\n", name, blocks, us / launches);
}

int main() {
  const int launches = 400;
  for (int blocks : {48, 176, 512}) {
    const int n = blocks * 256;
    int *a, *b;
    CK(hipMalloc(&a, sizeof(int) * n));
    CK(hipMalloc(&b, sizeof(int) * n));
    CK(hipMemset(a, 0, sizeof(int) * n));
    CK(hipMemset(b, 0, sizeof(int) * n));
#define RUN(K, T)                                                                                                 \
  chain(#K "<" #T ">", blocks, launches, [&](int k) {                                                             \
    hipLaunchKernelGGL(K<T>, dim3(blocks), dim3(256), 0, 0, (k & 1) ? b : a, (k & 1) ? a : b, n);                  \
  })
    RUN(k_trips, 0); RUN(k_trips, 1); RUN(k_trips, 2); RUN(k_trips, 4); RUN(k_trips, 6);
    RUN(k_trips_fold, 1); RUN(k_trips_fold, 2); RUN(k_trips_fold, 4); RUN(k_trips_fold, 6);
    CK(hipFree(a));
    CK(hipFree(b));
  }
  return 0;
}
