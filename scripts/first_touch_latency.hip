// What the FIRST memory accesses of a short kernel cost, measured inside the kernel with the 100 MHz wall clock:
// K independent loads, one from each of K buffers, all issued before the first wait, for buffers that are
// separate small allocations, slices of one arena, or 256 KiB apart inside one large allocation; with and
// without a launch in between that reads one 64-byte sector from each of ~100k columns of a large matrix (what
// the row gather of a pivot set-up does).  A second round of K loads from the same buffers follows in the same
// kernel (addresses depend on the first round), to separate the first-touch cost from a plain round trip.
//   hipcc -O3 --offload-arch=gfx950 -o bin_scratch/first_touch_latency scripts/first_touch_latency.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kMaxBuf = 32;
struct Ptrs { const int* p[kMaxBuf]; };

template <int K>
__global__ __launch_bounds__(256) void k_first(Ptrs ptrs, unsigned* __restrict__ out, int rep) {
  const unsigned long long t0 = wall_clock64();
  int v[K];
#pragma unroll
  for (int k = 0; k < K; k++) v[k] = ptrs.p[k][threadIdx.x & 63];
  int sum = 0;
#pragma unroll
  for (int k = 0; k < K; k++) sum += v[k];
  const unsigned long long t1 = wall_clock64() + (sum == 12345 ? 1 : 0);
  int w[K];
#pragma unroll
  for (int k = 0; k < K; k++) w[k] = ptrs.p[k][64 + ((threadIdx.x + sum) & 63)];
  int sum2 = 0;
#pragma unroll
  for (int k = 0; k < K; k++) sum2 += w[k];
  const unsigned long long t2 = wall_clock64() + (sum2 == 12345 ? 1 : 0);
  if (threadIdx.x == 0) {
    out[(rep * gridDim.x + blockIdx.x) * 2] = (unsigned)(t1 - t0);
    out[(rep * gridDim.x + blockIdx.x) * 2 + 1] = (unsigned)(t2 - t1);
  }
}

__global__ __launch_bounds__(256) void k_thrash(const double* __restrict__ big, long cols, long lda, long row, double* __restrict__ sink) {
  const long c = (long)blockIdx.x * 256 + threadIdx.x;
  if (c < cols) {
    const double v = big[c * lda + row];
    if (v == 12345.0) sink[0] = v;
  }
}

template <int K>
static void run(const char* what, const Ptrs& p, bool thrash, const double* big, long cols, long lda, double* sink, unsigned* out_d) {
  const int blocks = 128, reps = 50;
  std::vector<unsigned> h(blocks * reps * 2);
  for (int r = 0; r < reps; r++) {
    if (thrash) hipLaunchKernelGGL(k_thrash, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, 0, big, cols, lda, (long)(r * 37 % 1000), sink);
    hipLaunchKernelGGL(k_first<K>, dim3(blocks), dim3(256), 0, 0, p, out_d, r);
  }
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(h.data(), out_d, sizeof(unsigned) * h.size(), hipMemcpyDeviceToHost));
  double first0 = 0, firstmax = 0, second0 = 0;
  for (int r = 10; r < reps; r++) {
    unsigned mx = 0;
    for (int b = 0; b < blocks; b++) mx = std::max(mx, h[(r * blocks + b) * 2]);
    first0 += h[(r * blocks) * 2];
    second0 += h[(r * blocks) * 2 + 1];
    firstmax += mx;
  }
  const double n = reps - 10;
  printf("%-34s K=%2d %s: first round %.2f us (block 0), %.2f us (slowest block); second round %.2f us\n", what, K,
         thrash ? "after a row gather" : "back to back     ", first0 / n / 100.0, firstmax / n / 100.0, second0 / n / 100.0);
}

int main() {
  const long lda = 32768, cols = 98304;   // 25.8 GB, one 64-byte sector per column and launch
  double* big;
  CK(hipMalloc(&big, sizeof(double) * lda * cols));
  CK(hipMemset(big, 0, sizeof(double) * lda * cols));
  double* sink;
  CK(hipMalloc(&sink, 64));
  unsigned* out_d;
  CK(hipMalloc(&out_d, sizeof(unsigned) * 128 * 50 * 2));
  Ptrs small{}, arena{}, strided{};
  for (int k = 0; k < kMaxBuf; k++) {
    int* q;
    CK(hipMalloc(&q, 4096));
    CK(hipMemset(q, 0, 4096));
    small.p[k] = q;
  }
  int* a;
  CK(hipMalloc(&a, 4096 * kMaxBuf));
  CK(hipMemset(a, 0, 4096 * kMaxBuf));
  for (int k = 0; k < kMaxBuf; k++) arena.p[k] = a + k * 1024;
  int* s;
  CK(hipMalloc(&s, (size_t)262144 * kMaxBuf));
  CK(hipMemset(s, 0, (size_t)262144 * kMaxBuf));
  for (int k = 0; k < kMaxBuf; k++) strided.p[k] = s + (size_t)k * 65536;
  for (int thrash = 0; thrash < 2; thrash++) {
    run<1>("separate 4 KiB allocations", small, thrash, big, cols, lda, sink, out_d);
    run<8>("separate 4 KiB allocations", small, thrash, big, cols, lda, sink, out_d);
    run<32>("separate 4 KiB allocations", small, thrash, big, cols, lda, sink, out_d);
    run<8>("slices of one 128 KiB arena", arena, thrash, big, cols, lda, sink, out_d);
    run<32>("slices of one 128 KiB arena", arena, thrash, big, cols, lda, sink, out_d);
    run<8>("256 KiB apart in one allocation", strided, thrash, big, cols, lda, sink, out_d);
    run<32>("256 KiB apart in one allocation", strided, thrash, big, cols, lda, sink, out_d);
  }
  return 0;
}
