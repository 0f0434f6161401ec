// What cold instruction fetch costs a short kernel: the same 4096 dependent add-xor pairs executed as
// straight-line code (about 48 KiB of instructions, every line fetched once) and as a loop (one line, fetched once),
// timed inside the kernel with the 100 MHz wall clock, one wavefront per workgroup, a different kernel in between.
//   hipcc -O3 --offload-arch=gfx950 -o bin_scratch/code_fetch_latency scripts/code_fetch_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int kN, bool kUnrolled>
__global__ __launch_bounds__(64) void k_code(const int* __restrict__ in, unsigned* __restrict__ out, int rep) {
  const unsigned long long t0 = wall_clock64();
  int v = in[threadIdx.x];
  const int a = in[64 + threadIdx.x];
  const unsigned long long t1 = wall_clock64() + (v == 12345 ? 1 : 0);
  if (kUnrolled) {
#pragma unroll
    for (int k = 0; k < kN; k++) v = (v + (int)(k * 2654435761u >> 7)) ^ a;   // a different literal per step: nothing folds
  } else {
#pragma unroll 1
    for (int k = 0; k < kN; k++) v = (v + (int)(k * 2654435761u >> 7)) ^ a;
  }
  const unsigned long long t2 = wall_clock64() + (v == 12345 ? 1 : 0);
  if (threadIdx.x == 0) {
    out[(rep * gridDim.x + blockIdx.x) * 2] = (unsigned)(t1 - t0);
    out[(rep * gridDim.x + blockIdx.x) * 2 + 1] = (unsigned)(t2 - t1);
  }
}
__global__ void k_other(int* p) { if (threadIdx.x == 999) p[0] = 1; }

template <int kN, bool kUnrolled>
static void run(const int* in, unsigned* out_d, int* scratch) {
  const int blocks = 256, reps = 40;
  CK(hipMemset(out_d, 0, sizeof(unsigned) * blocks * reps * 2));
  std::vector<unsigned> h(blocks * reps * 2);
  for (int r = 0; r < reps; r++) {
    hipLaunchKernelGGL(k_other, dim3(256), dim3(256), 0, 0, scratch);
    hipLaunchKernelGGL((k_code<kN, kUnrolled>), dim3(blocks), dim3(64), 0, 0, in, out_d, r);
  }
  CK(hipGetLastError());
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(h.data(), out_d, sizeof(unsigned) * h.size(), hipMemcpyDeviceToHost));
  double load = 0, body = 0;
  for (int r = 10; r < reps; r++) { load += h[(r * blocks) * 2]; body += h[(r * blocks) * 2 + 1]; }
  printf("%5d steps %s: first loads %.2f us, body %.2f us (%.1f ns per step)\n", kN, kUnrolled ? "straight-line" : "loop         ",
         load / 30 / 100.0, body / 30 / 100.0, body / 30 * 10.0 / kN);
}
int main() {
  int* in; unsigned* out_d; int* scratch;
  CK(hipMalloc(&in, 4096)); CK(hipMemset(in, 0, 4096));
  CK(hipMalloc(&out_d, sizeof(unsigned) * 256 * 40 * 2));
  CK(hipMalloc(&scratch, 64));
  run<256, true>(in, out_d, scratch);  run<256, false>(in, out_d, scratch);
  run<1024, true>(in, out_d, scratch); run<1024, false>(in, out_d, scratch);
  run<4096, true>(in, out_d, scratch); run<4096, false>(in, out_d, scratch);
  return 0;
}
