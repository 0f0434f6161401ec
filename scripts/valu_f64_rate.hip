// Sustained rate of separately rounded fp64 multiply + add on every SIMD of the chip, by the number of
// independent chains a wavefront interleaves and the wavefronts per SIMD, and the shader clock it holds: how much
// arithmetic a pass over the tableau can carry per byte, and how many entries a lane must work on at once.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -o bin_scratch/valu_f64_rate scripts/valu_f64_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

// kVectorR: the common factor sits in a vector register (as after an LDS broadcast) instead of a scalar one.
// kMin: a v_min3_f32 over the upper halves after every pair of adds, as the pass kernel tracks small magnitudes.
template <int kChains, bool kVectorR, bool kMin>
__global__ __launch_bounds__(256) void k_rate(double* out, int iters, long long* clocks) {
  double t[kChains], c[kChains];
  for (int q = 0; q < kChains; q++) { t[q] = threadIdx.x * 1e-3 + q; c[q] = 1.0 + 1e-9 * (threadIdx.x + q); }
  double r = 1.0 + 1e-12 * blockIdx.x;
  if (kVectorR) r += 1e-13 * threadIdx.x;
  float low = 1e30f;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int rep = 0; rep < 8 / kChains; rep++) {
#pragma unroll
      for (int q = 0; q < kChains; q++) {
        t[q] = __dadd_rn(t[q], __dmul_rn(c[q], r));
        if (kMin && (q & 1)) asm("v_min3_f32 %0, %1, |%2|, |%3|" : "=v"(low) : "v"(low), "v"(__double2hiint(t[q - 1])), "v"(__double2hiint(t[q])));
      }
    }
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  double s = low;
  for (int q = 0; q < kChains; q++) s += t[q];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clocks[2 * blockIdx.x] = c1 - c0; clocks[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int kChains, bool kVectorR, bool kMin>
void run(int waves_per_simd, double* out, long long* clocks) {
  const int iters = 100000;
  const int grid = 256 * waves_per_simd;   // 4 wavefronts per workgroup, one per SIMD
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e30f;
  long long h[2] = {0, 1};
  for (int rep = 0; rep < 3; rep++) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_rate<kChains, kVectorR, kMin>), dim3(grid), dim3(256), 0, 0, out, iters, clocks);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) { best = ms; (void)hipMemcpy(h, clocks, sizeof(h), hipMemcpyDeviceToHost); }
  }
  const double per_wave = (double)iters * (kMin ? 20 : 16);   // instructions a wavefront issues
  printf("chains %d r in %s%s wavefronts/SIMD %d: %.3f ms, shader clock %.2f GHz, %.2f cycles per instruction per wavefront, %.2f per SIMD\n",
         kChains, kVectorR ? "VGPR" : "SGPR", kMin ? " +min3" : "", waves_per_simd, best, h[0] / (h[1] * 10.0), h[0] / per_wave, h[0] / per_wave / waves_per_simd);
}

int main() {
  double* out; long long* clocks;
  (void)hipMalloc(&out, sizeof(double) * 256 * 8 * 256);
  (void)hipMalloc(&clocks, sizeof(long long) * 2 * 256 * 8);
  for (int w : {1, 2}) {
    run<2, false, false>(w, out, clocks); run<2, true, false>(w, out, clocks);
    run<2, false, true>(w, out, clocks); run<2, true, true>(w, out, clocks);
    run<8, false, false>(w, out, clocks); run<8, true, false>(w, out, clocks);
    run<8, false, true>(w, out, clocks); run<8, true, true>(w, out, clocks);
  }
  return 0;
}
