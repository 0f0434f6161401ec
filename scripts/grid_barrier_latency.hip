// What a grid-wide barrier costs inside ONE kernel on this GPU, against the kernel boundary it would replace
// (scripts/trip_latency.hip: 2.5-2.9 us per dependent launch).  A persistent kernel of G co-resident workgroups runs
// K rounds; in every round each workgroup publishes a 16-byte partial result, all workgroups meet at a barrier, and
// every workgroup folds all G partials (what the set-up kernels of a chain of pivots do between their phases).
// Variants:
//   flat / hier    one arrival counter, or one per XCD (blockIdx % 8) whose last arriver reports to a global one
//   fence / sc1    data made visible with agent-scope release/acquire (write-back + invalidate of the XCD's L2), or
//                  with agent-scope RELAXED atomic stores and loads on the data itself (the accesses bypass the
//                  non-coherent levels one by one, no whole-cache operation) and a plain wait for the stores
// Every round checks that the folded value is the one expected (a stale partial shows as an error count).
//   hipcc -O3 --offload-arch=gfx950 -o bin_scratch/grid_barrier_latency scripts/grid_barrier_latency.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Sync {
  unsigned int global;        // arrivals (flat) or XCD groups that have arrived (hier), monotonic
  unsigned int pad[31];
  unsigned int xcd[8][32];    // one line per XCD group
};

__device__ __forceinline__ unsigned int ld_relaxed(const unsigned int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// All threads of all G workgroups; `round` counts from 1.  kFence: release/acquire at agent scope.
template <bool kHier, bool kFence>
__device__ __forceinline__ void grid_barrier(Sync* s, unsigned int round, unsigned int G, unsigned int* spins) {
  __syncthreads();
  if (threadIdx.x == 0) {
    if (kFence) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // L2 write-back
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's published stores have been acknowledged
    if (kHier) {
      const unsigned int x = blockIdx.x & 7u, n_x = (G + 7u - x) / 8u;   // workgroups of this group
      const unsigned int before = __hip_atomic_fetch_add(&s->xcd[x][0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (before + 1u == round * n_x) __hip_atomic_fetch_add(&s->global, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned int groups = G < 8u ? G : 8u;
      unsigned int n = 0;
      while (ld_relaxed(&s->global) < round * groups && ++n < (1u << 24)) __builtin_amdgcn_s_sleep(1);
      *spins += n;
    } else {
      __hip_atomic_fetch_add(&s->global, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned int n = 0;
      while (ld_relaxed(&s->global) < round * G && ++n < (1u << 24)) __builtin_amdgcn_s_sleep(1);
      *spins += n;
    }
    if (kFence) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

template <bool kHier, bool kFence, bool kFold>
__global__ __launch_bounds__(256) void k_rounds(Sync* s, unsigned long long* part, int rounds, unsigned int* errors, unsigned int* spin_total) {
  __shared__ unsigned long long s_w[4];
  const unsigned int G = gridDim.x;
  unsigned int spins = 0, bad = 0;
  for (int k = 1; k <= rounds; k++) {
    unsigned long long* mine = part + (size_t)(k & 1) * 2 * G;   // two sets of partials take turns
    if (threadIdx.x == 0) {
      const unsigned long long v = (unsigned long long)k * 1000003ull + blockIdx.x;
      if (kFence) { mine[2 * blockIdx.x] = v; mine[2 * blockIdx.x + 1] = ~v; }
      else {
        __hip_atomic_store(&mine[2 * blockIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&mine[2 * blockIdx.x + 1], ~v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    grid_barrier<kHier, kFence>(s, (unsigned int)k, G, &spins);
    if (kFold) {   // every workgroup folds all partials: maximum of the first words
      unsigned long long best = 0;
      for (unsigned int q = threadIdx.x; q < G; q += 256) {
        unsigned long long a, b;
        if (kFence) { a = mine[2 * q]; b = mine[2 * q + 1]; }
        else {
          a = __hip_atomic_load(&mine[2 * q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          b = __hip_atomic_load(&mine[2 * q + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (b != ~a) bad++;
        best = a > best ? a : best;
      }
      for (int off = 32; off > 0; off >>= 1) { const unsigned long long o = __shfl_xor(best, off, 64); best = o > best ? o : best; }
      if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = best;
      __syncthreads();
      unsigned long long w = s_w[0];
      for (int q = 1; q < 4; q++) w = s_w[q] > w ? s_w[q] : w;
      if (w != (unsigned long long)k * 1000003ull + (G - 1)) bad++;
      __syncthreads();
    }
  }
  if (bad) atomicAdd(errors, bad);
  if (threadIdx.x == 0 && blockIdx.x == 0) *spin_total = spins;
}

template <bool kHier, bool kFence, bool kFold>
static void run(const char* name, int G, int rounds) {
  Sync* s;
  unsigned long long* part;
  unsigned int *errors, *spin;
  CK(hipMalloc(&s, sizeof(Sync)));
  CK(hipMalloc(&part, sizeof(unsigned long long) * 4 * G));
  CK(hipMalloc(&errors, 4));
  CK(hipMalloc(&spin, 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float best = 1e30f;
  unsigned int h_err = 0, h_spin = 0;
  for (int rep = 0; rep < 4; rep++) {
    CK(hipMemset(s, 0, sizeof(Sync)));
    CK(hipMemset(part, 0, sizeof(unsigned long long) * 4 * G));
    CK(hipMemset(errors, 0, 4));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_rounds<kHier, kFence, kFold>), dim3(G), dim3(256), 0, 0, s, part, rounds, errors, spin);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
    unsigned int er = 0;
    CK(hipMemcpy(&er, errors, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&h_spin, spin, 4, hipMemcpyDeviceToHost));
    h_err += er;
  }
  printf("%-22s G %4d: %6.2f us per round  (stale or torn partials: %u, polls of workgroup 0 per round: %.1f)\n", name, G,
         best * 1000.0 / rounds, h_err, (double)h_spin / rounds);
  CK(hipFree(s)); CK(hipFree(part)); CK(hipFree(errors)); CK(hipFree(spin));
}

int main() {
  const int rounds = 400;
  for (int G : {32, 64, 128, 256, 512, 768}) {
    run<false, true, false>("flat fence", G, rounds);
    run<false, false, false>("flat sc1", G, rounds);
    run<true, false, false>("hier sc1", G, rounds);
    run<false, true, true>("flat fence + fold", G, rounds);
    run<false, false, true>("flat sc1 + fold", G, rounds);
    run<true, false, true>("hier sc1 + fold", G, rounds);
  }
  return 0;
}
