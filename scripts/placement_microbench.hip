// Does the placement of a 17 GB working set inside HBM change the read-modify-write rate?
// One large allocation; the kernel streams K regions of 17.2/K GB each whose start offsets are
// given on the command line or spread evenly.  Build:
//   hipcc -O3 --offload-arch=gfx950 -o bin_scratch/placement scripts/placement_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Regions { char* base[16]; };

// each workgroup streams 64 KiB chunks (256 lanes x 16 B x 16), all loads before the stores;
// consecutive chunks rotate over the regions
__global__ __launch_bounds__(256, 2) void k_rmw(Regions rg, int k, long chunks_per_region, double f) {
  for (long c = blockIdx.x; c < k * chunks_per_region; c += gridDim.x) {
    double2* p = (double2*)rg.base[c % k] + (c / k) * 4096 + threadIdx.x;
    double2 v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) { v[u].x = __builtin_nontemporal_load(&p[u * 256].x); v[u].y = __builtin_nontemporal_load(&p[u * 256].y); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 16; u++) { __builtin_nontemporal_store(v[u].x * f + 1.0, &p[u * 256].x); __builtin_nontemporal_store(v[u].y * f + 1.0, &p[u * 256].y); }
  }
}

// read-only variant: every lane sums what it loads (one store per workgroup pass keeps the loads alive)
__global__ __launch_bounds__(256, 2) void k_ro(Regions rg, int k, long chunks_per_region, double* sink) {
  double acc = 0.0;
  for (long c = blockIdx.x; c < k * chunks_per_region; c += gridDim.x) {
    const double2* p = (const double2*)rg.base[c % k] + (c / k) * 4096 + threadIdx.x;
    double2 v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) { v[u].x = __builtin_nontemporal_load(&p[u * 256].x); v[u].y = __builtin_nontemporal_load(&p[u * 256].y); }
#pragma unroll
    for (int u = 0; u < 16; u++) acc += v[u].x + v[u].y;
  }
  if (acc == 123.456) sink[0] = acc;
}

// out-of-place: read 64 KiB chunks from src, write them to dst
__global__ __launch_bounds__(256, 2) void k_copy(const double2* __restrict__ src, double2* __restrict__ dst, long chunks, double f) {
  for (long c = blockIdx.x; c < chunks; c += gridDim.x) {
    const double2* p = src + c * 4096 + threadIdx.x;
    double2* q = dst + c * 4096 + threadIdx.x;
    double2 v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) { v[u].x = __builtin_nontemporal_load(&p[u * 256].x); v[u].y = __builtin_nontemporal_load(&p[u * 256].y); }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 16; u++) { __builtin_nontemporal_store(v[u].x * f + 1.0, &q[u * 256].x); __builtin_nontemporal_store(v[u].y * f + 1.0, &q[u * 256].y); }
  }
}

int main(int argc, char** argv) {
  const double total_gb = 250.0;
  const size_t total = (size_t)(total_gb * 1e9);
  size_t work = (size_t)17200 << 20;
  char* base = nullptr;
  CK(hipMalloc(&base, total));
  CK(hipMemset(base, 0, total));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto run = [&](const char* label, int k, const std::vector<double>& offs_gb) -> int {
    Regions rg{};
    const size_t region = (work / k) >> 16 << 16;
    for (int i = 0; i < k; i++) {
      size_t off = ((size_t)(offs_gb[i] * 1e9) >> 21) << 21;
      if (off + region > total) { printf("%s: region %d out of range\n", label, i); return 0; }
      rg.base[i] = base + off;
    }
    const long chunks = (long)(region / 65536);
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL(k_rmw, dim3(8192), dim3(256), 0, 0, rg, k, chunks, 0.5);
    CK(hipEventRecord(e0));
    const int reps = 5;
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_rmw, dim3(8192), dim3(256), 0, 0, rg, k, chunks, 0.5);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = 2.0 * (double)region * k;
    printf("%-44s %8.3f ms  %7.1f GB/s\n", label, ms / reps, bytes / (ms / reps * 1e-3) / 1e9);
    return 0;
  };
  if (argc > 1 && std::string(argv[1]) == "copy") {   // 17.2 GB read from one place, written to another
    const double pairs[][2] = {{0, 20}, {0, 40}, {0, 100}, {0, 150}, {100, 0}, {100, 120}, {150, 0}, {0, 215}};
    for (auto& pr : pairs) {
      const double2* src = (const double2*)(base + (((size_t)(pr[0] * 1e9) >> 21) << 21));
      double2* dst = (double2*)(base + (((size_t)(pr[1] * 1e9) >> 21) << 21));
      const long chunks = (long)(work / 65536);
      for (int w = 0; w < 2; w++) hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, 0, src, dst, chunks, 0.5);
      CK(hipEventRecord(e0));
      for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, 0, src, dst, chunks, 0.5);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("copy 17.2 GB from %5.0f GB to %5.0f GB  %8.3f ms  %7.1f GB/s (read + write)\n", pr[0], pr[1], ms / 5, 2.0 * work / (ms / 5 * 1e-3) / 1e9);
    }
    return 0;
  }
  if (argc > 1 && std::string(argv[1]) == "readonly") {   // read-only streams, 17.2 GB, one class vs two
    auto run_ro = [&](const char* label, int k, const std::vector<double>& offs_gb) -> int {
      Regions rg{};
      const size_t region = (work / k) >> 16 << 16;
      for (int i = 0; i < k; i++) rg.base[i] = base + (((size_t)(offs_gb[i] * 1e9) >> 21) << 21);
      const long chunks = (long)(region / 65536);
      for (int w = 0; w < 2; w++) hipLaunchKernelGGL(k_ro, dim3(8192), dim3(256), 0, 0, rg, k, chunks, (double*)base);
      CK(hipEventRecord(e0));
      for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k_ro, dim3(8192), dim3(256), 0, 0, rg, k, chunks, (double*)base);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("read-only %-34s %8.3f ms  %7.1f GB/s\n", label, ms / 5, (double)region * k / (ms / 5 * 1e-3) / 1e9);
      return 0;
    };
    run_ro("1 region at 0", 1, {0});
    run_ro("1 region at 100", 1, {100});
    run_ro("2 regions 0,36", 2, {0, 36});
    run_ro("2 regions 0,100", 2, {0, 100});
    run_ro("2 regions 0,150", 2, {0, 150});
    run_ro("4 regions 0,80,150,215", 4, {0, 80, 150, 215});
    return 0;
  }
  if (argc > 1 && std::string(argv[1]) == "sweep") {   // two 2 GiB regions: one fixed, one swept over the allocation
    work = (size_t)4096 << 20;
    const double fixed = argc > 2 ? atof(argv[2]) : 0.0;
    for (double o = 0; o < total_gb - 3; o += 6) {
      if (o > fixed - 2.2 && o < fixed + 2.2) continue;
      char label[64];
      snprintf(label, sizeof label, "2 x 2 GiB at %.0f and %.0f GB", fixed, o);
      run(label, 2, {fixed, o});
    }
    return 0;
  }
  if (argc > 1) {   // explicit offsets in GB
    std::vector<double> offs;
    for (int i = 1; i < argc; i++) offs.push_back(atof(argv[i]));
    return run("explicit offsets", (int)offs.size(), offs);
  }
  run("1 region at 0", 1, {0});
  run("1 region at 60", 1, {60});
  run("1 region at 120", 1, {120});
  run("1 region at 180", 1, {180});
  run("1 region at 230", 1, {230});
  run("2 regions 0,120", 2, {0, 120});
  run("2 regions 0,72", 2, {0, 72});
  run("2 regions 0,36", 2, {0, 36});
  run("2 regions 60,180", 2, {60, 180});
  run("4 regions 0,60,120,180", 4, {0, 60, 120, 180});
  run("4 regions 0,36,72,108", 4, {0, 36, 72, 108});
  run("4 regions 0,72,144,216", 4, {0, 72, 144, 216});
  run("8 regions every 30", 8, {0, 30, 60, 90, 120, 150, 180, 210});
  run("8 regions every 36", 8, {0, 36, 72, 108, 144, 180, 216, 240});
  run("16 regions every 15", 16, {0, 15, 30, 45, 60, 75, 90, 105, 120, 135, 150, 165, 180, 195, 210, 225});
  return 0;
}
