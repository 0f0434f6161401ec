// lpamd_ratio_test_noalloc: the MAX_COST ratio test (reference lp_simplex_tableau.cc:207-227) on
// device vectors the caller owns, for callers that cannot allocate (code under HIP graph capture,
// tight serving loops, harnesses that forbid hipMalloc).  Same arithmetic and tie-break as
// k_ratio_partial + the fold of k_pivot_finalize: rows with (float)a <= -1e-6f compete with
// ratio b / (-a), the smaller ratio wins, the lower row on ties.
//
// The two-stage reduction needs one (ratio, row) partial per workgroup.  The engine keeps those in
// memory it allocated at lpamd_create; here they live in a file-scope device array, so the call
// itself allocates nothing.  Every GPU holds its own copy of the array at its own address: the
// address is looked up once per device with hipGetSymbolAddress and only that pointer is ever
// handed to kernels.  All calls on one GPU share the array, so calls that overlap on one GPU
// (different streams or host threads) race; calls on one stream are ordered and safe.
#pragma once

#include <mutex>

namespace lpamd {

constexpr int64_t kNoAllocMaxRows = (int64_t)1 << 24;   // 16 Mi rows: 64 Ki rows per partial at most
constexpr int kNoAllocPartials = 1024;

struct RatioPartial {
  double ratio;
  int64_t row;
};

// partial results of lpamd_ratio_test_noalloc; 16 KiB per GPU until the process ends
static __device__ __align__(256) RatioPartial scratch[kNoAllocPartials];

__global__ __launch_bounds__(kPartialThreads) void k_noalloc_ratio_partial(
    const double* __restrict__ col, const double* __restrict__ rhs, int64_t m, RatioPartial* __restrict__ part) {
  __shared__ double s_v[kPartialThreads / kWave];
  __shared__ int64_t s_i[kPartialThreads / kWave];
  double bv = DBL_MAX;
  int64_t bi = -1;
  for (int64_t i = (int64_t)blockIdx.x * kPartialThreads + threadIdx.x; i < m; i += (int64_t)gridDim.x * kPartialThreads) {
    const double a = col[i];
    if ((float)a > -1e-6f) continue;   // skips _IsZero and _IsNonNegative entries alike
    const double ratio = rhs[i] / (-a);
    if (bv > ratio) { bv = ratio; bi = i; }
  }
  block_arg<false, kPartialThreads>(bv, bi, s_v, s_i);
  if (threadIdx.x == 0) { part[blockIdx.x].ratio = bv; part[blockIdx.x].row = bi; }
}

__global__ __launch_bounds__(kPartialThreads) void k_noalloc_ratio_fold(
    const RatioPartial* __restrict__ part, int n_part, int64_t* __restrict__ row_out, double* __restrict__ ratio_out) {
  __shared__ double s_v[kPartialThreads / kWave];
  __shared__ int64_t s_i[kPartialThreads / kWave];
  double bv = DBL_MAX;
  int64_t bi = -1;
  for (int k = threadIdx.x; k < n_part; k += kPartialThreads) take_min(bv, bi, part[k].ratio, part[k].row);
  block_arg<false, kPartialThreads>(bv, bi, s_v, s_i);
  if (threadIdx.x == 0) {
    *row_out = bi;
    if (ratio_out) *ratio_out = bi < 0 ? 0.0 : bv;
  }
}

}  // namespace lpamd

namespace {

constexpr int kNoAllocMaxDevices = 64;
std::mutex g_noalloc_mutex;
lpamd::RatioPartial* g_noalloc_scratch[kNoAllocMaxDevices] = {};   // per device: where `scratch` lives on that GPU

// Address of the scratch array on the current device, looked up on the first call per device.
int noalloc_scratch(lpamd::RatioPartial** out) {
  int dev = -1;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= kNoAllocMaxDevices) return set_err(LPAMD_E_ARG, "ratio_test_noalloc: device ordinal %d not supported", dev);
  std::lock_guard<std::mutex> lock(g_noalloc_mutex);
  if (!g_noalloc_scratch[dev]) {
    void* p = nullptr;
    HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(lpamd::scratch)));
    if (!p) return set_err(LPAMD_E_HIP, "ratio_test_noalloc: hipGetSymbolAddress returned a null address");
    g_noalloc_scratch[dev] = static_cast<lpamd::RatioPartial*>(p);
  }
  *out = g_noalloc_scratch[dev];
  return LPAMD_OK;
}

}  // namespace

extern "C" int lpamd_ratio_test_noalloc(const double* col_dev, const double* rhs_dev, int64_t m, int64_t* row_out_dev,
                                        double* ratio_out_dev, void* hip_stream) {
  if (!col_dev || !rhs_dev || !row_out_dev) return set_err(LPAMD_E_ARG, "ratio_test_noalloc: null buffer");
  if (m <= 0 || m > lpamd::kNoAllocMaxRows)
    return set_err(LPAMD_E_ARG, "ratio_test_noalloc: m=%lld outside 1..%lld", (long long)m, (long long)lpamd::kNoAllocMaxRows);
  lpamd::RatioPartial* part = nullptr;
  LP_TRY(noalloc_scratch(&part));
  hipStream_t s = (hipStream_t)hip_stream;   // NULL: the device's default stream
  const int n_part = (int)std::min<int64_t>(lpamd::kNoAllocPartials, (m + lpamd::kPartialThreads - 1) / lpamd::kPartialThreads);
  // every partial the fold reads is written by the first kernel, so nothing has to be cleared between calls
  hipLaunchKernelGGL(lpamd::k_noalloc_ratio_partial, dim3(n_part), dim3(lpamd::kPartialThreads), 0, s, col_dev, rhs_dev, m, part);
  LP_TRY(check_launch("k_noalloc_ratio_partial"));
  hipLaunchKernelGGL(lpamd::k_noalloc_ratio_fold, dim3(1), dim3(lpamd::kPartialThreads), 0, s, part, n_part, row_out_dev, ratio_out_dev);
  return check_launch("k_noalloc_ratio_fold");
}
