// Host side of the C ABI declared in include/lpamd.h: owns the device-resident
// tableau shard, enqueues the kernels of lpamd_kernels.hip.h on one HIP stream
// and keeps the pivot loop on the device (the host only reads the 64-byte
// PivotState back once per batch of pivots).
#include "lpamd_kernels.hip.h"
#include "lpamd_prbound_kernels.hip.h"
#include "lpamd_fused_kernels.hip.h"
#include "lpamd_prbtail_kernels.hip.h"
#include "lpamd_shardstep_kernels.hip.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

#include "lpamd.h"

using namespace lpamd;

namespace {

thread_local char g_err[512] = "";

int set_err(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess)                                                                      \
      return set_err(LPAMD_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

#define LP_TRY(expr)            \
  do {                          \
    int _rc = (expr);           \
    if (_rc != LPAMD_OK) return _rc; \
  } while (0)

// host copies of the reference's float-typed predicates (reference base.h:61-66)
inline bool h_is_zero(double x) { return std::fabs((float)x) < 1e-6f; }
inline bool h_is_neg(double x) { return (float)x <= -1e-6f; }
inline bool h_is_nonneg(double x) { return (float)x > -1e-6f; }
inline double h_flushz(double v) { return std::fabs(v) < (double)1e-6f ? 0.0 : v; }

int status_to_result(int32_t st) { return st == ST_RUNNING ? LPAMD_LIMIT : st - 1; }

}  // namespace

struct lpamd_engine {
  Shape sh{};
  int32_t ncl = 0;          // nloc + 1
  int32_t n_slabs = 0;
  int32_t update_grid = 2048;   // workgroups of the rank-1 update
  bool update_nt = false;       // nontemporal loads/stores: shards beyond the Infinity Cache
  int32_t update_unroll_cap = 16;
  bool update_prec = true;      // load col' together with the tableau batch
  bool update_prec_fused = true;
  bool fuse_pr_enabled = true;  // max-reduction pricing of the next iteration inside the update
  // Two-launch MAX_COST loop (k_pivot_setup_small + k_update) on one GPU while columns are short
  bool two_launch = false;
  int price_buf = 0;            // which half of part_v/part_i holds the fresh pricing partials (two-launch loop)
  int update_slot = -1;         // counter slot handed to the next k_update launch (-1: three-launch loop)
  // Several pivots per pass over the tableau (MAX_COST on one GPU, columns too long for the two-launch loop):
  // colp holds chain_depth col' vectors (lda apart), rowp chain_depth row' vectors (rstride apart)
  int32_t chain_depth = 1;      // pivots a pass may apply (1: one pivot per pass)
  int32_t mr_depth = 1;         // the same for MAX_REDUCTION passes priced from bounds (their set-up costs four times a MAX_COST step's: deeper passes pay)
  int32_t alloc_depth = 1;      // steps the per-step buffers hold
  int32_t loop_depth = 1;       // depth of the device loop that ran last
  int32_t slab_shape = 0;       // k_update_slab variant (0: k_update_chain)
  int32_t slab_grid = 4096;
  int32_t deep_batch = 0;       // > 0: k_update_deep with that many columns per batch serves the passes of more than 16 steps
  bool deep_any_depth = false;  // ... and the shorter ones too (LPAMD_DEEP, experiments)
  int32_t deep_grid = 0;        // workgroups of k_update_deep (0: sized from the shard, see launch_deep)
  int64_t rstride = 0;
  // steps 1.. of a MAX_COST chain set up by one launch of resident workgroups (k_chain_fused; one GPU)
  bool fused = false;
  bool fused_pending = false;   // the pivot count lacks the fused steps until k_chain_list has run
  int32_t fused_rows = 0, fused_cols = 0;   // row / column workgroups
  int32_t fused_threads = 256;              // threads per workgroup (256, 512 or 1024)
  bool fused_reg = false;                   // one row / column per thread, its history in registers
  uint64_t* rec_ratio = nullptr;  // [2 parities][kFusedReplicas copies][fused_region_words(fused_rows, kFusedRatioWords)]
  uint64_t* rec_price = nullptr;  // [2][kFusedReplicas][fused_region_words(fused_cols, kFusedPriceWords)]
  uint32_t fused_epoch = 0;
  ChainStep* steps = nullptr;   // [kMaxChain]
  double* colv = nullptr;       // entering column of the next step as the steps before it will leave it
  // MAX_REDUCTION pricing from bounds (lpamd_prbound_kernels.hip.h): lets those pivots chain as well
  bool pr_bounds = false;
  PrbInfo* prb_info = nullptr;
  double* prb_c = nullptr;       // [kMaxChain * kPrbRows] col'_s at the rows of the bound
  double* prb_cache = nullptr;   // [kPrbRows * nloc] rows of the bound, one slot each, row-major
  bool prb_use_cache = true;
  double* prb_v = nullptr;       // [3 * kMaxPartials] partials: smallest right-hand sides, largest bounds
  int64_t* prb_i = nullptr;      // [3 * kMaxPartials]
  double* prb_ep_v = nullptr;    // [kPrbSliced * kPrbSlices] ratio partials of a short list evaluated in slices
  int64_t* prb_ep_i = nullptr;
  int32_t prb_rows = kPrbRows;   // rows the bounds are taken from
  bool prb_carry = true;         // the cache slots outlive a pass (PrbInfo::cache_iter)
  uint64_t tab_version = 1, prb_cache_version = 0;   // bumped by everything that writes the tableau; what the slots were last good for
  int32_t prb_bound_blocks = 0;  // > 0: at most that many workgroups for k_prb_bounds (each walks several tiles)
  int32_t prb_renew = 4;         // a cache slot this many steps behind is brought up to date by the pricing that reads it
  double* prb_ub = nullptr;      // [ncl] bound per candidate column
  uint8_t* prb_ucand = nullptr;  // [ncl] candidate flags
  int32_t* prb_elist = nullptr;  // [ncl] columns to evaluate exactly
  double* prb_escore = nullptr;  // [ncl] exact score ...
  int64_t* prb_erow = nullptr;   // [ncl] ... and row of the evaluated columns
  // steps 1.. of a pass: everything behind the bounds in one launch of resident workgroups (k_prb_tail; one GPU)
  bool prb_tail = false;
  int32_t tail_rows = 0, tail_cols = 0;   // row / column workgroups
  double* prb_top_u = nullptr;   // [2 * kMaxPartials] per workgroup of k_prb_bounds: largest and second largest bound
  int32_t* prb_top_l = nullptr;  // [kMaxPartials] ... and the column of the largest
  uint64_t* rec_tail = nullptr;  // [kFusedReplicas copies] of the slice records, then of the records of the second round
  uint32_t tail_epoch = 0;
  // column shards, MAX_COST: set-up of a chained step and this rank's slot for the next one in one launch (k_shard_step)
  bool shard_step = false;
  uint64_t* rec_shard = nullptr; // [kFusedReplicas copies][fused_region_words(n_price_part, kShardRecWords)]
  uint32_t shard_epoch = 0;
  PassInfo* pass = nullptr;     // what the pass kernel has to do (k_chain_list)
  int32_t* list_full = nullptr; // [ncl] columns every step of the chain touches ...
  int32_t* list_part = nullptr; // [ncl] ... and columns some steps leave alone
  double* rsl_full = nullptr;   // [ncl * chain_depth] row' values of the listed columns, one run per column
  uint64_t* ptr_full = nullptr; // [ncl + 8] addresses of the columns of list_full (k_update_deep)
  double* rsl_part = nullptr;
  int64_t slot_counter = 0;
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  // device memory
  Tab T{};                  // two buffers: even / odd local columns
  int64_t half_cols[2] = {0, 0};
  bool placement_split = false;   // the halves were measured to lie in different HBM placement classes
  double* obj = nullptr;
  uint8_t* is_basic = nullptr;
  int64_t* basis = nullptr;
  double* colp = nullptr;
  double* rowp = nullptr;   // also scratch for row gathers / column values
  int32_t* active = nullptr;
  double* cmin = nullptr;       // max-reduction pricing: per-column minimum ratio ...
  int64_t* crow = nullptr;      // ... its row (-1 none, -2 not a candidate) ...
  int64_t* pr_stamp = nullptr;  // ... and the pivot count for which the pair is valid
  double* rhs_next = nullptr;   // [chain_depth * lda] right-hand side after pivot s of the pass, s = 0..
  uint8_t* cand_next = nullptr; // pricing candidates after the pivot being applied
  Candidate* cand = nullptr;
  Candidate* cand_la = nullptr;   // look-ahead exchange: winner of the next pivot, written on the exchange stream
  int32_t pending_steps = 0;      // pivots set up and waiting for the pass (lpamd_lookahead_setup + lpamd_chain_step calls)
  bool update_pending = false;    // a pivot has been set up (k_pivot_finalize) and its rank-1 update is not enqueued yet
  double* part_v = nullptr;     // [2 * kMaxPartials]: pricing partials, then ratio partials
  int64_t* part_i = nullptr;
  int32_t n_price_part = 1, n_ratio_part = 1, n_row_blocks = 1;
  int64_t apply_since_pull = 0;        // pivots enqueued since the host last read the device state
  bool price_partials_fresh = false;   // part_v/part_i[0..n_price_part) describe the current objective row and basis
  PivotState* st = nullptr;
  int64_t* trace = nullptr;
  int64_t trace_cap = 0;
  // host mirrors
  PivotState h_st{};
  bool loaded = false;
  // Basic columns are no longer unit columns (PivotState::dirty): pivots are chosen by the loop that scans like the reference
  bool dirty_basis = false;
  unsigned long long* first_basic = nullptr;   // [m] k_basic_scan_rows
  long long* basis_disc = nullptr;             // [m] k_basis_discovery
  bool dense_first = false;   // host view of PivotState::dense_next: the next pivot must be a single one that reaches every column
  // profiling of the update kernel
  bool profiling = false;
  std::vector<hipEvent_t> ev;
  size_t ev_used = 0;
  double prof_ms = 0.0;
  int64_t prof_launches = 0;
  int64_t prof_active = 0;
};

namespace {

int64_t round_up(int64_t x, int64_t q) { return (x + q - 1) / q * q; }

bool owns(const lpamd_engine* e, int64_t gcol) {
  if (gcol == e->sh.N) return true;  // RHS replica
  return ordinal_of(gcol, e->sh.N) % e->sh.world == e->sh.rank;
}
int64_t local_of(const lpamd_engine* e, int64_t gcol) {
  if (gcol == e->sh.N) return e->sh.nloc;
  return ordinal_of(gcol, e->sh.N) / e->sh.world;
}
int64_t global_of(const lpamd_engine* e, int64_t l) {
  if (l == e->sh.nloc) return e->sh.N;
  return gcol_of(l * e->sh.world + e->sh.rank, e->sh.N);
}

int ensure_trace(lpamd_engine* e, int64_t need) {
  if (need <= e->trace_cap) return LPAMD_OK;
  int64_t cap = std::max<int64_t>(4096, e->trace_cap);
  while (cap < need) cap *= 2;
  int64_t* nt = nullptr;
  HIP_TRY(hipMalloc(&nt, sizeof(int64_t) * 2 * cap));
  if (e->trace) {
    HIP_TRY(hipMemcpyAsync(nt, e->trace, sizeof(int64_t) * 2 * e->trace_cap, hipMemcpyDeviceToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    HIP_TRY(hipFree(e->trace));
  }
  e->trace = nt;
  e->trace_cap = cap;
  return LPAMD_OK;
}

int check_launch(const char* what);
int run_loop_device(lpamd_engine* e, int strategy, int64_t budget, int32_t* result);

// ---- tableau allocation with HBM placement ------------------------------------
// Measured on MI355X (scripts/placement_microbench.hip, profiles/): the address space of
// the HBM falls into placement classes of 32 GiB units, and a read-modify-write stream whose
// working set lies inside one class sustains about 5.4 TB/s against 6.4 TB/s when it spans
// two.  Where hipMalloc puts a buffer cannot be chosen, but it can be measured: the shard is
// kept as two buffers (even / odd local columns, which the update touches at the same
// time), and the second one is re-allocated, with the rejected candidates held back as
// spacers, until a short probe shows that the pair streams faster than the first buffer
// alone.  Shards of 64 GiB and more span classes anyway; shards that fit the Infinity Cache
// do not care.  LPAMD_PLACEMENT=0 takes the first allocation.
__global__ __launch_bounds__(256, 2) void k_placement_probe(double2* __restrict__ a, double2* __restrict__ b, long chunks_per_side) {
  for (long c = blockIdx.x; c < 2 * chunks_per_side; c += gridDim.x) {
    double2* p = (c & 1 ? b : a) + (c >> 1) * 4096 + threadIdx.x;   // 64 KiB chunks, alternating sides
    double2 v[16];
#pragma unroll
    for (int u = 0; u < 16; u++) {
      v[u].x = __builtin_nontemporal_load(&p[u * 256].x);
      v[u].y = __builtin_nontemporal_load(&p[u * 256].y);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 16; u++) {
      __builtin_nontemporal_store(v[u].x, &p[u * 256].x);
      __builtin_nontemporal_store(v[u].y, &p[u * 256].y);
    }
  }
}

// milliseconds of one probe pass over `bytes_per_side` of each side
int placement_probe_ms(lpamd_engine* e, double* a, double* b, size_t bytes_per_side, float* ms_out) {
  hipEvent_t e0, e1;
  HIP_TRY(hipEventCreate(&e0));
  HIP_TRY(hipEventCreate(&e1));
  const long chunks = (long)(bytes_per_side / 65536);
  hipLaunchKernelGGL(k_placement_probe, dim3(4096), dim3(256), 0, e->stream, (double2*)a, (double2*)b, chunks);
  HIP_TRY(hipEventRecord(e0, e->stream));
  for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k_placement_probe, dim3(4096), dim3(256), 0, e->stream, (double2*)a, (double2*)b, chunks);
  HIP_TRY(hipEventRecord(e1, e->stream));
  HIP_TRY(hipEventSynchronize(e1));
  HIP_TRY(hipEventElapsedTime(ms_out, e0, e1));
  *ms_out /= 3;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return check_launch("k_placement_probe");
}

int allocate_tableau(lpamd_engine* e) {
  // Column pitch of the tableau: the rows rounded up (LPAMD_LDA_PAD=<rows>: that many more; experiments).  The kernels that
  // set a pivot up read ONE row of the column-major tableau, a sector of every column, and with columns a large power of
  // two apart those sectors fall on few memory channels: at m=32768 n=65536 the gather of 98 k sectors takes 1.6 .. 4.7 us,
  // 2.7 .. 3.1 us with the pitch 4 KiB longer.  The pass kernel, however, streams 0.8 % slower over columns that are not
  // 256 KiB apart (3.7 % at m=4096 n=8192), which costs more than the gathers gain, so the default is no padding.  Padding
  // rows are never read or written: every loop over a column runs to sh.lda.
  int64_t pad = 0;
  if (const char* env = getenv("LPAMD_LDA_PAD")) pad = round_up(std::max(0, atoi(env)), kSlabRows);
  e->T.lda = e->sh.lda + pad;
  e->half_cols[0] = (e->ncl + 1) / 2;
  e->half_cols[1] = e->ncl / 2;
  size_t bytes[2];
  for (int h = 0; h < 2; h++) bytes[h] = sizeof(double) * (size_t)e->T.lda * (size_t)e->half_cols[h];
  hipError_t err = hipMalloc(&e->T.half[0], bytes[0]);
  if (err != hipSuccess)
    return set_err(LPAMD_E_HIP, "hipMalloc of %.2f GB for the tableau shard failed: %s", bytes[0] / 1e9, hipGetErrorString(err));
  if (bytes[1] == 0) return LPAMD_OK;
  bool probe = bytes[1] >= ((size_t)1 << 30) && bytes[1] < ((size_t)32 << 30);
  if (const char* env = getenv("LPAMD_PLACEMENT")) probe = probe && atoi(env) != 0;
  std::vector<double*> spacers;
  double* chosen = nullptr;
  if (probe) {
    const size_t side = (size_t)512 << 20;   // 512 MiB per side: beyond the 256 MiB Infinity Cache
    float alone = 0.f;
    LP_TRY(placement_probe_ms(e, e->T.half[0], e->T.half[0] + side / sizeof(double), side, &alone));
    // Every rejected candidate stays allocated, so the next one lands further on; small shards
    // add a spacer so that an attempt advances by at least 12 GB of the 32 GiB units.
    const size_t step = (size_t)12 << 30;
    for (int attempt = 0; attempt < 16 && !chosen; attempt++) {
      // the search never takes the device near its limit: a candidate, its spacer and 4 GiB to spare must fit
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); break; }
      if (free_b < std::max(bytes[1], step) + ((size_t)4 << 30)) break;
      double* cand = nullptr;
      if (hipMalloc(&cand, bytes[1]) != hipSuccess) { (void)hipGetLastError(); break; }
      float pair = 0.f;
      const int rc = placement_probe_ms(e, e->T.half[0], cand, side, &pair);
      if (rc != LPAMD_OK) { (void)hipFree(cand); for (double* sp : spacers) (void)hipFree(sp); return rc; }
      if (getenv("LPAMD_PLACEMENT_VERBOSE"))
        fprintf(stderr, "lpamd placement: attempt %d, %.2f GB halves, probe alone %.3f ms, pair %.3f ms\n", attempt, bytes[1] / 1e9, alone, pair);
      if (pair < 0.93f * alone) { chosen = cand; e->placement_split = true; break; }
      spacers.push_back(cand);
      if (bytes[1] < step) {
        double* extra = nullptr;
        if (hipMalloc(&extra, step - bytes[1]) == hipSuccess) spacers.push_back(extra);
        else { (void)hipGetLastError(); break; }
      }
    }
    if (!chosen && !spacers.empty()) {   // no better placement found: keep the first candidate
      chosen = spacers.front();
      spacers.erase(spacers.begin());
    }
    for (double* sp : spacers) (void)hipFree(sp);
  }
  if (!chosen) {
    err = hipMalloc(&chosen, bytes[1]);
    if (err != hipSuccess)
      return set_err(LPAMD_E_HIP, "hipMalloc of %.2f GB for the tableau shard failed: %s", bytes[1] / 1e9, hipGetErrorString(err));
  }
  e->T.half[1] = chosen;
  return LPAMD_OK;
}

int push_state(lpamd_engine* e);
int enqueue_chain_update(lpamd_engine* e);

// The chain depth of an HBM-sized shard on one GPU, chosen by measurement.  Up to about 20 steps a pass costs the HBM time
// of one stream over the shard; beyond, the fp64 vector rate (two separately rounded operations per entry and step) adds
// 0.1 .. 0.2 ms per step, and where that sets in differs by a step or two from one machine to the next (shader clock under
// load, HBM placement).  More steps per pass mean more pivots per second, so the deepest chain is taken whose pass still
// streams at kProbeTarget of the 8 TB/s peak: k_chain_list and the pass kernel are timed at every candidate depth on the
// freshly allocated shard (contents do not matter before the upload), with as many columns in the list as a pass
// streams and steps that touch every column.  Three passes per candidate, from the deepest down.
constexpr int kProbeMaxDepth = 24, kProbeMinDepth = 17;
constexpr double kProbeTarget = 0.74, kProbePeakBytesPerMs = 8.0e9;   // the timed runs read 0.5 % below the probe, under rocprofv3 1.8 % below: 0.72 holds there too
int probe_chain_depth(lpamd_engine* e) {
  const int32_t ncl_all = e->ncl;
  const int64_t col_bytes = (int64_t)sizeof(double) * e->sh.lda;
  // as many columns as a pass streams: the non-basic ones and the right-hand side (m of the N variables are basic, and
  // their unit columns are skipped), from the front of the shard like the dense columns of a fresh tableau
  const int64_t dense = std::max<int64_t>(64, e->sh.N - e->sh.m + 2);
  e->ncl = (int32_t)std::min<int64_t>(ncl_all, dense);   // all of them: half of a 137 GB stream read 5 % high (m=65536 n=262144; the probe then takes 1.2 s)
  const bool verbose = getenv("LPAMD_DEPTH_PROBE_VERBOSE") != nullptr;
  // Random entries, as a dense LP has them (the synthetic generator; every probed column is a structural one): a shard of
  // equal numbers streams 7 % faster than one of random numbers at 24 steps per pass (5.77 against 6.16 ms at m=32768
  // n=65536), presumably the vector units' power.  col'_s: copies of a column; row'_s: 0.001 everywhere, so that nothing
  // grows or comes near the small-magnitude threshold over the few hundred steps of the probe.
  e->tab_version++;
  hipLaunchKernelGGL(k_fill_synthetic, dim3((unsigned)e->ncl, (unsigned)((e->sh.lda + 255) / 256)), dim3(256), 0, e->stream, e->T, e->obj,
                     e->is_basic, e->basis, e->sh, (int64_t)e->ncl, (uint64_t)12345);
  LP_TRY(check_launch("k_fill_synthetic"));
  for (int s = 0; s < kProbeMaxDepth; s++)
    HIP_TRY(hipMemcpyAsync(e->colp + (int64_t)s * e->sh.lda, e->T.col(s % e->ncl), sizeof(double) * e->sh.lda, hipMemcpyDeviceToDevice, e->stream));
  hipLaunchKernelGGL(k_fill, dim3((unsigned)((e->rstride * kProbeMaxDepth + 255) / 256)), dim3(256), 0, e->stream, e->rowp, e->rstride * kProbeMaxDepth, 1e-3);
  LP_TRY(check_launch("k_fill"));
  int chosen = 16;
  double chosen_frac = 0.0, fracs[kMaxChain + 1] = {0.0};
  std::vector<ChainStep> hsteps(kMaxChain);
  for (int d = kProbeMaxDepth; d >= kProbeMinDepth; d--) {
    e->chain_depth = d;
    std::memset(hsteps.data(), 0, sizeof(ChainStep) * kMaxChain);
    for (int s = 1; s < d; s++) hsteps[s].done = 1;
    HIP_TRY(hipMemcpyAsync(e->steps, hsteps.data(), sizeof(ChainStep) * kMaxChain, hipMemcpyHostToDevice, e->stream));
    std::memset(&e->h_st, 0, sizeof(PivotState));   // status RUNNING
    LP_TRY(push_state(e));
    double ms = 0.0;
    for (int rep = 0; rep < 3; rep++) {
      HIP_TRY(hipMemsetAsync(e->pass, 0, sizeof(PassInfo), e->stream));
      e->profiling = rep > 0;                        // the first pass warms up
      LP_TRY(enqueue_chain_update(e));
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    for (size_t k = 0; k + 1 < e->ev_used; k += 2) {
      float t = 0.f;
      HIP_TRY(hipEventElapsedTime(&t, e->ev[k], e->ev[k + 1]));   // k_chain_list + the pass, as the benchmark times them
      ms += t;
    }
    ms /= std::max<size_t>(1, e->ev_used / 2);
    e->ev_used = 0;
    e->profiling = false;
    const double frac = 2.0 * col_bytes * e->ncl / std::max(ms, 1e-6) / kProbePeakBytesPerMs;
    if (verbose) fprintf(stderr, "lpamd depth probe: %d steps per pass, %.3f ms over %d columns, %.4f of the HBM peak\n", d, ms, e->ncl, frac);
    fracs[d] = frac;
    if (frac >= kProbeTarget && chosen_frac == 0.0) { chosen = d; chosen_frac = frac; if (!verbose) break; }
  }
  if (chosen_frac == 0.0) {
    // no depth streams at the target (the longest streams stay a little below it at any depth: 0.73 at m=65536 n=262144):
    // the deepest chain within a hundredth of the best depth's fraction
    double best = 0.0;
    for (int d = kProbeMinDepth; d <= kProbeMaxDepth; d++) best = std::max(best, fracs[d]);
    for (int d = kProbeMaxDepth; d >= kProbeMinDepth; d--)
      if (fracs[d] >= best - 0.01) { chosen = d; chosen_frac = fracs[d]; break; }
  }
  if (verbose) fprintf(stderr, "lpamd depth probe: chose %d (%.4f)\n", chosen, chosen_frac);
  e->chain_depth = chosen;
  e->ncl = ncl_all;
  HIP_TRY(hipMemsetAsync(e->steps, 0, sizeof(ChainStep) * kMaxChain, e->stream));
  HIP_TRY(hipMemsetAsync(e->pass, 0, sizeof(PassInfo), e->stream));
  HIP_TRY(hipMemsetAsync(e->colp, 0, sizeof(double) * e->sh.lda * kProbeMaxDepth, e->stream));   // rows beyond m must read zero
  HIP_TRY(hipMemsetAsync(e->rhs_next, 0, sizeof(double) * e->sh.lda * kProbeMaxDepth, e->stream));
  std::memset(&e->h_st, 0, sizeof(PivotState));
  LP_TRY(push_state(e));
  return LPAMD_OK;
}

int pull_state(lpamd_engine* e) {
  e->apply_since_pull = 0;
  HIP_TRY(hipMemcpyAsync(&e->h_st, e->st, sizeof(PivotState), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->dense_first = e->h_st.dense_next != 0;
  return LPAMD_OK;
}
int push_state(lpamd_engine* e) {
  HIP_TRY(hipMemcpyAsync(e->st, &e->h_st, sizeof(PivotState), hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));   // h_st may be edited right after
  return LPAMD_OK;
}
// cmin/crow entries are trusted when pr_stamp equals the pivot count; anything that
// changes the tableau outside the update kernel, or rewinds the count, drops them.
int invalidate_pr(lpamd_engine* e) {
  HIP_TRY(hipMemsetAsync(e->pr_stamp, 0xFF, sizeof(int64_t) * e->ncl, e->stream));
  return LPAMD_OK;
}

void reset_state(lpamd_engine* e) {
  e->price_partials_fresh = false;
  std::memset(&e->h_st, 0, sizeof(PivotState));
  e->h_st.max_iter = std::numeric_limits<int64_t>::max();
}

int check_launch(const char* what) {
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) return set_err(LPAMD_E_HIP, "launch of %s failed: %s", what, hipGetErrorString(err));
  return LPAMD_OK;
}

// --- enqueue helpers (no synchronisation) ---------------------------------
int enqueue_price_partial(lpamd_engine* e) {
  e->price_partials_fresh = true;
  hipLaunchKernelGGL(k_price_partial, dim3(e->n_price_part), dim3(kPartialThreads), 0, e->stream, e->obj, e->is_basic, e->sh, e->st,
                     e->part_v, e->part_i);
  return check_launch("k_price_partial");
}

int enqueue_pricing(lpamd_engine* e, int strategy, Candidate* out) {
  if (strategy == LPAMD_MAX_COST) {
    if (!e->price_partials_fresh) LP_TRY(enqueue_price_partial(e));
    hipLaunchKernelGGL(k_price_fold, dim3(1), dim3(kPartialThreads), 0, e->stream, e->part_v, e->part_i, e->n_price_part, e->sh, e->st, out);
    return check_launch("k_price_fold");
  }
  if (strategy == LPAMD_MAX_REDUCTION) {
    hipLaunchKernelGGL(k_pr_columns, dim3(e->sh.nloc), dim3(256), 0, e->stream, e->T, e->obj, e->is_basic, e->sh, e->st,
                       e->cmin, e->crow, e->pr_stamp);
    LP_TRY(check_launch("k_pr_columns"));
    hipLaunchKernelGGL(k_pr_partial, dim3(e->n_price_part), dim3(kPartialThreads), 0, e->stream, e->obj, e->cmin, e->crow, e->sh, e->st,
                       e->part_v, e->part_i, e->part_i + 2 * kMaxPartials);
    LP_TRY(check_launch("k_pr_partial"));
    e->price_partials_fresh = false;   // the partial buffers now hold max-reduction scores
    hipLaunchKernelGGL(k_pr_fold, dim3(1), dim3(kPartialThreads), 0, e->stream, e->obj, e->crow, e->part_v, e->part_i,
                       e->part_i + 2 * kMaxPartials, e->n_price_part, e->sh, e->st, out);
    return check_launch("k_pr_fold");
  }
  // reference: neither branch of lp_simplex_tableau.cc:188/235 fires and the live
  // assert at :14 aborts; here the call is rejected instead.
  return set_err(LPAMD_E_UNSUPPORTED, "pivoting strategy %d is not implemented for the tableau path", strategy);
}

template <int kUnroll>
void launch_update(lpamd_engine* e, int grid, bool nt, bool fuse_pr) {
#define LPAMD_LAUNCH_UPDATE(NT, PC, PR)                                                                                         \
  hipLaunchKernelGGL((k_update<kUnroll, NT, PC, PR>), dim3(grid), dim3(kUpdateThreads), 0, e->stream, e->T, e->sh.lda, e->colp, \
                     e->rowp, e->active, e->st, e->sh, e->is_basic, e->basis, e->rhs_next, e->cand_next, e->cmin, e->crow,      \
                     e->pr_stamp, e->update_slot)
  if (fuse_pr) {   // max-reduction pricing of the next iteration rides along
    if (nt && e->update_prec_fused) LPAMD_LAUNCH_UPDATE(true, true, true);
    else if (nt) LPAMD_LAUNCH_UPDATE(true, false, true);
    else if (e->update_prec_fused) LPAMD_LAUNCH_UPDATE(false, true, true);
    else LPAMD_LAUNCH_UPDATE(false, false, true);
  } else if (nt && e->update_prec) LPAMD_LAUNCH_UPDATE(true, true, false);
  else if (nt) LPAMD_LAUNCH_UPDATE(true, false, false);
  else if (e->update_prec) LPAMD_LAUNCH_UPDATE(false, true, false);
  else LPAMD_LAUNCH_UPDATE(false, false, false);
#undef LPAMD_LAUNCH_UPDATE
}

int enqueue_update(lpamd_engine* e, bool fuse_pr) {
  e->tab_version++;
  e->loop_depth = 1;   // a pass with one pivot: the step records of an earlier chain do not describe it
  if (fuse_pr && e->update_slot >= 0) return set_err(LPAMD_E_STATE, "the two-launch loop has no fused max-reduction pricing");
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (e->profiling) {
    if (e->ev_used + 2 > e->ev.size()) {
      for (int k = 0; k < 2; k++) {
        hipEvent_t ev;
        HIP_TRY(hipEventCreate(&ev));
        e->ev.push_back(ev);
      }
    }
    e0 = e->ev[e->ev_used++];
    e1 = e->ev[e->ev_used++];
    HIP_TRY(hipEventRecord(e0, e->stream));
  }
  // accesses in flight per lane: as many as one column allows, at most 16
  const int n_slabs = std::min(e->n_slabs, e->update_unroll_cap);   // 512-row slabs = 256 vectors of 16 B
  const int grid = e->update_grid;
  if (n_slabs >= 32) launch_update<32>(e, grid, e->update_nt, fuse_pr);
  else if (n_slabs >= 16) launch_update<16>(e, grid, e->update_nt, fuse_pr);
  else if (n_slabs >= 8) launch_update<8>(e, grid, e->update_nt, fuse_pr);
  else if (n_slabs >= 4) launch_update<4>(e, grid, e->update_nt, fuse_pr);
  else if (n_slabs >= 2) launch_update<2>(e, grid, e->update_nt, fuse_pr);
  else launch_update<1>(e, grid, e->update_nt, fuse_pr);
  LP_TRY(check_launch("k_update"));
  if (e->profiling) HIP_TRY(hipEventRecord(e1, e->stream));
  return LPAMD_OK;
}

int enqueue_ratio_partial(lpamd_engine* e, bool fold_price, Candidate* cand, const double* xcol, const unsigned long long* first_basic = nullptr);
int enqueue_ratio_partial(lpamd_engine* e, bool fold_price, Candidate* cand, const double* xcol, const unsigned long long* first_basic) {
  double* rv = e->part_v + kMaxPartials;
  int64_t* ri = e->part_i + kMaxPartials;
  if (fold_price)
    hipLaunchKernelGGL(k_ratio_partial<true>, dim3(e->n_ratio_part), dim3(kPartialThreads), 0, e->stream, e->T, e->sh, e->st, e->part_v,
                       e->part_i, e->n_price_part, cand, xcol, rv, ri, first_basic);
  else
    hipLaunchKernelGGL(k_ratio_partial<false>, dim3(e->n_ratio_part), dim3(kPartialThreads), 0, e->stream, e->T, e->sh, e->st, e->part_v,
                       e->part_i, e->n_price_part, cand, xcol, rv, ri, first_basic);
  return check_launch("k_ratio_partial");
}

// pivot set-up + row gather (one launch)
int enqueue_finalize(lpamd_engine* e, bool find_row, const Candidate* cand, const double* xcol, int64_t d_override) {
  const double* rv = e->part_v + kMaxPartials;
  const int64_t* ri = e->part_i + kMaxPartials;
  const int grid = e->n_row_blocks + e->n_price_part;   // column workgroups double as the next pricing partials
  if (find_row)
    hipLaunchKernelGGL(k_pivot_finalize<true>, dim3(grid), dim3(kPartialThreads), 0, e->stream, e->T, e->sh, cand, xcol, rv, ri,
                       e->n_ratio_part, e->n_row_blocks, d_override, e->st, e->colp, e->obj, e->is_basic, e->basis, e->rowp, e->active,
                       e->trace, e->trace_cap, e->part_v, e->part_i, e->rhs_next, e->cand_next, e->pass);
  else
    hipLaunchKernelGGL(k_pivot_finalize<false>, dim3(grid), dim3(kPartialThreads), 0, e->stream, e->T, e->sh, cand, xcol, rv, ri,
                       e->n_ratio_part, e->n_row_blocks, d_override, e->st, e->colp, e->obj, e->is_basic, e->basis, e->rowp, e->active,
                       e->trace, e->trace_cap, e->part_v, e->part_i, e->rhs_next, e->cand_next, e->pass);
  LP_TRY(check_launch("k_pivot_finalize"));
  e->price_partials_fresh = true;
  return LPAMD_OK;
}

// pivot set-up, then the rank-1 update
int enqueue_finalize_update(lpamd_engine* e, bool find_row, const Candidate* cand, const double* xcol, int64_t d_override,
                            bool fuse_pr = false) {
  LP_TRY(enqueue_finalize(e, find_row, cand, xcol, d_override));
  return enqueue_update(e, fuse_pr && e->fuse_pr_enabled);
}

template <int kUnroll>
void launch_update_chain(lpamd_engine* e, int grid, bool nt) {
  if (nt)
    hipLaunchKernelGGL((k_update_chain<kUnroll, true>), dim3(grid), dim3(kUpdateThreads), 0, e->stream, e->T, e->sh.lda, e->colp, e->pass,
                       e->list_full, e->list_part, e->rsl_full, e->rsl_part, e->chain_depth, e->st, e->steps, e->sh, e->is_basic, e->basis);
  else
    hipLaunchKernelGGL((k_update_chain<kUnroll, false>), dim3(grid), dim3(kUpdateThreads), 0, e->stream, e->T, e->sh.lda, e->colp, e->pass,
                       e->list_full, e->list_part, e->rsl_full, e->rsl_part, e->chain_depth, e->st, e->steps, e->sh, e->is_basic, e->basis);
}

// col' in registers: slab shape (kR, kC) chosen by e->slab_shape, chain length kK >= chain_depth; false: not applicable
template <int kK, int kR, int kC>
void launch_slab(lpamd_engine* e) {
  const int n_slabs = (int)(e->sh.lda / (2 * kUpdateThreads * kR));
  const int n_grp = std::max(1, std::min((e->ncl + kC - 1) / kC, std::max(1, e->slab_grid / n_slabs)));
  const int grid = n_slabs * n_grp;
  if (e->update_nt)
    hipLaunchKernelGGL((k_update_slab<kK, kR, kC, true>), dim3(grid), dim3(kUpdateThreads), 0, e->stream, e->T, e->sh.lda, e->colp, e->pass,
                       e->list_full, e->list_part, e->rsl_full, e->rsl_part, e->chain_depth, e->st, e->steps, e->sh, e->is_basic, e->basis,
                       n_slabs);
  else
    hipLaunchKernelGGL((k_update_slab<kK, kR, kC, false>), dim3(grid), dim3(kUpdateThreads), 0, e->stream, e->T, e->sh.lda, e->colp, e->pass,
                       e->list_full, e->list_part, e->rsl_full, e->rsl_part, e->chain_depth, e->st, e->steps, e->sh, e->is_basic, e->basis,
                       n_slabs);
}
template <int kK>
bool launch_slab_shape(lpamd_engine* e) {
  switch (e->slab_shape) {
    case 1: launch_slab<kK, 1, 8>(e); return true;
    case 2: launch_slab<kK, 1, 4>(e); return true;
    case 3: if (e->sh.lda % 1024) return false; launch_slab<kK, 2, 4>(e); return true;
    case 4: if (e->sh.lda % 1024) return false; launch_slab<kK, 2, 8>(e); return true;
    case 5: if (e->sh.lda % 1024) return false; launch_slab<kK, 2, 2>(e); return true;
    default: return false;
  }
}
// col' in LDS, steps in the outer loop (chains longer than 16): 128 rows per workgroup, batches of deep_batch columns
template <int kC>
void launch_deep(lpamd_engine* e) {
  const int n_slabs = (int)(e->sh.lda / (2 * kWave));
  // Measured (full shard, 2-, 4- and 8-GPU shards of m=32768 n=65536): a wavefront should walk about 48 batches; fewer and
  // the start of a stream (col' slices into LDS, a first batch nobody prefetched) shows, more and the chip drains unevenly.
  constexpr int kWavesPerGroup = kDeepThreads / kWave, kBatchesPerWave = 48;
  const int n_auto = std::max(1, (e->ncl / kC + kWavesPerGroup * kBatchesPerWave / 2) / (kWavesPerGroup * kBatchesPerWave));
  const int n_grp = e->deep_grid > 0 ? std::max(1, e->deep_grid / n_slabs) : n_auto;
  const size_t lds = sizeof(double2) * kWave * (size_t)e->chain_depth;
  if (e->update_nt)
    hipLaunchKernelGGL((k_update_deep<kC, true>), dim3(n_slabs * n_grp), dim3(kDeepThreads), lds, e->stream, e->T, e->sh.lda, e->colp, e->pass,
                       e->list_full, e->list_part, e->rsl_full, e->rsl_part, e->chain_depth, e->st, e->steps, e->sh, e->is_basic, e->basis,
                       n_slabs, e->ptr_full);
  else
    hipLaunchKernelGGL((k_update_deep<kC, false>), dim3(n_slabs * n_grp), dim3(kDeepThreads), lds, e->stream, e->T, e->sh.lda, e->colp, e->pass,
                       e->list_full, e->list_part, e->rsl_full, e->rsl_part, e->chain_depth, e->st, e->steps, e->sh, e->is_basic, e->basis,
                       n_slabs, e->ptr_full);
}
bool launch_update_deep(lpamd_engine* e) {
  switch (e->deep_batch) {
    case 4: launch_deep<4>(e); return true;
    case 8: launch_deep<8>(e); return true;
    default: return false;
  }
}

bool launch_update_slab(lpamd_engine* e) {
  // the instantiated chain lengths; a chain shorter than the template takes the step-by-step path of the kernel
  if (e->chain_depth <= 4) return launch_slab_shape<4>(e);
  if (e->chain_depth <= 8) return launch_slab_shape<8>(e);
  if (e->chain_depth <= 12) return launch_slab_shape<12>(e);
  if (e->chain_depth <= 16) return launch_slab_shape<16>(e);
  if (e->chain_depth <= 20) return launch_slab_shape<20>(e);
  if (e->chain_depth <= 24) return launch_slab_shape<24>(e);   // (32 steps in registers measured slower than 24 at m=4096 n=8192)
  return false;   // longer chains do not fit the registers: k_update_deep, or k_update_chain with col' from L2
}

// Step j >= 1 of a chain: ratio partials (pricing fold + rebuilt column on one GPU; candidate record + column
// from the exchange over column shards), then the set-up.
int enqueue_chain_step(lpamd_engine* e, int j, const Candidate* xcand, double* xcol) {
  double* rv = e->part_v + kMaxPartials;
  int64_t* ri = e->part_i + kMaxPartials;
  double* colv = xcol ? xcol : e->colv;
  if (xcol)
    hipLaunchKernelGGL(k_chain_ratio<true>, dim3(e->n_ratio_part), dim3(kPartialThreads), 0, e->stream, e->T, e->sh, e->st, e->steps, j,
                       e->chain_depth, e->part_v, e->part_i, e->n_price_part, e->colp, e->rowp, e->rstride, e->rhs_next, colv, rv, ri, xcand);
  else
    hipLaunchKernelGGL(k_chain_ratio<false>, dim3(e->n_ratio_part), dim3(kPartialThreads), 0, e->stream, e->T, e->sh, e->st, e->steps, j,
                       e->chain_depth, e->part_v, e->part_i, e->n_price_part, e->colp, e->rowp, e->rstride, e->rhs_next, colv, rv, ri, xcand);
  LP_TRY(check_launch("k_chain_ratio"));
  hipLaunchKernelGGL((k_chain_finalize<true, false>), dim3(e->n_row_blocks + e->n_price_part), dim3(kPartialThreads), 0, e->stream, e->T, e->sh,
                     e->st, e->steps, j, rv, ri, e->n_ratio_part, e->n_row_blocks, e->colp, e->rowp, e->rstride, e->rhs_next, colv, xcand,
                     e->obj, e->is_basic, e->basis, e->trace, e->trace_cap, e->part_v, e->part_i);
  LP_TRY(check_launch("k_chain_finalize"));
  e->price_partials_fresh = true;
  return LPAMD_OK;
}

int enqueue_chain_update(lpamd_engine* e);

// The first two launches of a MAX_REDUCTION pricing from bounds: the rows of the bound, then the bound of every candidate
// column (with_top: and what k_prb_tail selects from).
int enqueue_prb_bounds(lpamd_engine* e, int j, bool with_top, int* n_bound_blocks_out) {
  // the slots of the pass before serve the first pricing of this one if nothing but that pass has touched the tableau since
  const int carry = (j == 0 && e->prb_carry && e->sh.world == 1 && e->prb_use_cache && e->prb_cache_version == e->tab_version) ? 1 : 0;
  double* pv = e->prb_v + kMaxPartials;
  int64_t* pi = e->prb_i + kMaxPartials;
  int64_t* pf = e->prb_i + 2 * kMaxPartials;
#define LPAMD_LAUNCH_ROWS(KEEP)                                                                                                      \
  hipLaunchKernelGGL(k_prb_rows<KEEP>, dim3(1), dim3(kPrbRowThreads), 0, e->stream, e->T, e->sh, e->st, e->steps, j, e->chain_depth, \
                     e->rhs_next, e->colp, e->prb_rows, e->prb_renew, e->trace, e->trace_cap, e->prb_info, e->prb_c, e->pass, carry)
  const int64_t seg_threads = kPrbRowThreads / kPrbSegments;
  const int64_t per_thread = (e->sh.m + (int64_t)kPrbSegments * seg_threads - 1) / ((int64_t)kPrbSegments * seg_threads);   // segments of ceil(m / 128) rows, eight threads each
  if (e->sh.m % kPrbSegments != 0 || e->sh.m < kPrbSegments) {   // uneven segments: the general kernel sizes
    if (e->sh.m <= 256 * kPrbSegments) LPAMD_LAUNCH_ROWS(32); else LPAMD_LAUNCH_ROWS(0);
  }
  else if (per_thread <= 4) LPAMD_LAUNCH_ROWS(4);
  else if (per_thread <= 8) LPAMD_LAUNCH_ROWS(8);
  else if (per_thread <= 16) LPAMD_LAUNCH_ROWS(16);
  else if (per_thread <= 32) LPAMD_LAUNCH_ROWS(32);
  else LPAMD_LAUNCH_ROWS(0);
#undef LPAMD_LAUNCH_ROWS
  LP_TRY(check_launch("k_prb_rows"));
  const bool cached = e->prb_use_cache;
  int n_bound_blocks = (int)std::min<int64_t>(kMaxPartials, std::max<int64_t>(1, (e->sh.nloc + kWave - 1) / kWave));   // tiles of 64 columns
  if (e->prb_bound_blocks > 0) n_bound_blocks = std::min(n_bound_blocks, e->prb_bound_blocks);
  hipLaunchKernelGGL(k_prb_bounds, dim3(n_bound_blocks), dim3(kPartialThreads), 0, e->stream, e->T, e->sh, e->st, e->steps, j, e->obj, e->is_basic,
                     e->rowp, e->rstride, e->prb_info, e->prb_c, e->prb_renew, cached ? e->prb_cache : nullptr, e->prb_ub, e->prb_ucand, pv, pi, pf,
                     with_top ? e->prb_top_u : nullptr, e->prb_top_l, with_top ? e->prb_top_u + kMaxPartials : nullptr);
  *n_bound_blocks_out = n_bound_blocks;
  return check_launch("k_prb_bounds");
}

// MAX_REDUCTION pricing of step j of a pass from bounds (six small launches, no pass over the tableau): the
// candidate record (column, row, score) lands in `cand`.
int enqueue_prb_pricing(lpamd_engine* e, int j, Candidate* cand, bool record_step = true) {
  double* pv = e->prb_v + kMaxPartials;
  int64_t* pi = e->prb_i + kMaxPartials;
  int64_t* pf = e->prb_i + 2 * kMaxPartials;
  double* rv = e->part_v + kMaxPartials;
  int64_t* ri = e->part_i + kMaxPartials;
  int n_bound_blocks = 0;
  LP_TRY(enqueue_prb_bounds(e, j, false, &n_bound_blocks));
  hipLaunchKernelGGL(k_prb_first, dim3(e->n_ratio_part), dim3(kPartialThreads), 0, e->stream, e->T, e->sh, e->st, e->steps, j, e->colp, e->rowp,
                     e->rstride, e->rhs_next, pv, pi, pf, n_bound_blocks, e->prb_info, rv, ri);
  LP_TRY(check_launch("k_prb_first"));
  hipLaunchKernelGGL(k_prb_list, dim3(e->n_price_part), dim3(kPartialThreads), 0, e->stream, e->sh, e->st, e->steps, j, e->obj, rv, ri,
                     e->n_ratio_part, e->prb_ub, e->prb_ucand, e->prb_info, e->prb_elist, e->prb_escore, e->prb_erow);
  LP_TRY(check_launch("k_prb_list"));
  const int n_slices = std::min(e->n_ratio_part, kPrbSlices);
  hipLaunchKernelGGL(k_prb_eval, dim3(2048), dim3(kPartialThreads), 0, e->stream, e->T, e->sh, e->st, e->steps, j, e->obj,
                     e->colp, e->rowp, e->rstride, e->rhs_next, e->prb_info, e->prb_elist, e->prb_escore, e->prb_erow, n_slices, e->prb_ep_v,
                     e->prb_ep_i);
  LP_TRY(check_launch("k_prb_eval"));
  hipLaunchKernelGGL(k_prb_pick, dim3(1), dim3(kPartialThreads), 0, e->stream, e->sh, e->st, e->steps, j, e->obj, e->prb_info, e->prb_elist,
                     e->prb_escore, e->prb_erow, n_slices, e->prb_ep_v, e->prb_ep_i, cand, record_step ? 1 : 0);
  return check_launch("k_prb_pick");
}

// Step j >= 1 of a pass on one GPU: rows, bounds, then exact evaluation, fold and set-up in one launch (k_prb_tail).
// slot_cand / slot_col (column shards): the launch ends with this rank's candidate packed for the exchange instead of the set-up.
int enqueue_prb_step_tail(lpamd_engine* e, int j, Candidate* slot_cand = nullptr, double* slot_col = nullptr) {
  int n_bound_blocks = 0;
  LP_TRY(enqueue_prb_bounds(e, j, true, &n_bound_blocks));
  const size_t words = (size_t)kFusedReplicas * (fused_region_words(e->tail_rows + e->tail_cols, kTailEvalWords) + fused_region_words(e->tail_rows + e->tail_cols, kTailBackWords));
  e->tail_epoch = (e->tail_epoch + 1) & 0xFFFFFFu;
  if (e->tail_epoch == 0) {   // the launch number wrapped: no record of an old launch may carry a tag that comes round again
    HIP_TRY(hipMemsetAsync(e->rec_tail, 0, sizeof(uint64_t) * words, e->stream));
    e->tail_epoch = 1;
  }
  const int grid = e->tail_rows + e->tail_cols;
#define LPAMD_LAUNCH_TAIL_AS(REC, SHARD)                                                                                                 \
  hipLaunchKernelGGL((k_prb_tail<REC, SHARD>), dim3(grid), dim3(kPartialThreads), 0, e->stream, e->T, e->sh, e->st, e->steps, j, e->tail_rows, e->colp, \
                     e->rowp, e->rstride, e->rhs_next, e->obj, e->is_basic, e->basis, e->trace, e->trace_cap, e->prb_top_u, e->prb_top_l,   \
                     e->prb_top_u + kMaxPartials, n_bound_blocks, e->prb_ub, e->prb_ucand, e->prb_info, e->rec_tail,                     \
                     e->rec_tail + (size_t)kFusedReplicas * fused_region_words(grid, kTailEvalWords), e->tail_epoch, slot_cand, slot_col)
#define LPAMD_LAUNCH_TAIL(REC) do { if (slot_cand) LPAMD_LAUNCH_TAIL_AS(REC, true); else LPAMD_LAUNCH_TAIL_AS(REC, false); } while (0)
  if (n_bound_blocks <= kPartialThreads) LPAMD_LAUNCH_TAIL(1);
  else if (n_bound_blocks <= 2 * kPartialThreads) LPAMD_LAUNCH_TAIL(2);
  else if (n_bound_blocks <= 4 * kPartialThreads) LPAMD_LAUNCH_TAIL(4);
  else LPAMD_LAUNCH_TAIL(8);
#undef LPAMD_LAUNCH_TAIL_AS
#undef LPAMD_LAUNCH_TAIL
  return check_launch("k_prb_tail");
}

// One pass of MAX_REDUCTION pivots: every step is priced from bounds and set up on entries rebuilt through the
// steps before it; the pass kernel then applies the chain.
int enqueue_prb_pass(lpamd_engine* e) {
  LP_TRY(enqueue_prb_pricing(e, 0, e->cand));
  LP_TRY(enqueue_finalize(e, false, e->cand, nullptr, -1));
  for (int j = 1; j < e->chain_depth; j++) {
    if (e->prb_tail) { LP_TRY(enqueue_prb_step_tail(e, j)); continue; }
    LP_TRY(enqueue_prb_pricing(e, j, e->cand));
    hipLaunchKernelGGL((k_chain_finalize<false, true>), dim3(e->n_row_blocks + e->n_price_part), dim3(kPartialThreads), 0, e->stream, e->T,
                       e->sh, e->st, e->steps, j, nullptr, nullptr, 0, e->n_row_blocks, e->colp, e->rowp, e->rstride, e->rhs_next, nullptr,
                       e->cand, e->obj, e->is_basic, e->basis, e->trace, e->trace_cap, e->part_v, e->part_i);
    LP_TRY(check_launch("k_chain_finalize"));
  }
  e->price_partials_fresh = false;
  LP_TRY(enqueue_chain_update(e));
  e->prb_cache_version = e->tab_version;
  return LPAMD_OK;
}

// Column shards, MAX_REDUCTION from bounds.  Step j = the pivots already set up in this pass.
// 1. this rank's best candidate (exact score, row) and its column, packed into `slot` for the exchange
int prb_shard_price(lpamd_engine* e, uint64_t* slot, int header_words) {
  const int j = e->update_pending ? e->pending_steps : 0;
  Candidate* rec = reinterpret_cast<Candidate*>(slot);
  // steps 1.. of a pass: everything behind the bounds, the record and the column in one launch (k_prb_tail<.., shard>)
  if (e->prb_tail && j >= 1) return enqueue_prb_step_tail(e, j, rec, reinterpret_cast<double*>(slot + header_words));
  LP_TRY(enqueue_prb_pricing(e, j, rec, false));
  hipLaunchKernelGGL(k_prb_export, dim3((unsigned)((e->sh.lda + kPartialThreads - 1) / kPartialThreads)), dim3(kPartialThreads), 0, e->stream, e->T,
                     e->sh, e->st, e->steps, j, e->colp, e->rowp, e->rstride, rec, reinterpret_cast<double*>(slot + header_words));
  return check_launch("k_prb_export");
}
// 2. set-up of that step for the winner among all ranks: record in `cand` (the engine's own for step 0, where the
// fold has settled the status as well), its column in `xbuf`
int prb_shard_setup(lpamd_engine* e, const Candidate* cand, const double* xbuf) {
  const int j = e->update_pending ? e->pending_steps : 0;
  if (j >= e->chain_depth) return set_err(LPAMD_E_STATE, "the pass already holds %d pivots", j);
  e->apply_since_pull++;
  LP_TRY(ensure_trace(e, e->h_st.iter + e->apply_since_pull + 2));
  if (j == 0) {
    LP_TRY(enqueue_finalize(e, false, cand, xbuf, -1));
    e->update_pending = true;
    e->pending_steps = 1;
    return LPAMD_OK;
  }
  hipLaunchKernelGGL((k_chain_finalize<false, false>), dim3(e->n_row_blocks + e->n_price_part), dim3(kPartialThreads), 0, e->stream, e->T, e->sh,
                     e->st, e->steps, j, nullptr, nullptr, 0, e->n_row_blocks, e->colp, e->rowp, e->rstride, e->rhs_next, xbuf, cand, e->obj,
                     e->is_basic, e->basis, e->trace, e->trace_cap, e->part_v, e->part_i);
  LP_TRY(check_launch("k_chain_finalize"));
  e->pending_steps++;
  return LPAMD_OK;
}

// The further pivots of a pass and the pass itself, for the pivot that k_pivot_finalize has just set up.
int enqueue_chain_tail(lpamd_engine* e, int depth) {
  if (e->fused && depth > 1) {
    e->fused_epoch = (e->fused_epoch + 1) & 0xFFFFFFu;
    if (e->fused_epoch == 0) {   // the launch number wrapped: no record of an old launch may carry a tag that comes round again
      HIP_TRY(hipMemsetAsync(e->rec_ratio, 0, sizeof(uint64_t) * 2 * kFusedReplicas * fused_region_words(e->fused_rows, kFusedRatioWords), e->stream));
      HIP_TRY(hipMemsetAsync(e->rec_price, 0, sizeof(uint64_t) * 2 * kFusedReplicas * fused_region_words(e->fused_cols, kFusedPriceWords), e->stream));
      e->fused_epoch = 1;
    }
#define LPAMD_LAUNCH_FUSED(THREADS, REG, HIST)                                                                                        \
  hipLaunchKernelGGL((k_chain_fused<THREADS, REG, HIST>), dim3(e->fused_rows + e->fused_cols), dim3(THREADS), 0, e->stream, e->T, e->sh, e->st, \
                     e->steps, depth, e->fused_rows, e->part_v, e->part_i, e->n_price_part, e->colp, e->rowp, e->rstride, e->rhs_next,     \
                     e->colv, e->obj, e->is_basic, e->basis, e->trace, e->trace_cap, e->rec_ratio, e->rec_price, e->fused_epoch)
    const bool h24 = e->chain_depth <= 24;
    if (e->fused_threads == 1024) LPAMD_LAUNCH_FUSED(1024, false, 32);
    else if (e->fused_threads == 512) { if (!e->fused_reg) LPAMD_LAUNCH_FUSED(512, false, 32); else if (h24) LPAMD_LAUNCH_FUSED(512, true, 24); else LPAMD_LAUNCH_FUSED(512, true, 32); }
    else { if (!e->fused_reg) LPAMD_LAUNCH_FUSED(256, false, 32); else if (h24) LPAMD_LAUNCH_FUSED(256, true, 24); else LPAMD_LAUNCH_FUSED(256, true, 32); }
#undef LPAMD_LAUNCH_FUSED
    LP_TRY(check_launch("k_chain_fused"));
    e->price_partials_fresh = true;
    e->fused_pending = true;
    return enqueue_chain_update(e);
  }
  for (int j = 1; j < depth; j++) LP_TRY(enqueue_chain_step(e, j, nullptr, nullptr));
  return enqueue_chain_update(e);
}

int enqueue_chain_update(lpamd_engine* e) {
  e->tab_version++;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (e->profiling) {
    if (e->ev_used + 2 > e->ev.size()) {
      for (int k = 0; k < 2; k++) {
        hipEvent_t ev;
        HIP_TRY(hipEventCreate(&ev));
        e->ev.push_back(ev);
      }
    }
    e0 = e->ev[e->ev_used++];
    e1 = e->ev[e->ev_used++];
    HIP_TRY(hipEventRecord(e0, e->stream));
  }
  // which pass kernel: the one with col' in LDS for chains beyond the 16 steps the registers hold (the depth differs by strategy)
  const bool deep = e->deep_batch > 0 && (e->chain_depth > 16 || e->deep_any_depth);
#define LPAMD_LAUNCH_LIST(THREADS)                                                                                                       \
  hipLaunchKernelGGL(k_chain_list<THREADS>, dim3((unsigned)((e->ncl + THREADS - 1) / THREADS)), dim3(THREADS), 0, e->stream, e->rowp, e->rstride, \
                     e->chain_depth, e->steps, e->st, e->ncl, e->pass, e->list_full, e->list_part, e->rsl_full, e->rsl_part, e->chain_depth, \
                     deep ? e->deep_batch : 0, e->T, e->ptr_full, e->fused_pending ? 1 : 0)
  if (e->ncl <= 32768) LPAMD_LAUNCH_LIST(256); else LPAMD_LAUNCH_LIST(1024);
#undef LPAMD_LAUNCH_LIST
  e->fused_pending = false;
  LP_TRY(check_launch("k_chain_list"));
  const int n_slabs = std::min(e->n_slabs, e->update_unroll_cap);
  const int grid = e->update_grid;
  if (deep && launch_update_deep(e)) { /* col' in LDS */ }
  else if (launch_update_slab(e)) { /* col' in registers */ }
  else if (n_slabs >= 16) launch_update_chain<16>(e, grid, e->update_nt);
  else if (n_slabs >= 8) launch_update_chain<8>(e, grid, e->update_nt);
  else if (n_slabs >= 4) launch_update_chain<4>(e, grid, e->update_nt);
  else if (n_slabs >= 2) launch_update_chain<2>(e, grid, e->update_nt);
  else launch_update_chain<1>(e, grid, e->update_nt);
  LP_TRY(check_launch("k_update_chain"));
  if (e->profiling) HIP_TRY(hipEventRecord(e1, e->stream));
  return LPAMD_OK;
}

// ratio test (if the row is not known yet) + set-up + update for the candidate in `cand`
int enqueue_pivot(lpamd_engine* e, bool find_row, Candidate* cand, const double* xcol, int64_t d_override, bool fuse_pr = false) {
  if (find_row) LP_TRY(enqueue_ratio_partial(e, false, cand, xcol));
  return enqueue_finalize_update(e, find_row, cand, xcol, d_override, fuse_pr);
}

// One iteration of the main loop on one GPU: 4 launches for MAX_COST (pricing
// partials, ratio partials, set-up + row gather, update), 4 for MAX_REDUCTION.
int enqueue_iteration(lpamd_engine* e, int strategy) {
  if (strategy == LPAMD_MAX_COST && e->two_launch) {
    // pricing fold + ratio test + set-up in one kernel (partials read from one half of the buffers, the
    // next ones written to the other), then the update with this pivot's counter slot
    if (!e->price_partials_fresh) { LP_TRY(enqueue_price_partial(e)); e->price_buf = 0; }
    const int cur = e->price_buf, nxt = 1 - cur, slot = (int)(e->slot_counter++ & 1);
    const int grid = e->n_row_blocks + e->n_price_part;
    hipLaunchKernelGGL(k_pivot_setup_small, dim3(grid), dim3(kPartialThreads), 0, e->stream, e->T, e->sh, e->part_v + cur * kMaxPartials,
                       e->part_i + cur * kMaxPartials, e->n_price_part, e->n_row_blocks, e->st, slot, e->colp, e->obj, e->is_basic, e->basis,
                       e->rowp, e->active, e->trace, e->trace_cap, e->part_v + nxt * kMaxPartials, e->part_i + nxt * kMaxPartials, e->rhs_next,
                       e->cand_next, e->pass);
    LP_TRY(check_launch("k_pivot_setup_small"));
    e->price_buf = nxt;
    e->update_slot = slot;
    const int rc = enqueue_update(e, false);
    e->update_slot = -1;
    return rc;
  }
  if (strategy == LPAMD_MAX_COST && e->chain_depth > 1) {
    // up to chain_depth pivots, one pass over the tableau (2 * chain_depth + 1 launches)
    if (!e->price_partials_fresh) LP_TRY(enqueue_price_partial(e));
    LP_TRY(enqueue_ratio_partial(e, true, e->cand, nullptr));
    LP_TRY(enqueue_finalize(e, true, e->cand, nullptr, -1));
    return enqueue_chain_tail(e, e->chain_depth);
  }
  if (strategy == LPAMD_MAX_COST) {
    // the previous pivot's set-up kernel already left the pricing partials of this iteration
    if (!e->price_partials_fresh) LP_TRY(enqueue_price_partial(e));
    LP_TRY(enqueue_ratio_partial(e, true, e->cand, nullptr));
    return enqueue_finalize_update(e, true, e->cand, nullptr, -1);
  }
  if (strategy == LPAMD_MAX_REDUCTION && e->pr_bounds) {
    const int32_t keep = e->chain_depth;     // everything a pass enqueues reads the depth from here
    e->chain_depth = e->mr_depth;
    const int rc = enqueue_prb_pass(e);
    e->chain_depth = keep;
    return rc;
  }
  LP_TRY(enqueue_pricing(e, strategy, e->cand));
  return enqueue_finalize_update(e, false, e->cand, nullptr, -1, strategy == LPAMD_MAX_REDUCTION);
}

int read_local(const lpamd_engine* e, const void* src, void* dst, size_t bytes);

// The main loop for a tableau whose basic columns are no longer unit columns (PivotState::dirty; one GPU): one pivot per
// pass, the leaving variable and the rows that take part found the reference's way on every iteration.  MAX_COST
// (reference lp_simplex_tableau.cc:188-227): a row takes part in the ratio test only if some basic column has a non-zero
// entry in it, and the first such column leaves.  MAX_REDUCTION (reference lp_simplex_tableau.cc:235-374): the basic
// variable of a row comes from the discovery loop over the basic columns; a row without one makes the result UNBOUNDED.
// The host reads the choice back once per pivot: this loop serves inputs on which the reference's arithmetic has already
// lost its footing (entries of 4e9 and more), and what counts here is that the same pivots are taken.
int run_loop_scanning(lpamd_engine* e, int strategy, int64_t budget, int32_t* result) {
  if (e->sh.world != 1) return set_err(LPAMD_E_UNSUPPORTED, "basic columns are no longer unit columns (entries of 4e9 and more); the loop that handles this needs the tableau on one GPU");
  const Shape& sh = e->sh;
  if (!e->first_basic) {
    HIP_TRY(hipMalloc(&e->first_basic, sizeof(unsigned long long) * sh.m));
    HIP_TRY(hipMalloc(&e->basis_disc, sizeof(long long) * sh.m));
  }
  LP_TRY(pull_state(e));
  const int64_t stop_at = budget < 0 ? std::numeric_limits<int64_t>::max() : e->h_st.iter + budget;
  e->h_st.status = ST_RUNNING;
  e->h_st.max_iter = std::numeric_limits<int64_t>::max();
  e->h_st.dirty = 2;   // known: the first kernels of an iteration go on
  e->h_st.dirty_step = 0;
  e->h_st.n_active = e->h_st.n_active2 = 0;
  LP_TRY(push_state(e));
  e->dirty_basis = true;
  for (;;) {
    if (e->h_st.iter >= stop_at) { *result = LPAMD_LIMIT; return LPAMD_OK; }
    LP_TRY(ensure_trace(e, e->h_st.iter + 2));
    Candidate c;
    long long d = -1;
    if (strategy == LPAMD_MAX_COST) {
      HIP_TRY(hipMemsetAsync(e->first_basic, 0xFF, sizeof(unsigned long long) * sh.m, e->stream));
      hipLaunchKernelGGL(k_basic_scan_rows, dim3(sh.nloc), dim3(256), 0, e->stream, e->T, sh, e->is_basic, e->first_basic);
      LP_TRY(check_launch("k_basic_scan_rows"));
      e->price_partials_fresh = false;
      LP_TRY(enqueue_pricing(e, LPAMD_MAX_COST, e->cand));
      LP_TRY(enqueue_ratio_partial(e, false, e->cand, nullptr, e->first_basic));
      LP_TRY(read_local(e, e->cand, &c, sizeof(c)));
      if (c.gcol < 0) { *result = LPAMD_SOLVED; return LPAMD_OK; }              // :201-205
      std::vector<double> rv(e->n_ratio_part);
      std::vector<int64_t> ri(e->n_ratio_part);
      LP_TRY(read_local(e, e->part_v + kMaxPartials, rv.data(), sizeof(double) * rv.size()));
      LP_TRY(read_local(e, e->part_i + kMaxPartials, ri.data(), sizeof(int64_t) * ri.size()));
      double bv = 0.0;
      int64_t bi = -1;
      for (size_t k = 0; k < rv.size(); k++)   // smaller ratio, then lower row
        if (ri[k] >= 0 && (bi < 0 || rv[k] < bv || (rv[k] == bv && ri[k] < bi))) { bv = rv[k]; bi = ri[k]; }
      if (bi < 0) { *result = LPAMD_UNBOUNDED; return LPAMD_OK; }               // :230-233
      unsigned long long fb = 0;
      LP_TRY(read_local(e, e->first_basic + bi, &fb, sizeof(fb)));
      c.row = bi;
      d = (long long)fb;
    } else {
      HIP_TRY(hipMemsetAsync(e->basis_disc, 0xFF, sizeof(long long) * sh.m, e->stream));   // -1 everywhere
      hipLaunchKernelGGL(k_basis_discovery, dim3(sh.nloc), dim3(256), 0, e->stream, e->T, sh, e->is_basic, e->basis_disc);
      LP_TRY(check_launch("k_basis_discovery"));
      LP_TRY(enqueue_pricing(e, LPAMD_MAX_REDUCTION, e->cand));
      LP_TRY(read_local(e, e->cand, &c, sizeof(c)));
      if (c.gcol < 0) { *result = LPAMD_SOLVED; return LPAMD_OK; }              // :365-369
      if (c.row < 0) { *result = LPAMD_UNBOUNDED; return LPAMD_OK; }            // :370-373
      LP_TRY(read_local(e, e->basis_disc + c.row, &d, sizeof(d)));
      if (d < 0) { *result = LPAMD_UNBOUNDED; return LPAMD_OK; }                // no basic variable was discovered for the row: d stays undefined, :370
    }
    HIP_TRY(hipMemcpyAsync(e->cand, &c, sizeof(c), hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));   // c is a stack object
    LP_TRY(enqueue_finalize_update(e, false, e->cand, nullptr, (int64_t)d, strategy == LPAMD_MAX_REDUCTION));
    LP_TRY(pull_state(e));
    if (e->h_st.status != ST_RUNNING) { *result = status_to_result(e->h_st.status); return LPAMD_OK; }
  }
}

// Device-resident main loop: at most `budget` pivots (budget < 0: none).
int run_loop(lpamd_engine* e, int strategy, int64_t budget, int32_t* result) {
  if (e->dirty_basis) return run_loop_scanning(e, strategy, budget, result);
  {
    int64_t before = 0;
    LP_TRY(pull_state(e));
    before = e->h_st.iter;
    LP_TRY(run_loop_device(e, strategy, budget, result));
    if (e->h_st.status != ST_DIRTY) return LPAMD_OK;
    // the device loop met a column whose entries no longer cancel to a unit column: the rest of the run scans like the reference
    const int64_t used = e->h_st.iter - before;
    return run_loop_scanning(e, strategy, budget < 0 ? budget : std::max<int64_t>(0, budget - used), result);
  }
}

int run_loop_device(lpamd_engine* e, int strategy, int64_t budget, int32_t* result) {
  LP_TRY(pull_state(e));
  if (budget == 0) { *result = LPAMD_LIMIT; return LPAMD_OK; }   // nothing to enqueue: the budget is spent before the first pivot
  e->h_st.status = ST_RUNNING;
  e->h_st.max_iter = budget < 0 ? std::numeric_limits<int64_t>::max() : e->h_st.iter + budget;
  e->h_st.n_active = e->h_st.n_active2 = 0;   // both counter slots of the two-launch loop start empty
  LP_TRY(push_state(e));
  // A run that ended on a terminal status may have left other data in the partial buffers (the set-up kernel is a
  // no-op then, max-reduction pricing shares the buffers): every run prices its first iteration afresh.
  e->price_partials_fresh = false;
  e->price_buf = 0;
  e->slot_counter = 0;
  struct RestorePriceBuf {   // every other path keeps the pricing partials in the first half
    lpamd_engine* e;
    ~RestorePriceBuf() {
      if (e->price_buf != 0) { e->price_buf = 0; e->price_partials_fresh = false; }
    }
  } restore{e};
  // With a budget the whole run is enqueued at once (kernels turn into no-ops
  // once the status leaves RUNNING); without one the batch grows geometrically.
  if (e->dense_first && !e->two_launch) {
    // the first pivot after an upload with entries below the drop threshold: one pivot per pass, every column (the
    // two-launch loop is one pivot per pass anyway)
    if (strategy == LPAMD_MAX_COST) {
      LP_TRY(enqueue_price_partial(e));
      LP_TRY(enqueue_ratio_partial(e, true, e->cand, nullptr));
      LP_TRY(enqueue_finalize_update(e, true, e->cand, nullptr, -1));
    } else {
      LP_TRY(enqueue_pricing(e, strategy, e->cand));
      LP_TRY(enqueue_finalize_update(e, false, e->cand, nullptr, -1, strategy == LPAMD_MAX_REDUCTION));
    }
    LP_TRY(pull_state(e));
    e->price_partials_fresh = e->price_partials_fresh && strategy == LPAMD_MAX_COST;
    if (e->h_st.status != ST_RUNNING) { *result = status_to_result(e->h_st.status); return LPAMD_OK; }
    if (budget >= 0 && e->h_st.iter >= e->h_st.max_iter) { *result = LPAMD_LIMIT; return LPAMD_OK; }
  }
  int64_t batch = budget >= 0 ? 1024 : 8;
  const int64_t depth = strategy == LPAMD_MAX_COST ? e->chain_depth : (strategy == LPAMD_MAX_REDUCTION && e->pr_bounds) ? e->mr_depth : 1;   // pivots an iteration may apply
  e->loop_depth = (int32_t)depth;
  for (;;) {
    int64_t todo = batch;
    if (budget >= 0) todo = std::min<int64_t>(batch, e->h_st.max_iter - e->h_st.iter);
    todo = (todo + depth - 1) / depth;
    LP_TRY(ensure_trace(e, e->h_st.iter + depth * todo));
    const int64_t before = e->h_st.iter;
    for (int64_t k = 0; k < todo; k++) LP_TRY(enqueue_iteration(e, strategy));
    LP_TRY(pull_state(e));
    if (e->h_st.status != ST_RUNNING) break;
    if (e->h_st.iter == before) {   // running but no pivot applied: never spin on the GPU
      e->h_st.status = ST_ERROR;
      set_err(LPAMD_E_STATE, "pivot loop made no progress");
      break;
    }
    if (budget >= 0 && e->h_st.iter >= e->h_st.max_iter) { e->h_st.status = ST_LIMIT; break; }
    batch = std::min<int64_t>(batch * 2, 1024);
  }
  *result = status_to_result(e->h_st.status);
  return LPAMD_OK;
}

int read_local(const lpamd_engine* e, const void* src, void* dst, size_t bytes) {
  HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return LPAMD_OK;
}

int gather_row_host(const lpamd_engine* e, int64_t r, std::vector<double>& row_local) {
  row_local.resize(e->ncl);
  hipLaunchKernelGGL(k_gather_row, dim3((e->ncl + 255) / 256), dim3(256), 0, e->stream, e->T, e->ncl, r, e->rowp);
  LP_TRY(check_launch("k_gather_row"));
  return read_local(e, e->rowp, row_local.data(), sizeof(double) * e->ncl);
}

int single_gpu_only(const lpamd_engine* e, const char* what) {
  if (e->sh.world != 1) return set_err(LPAMD_E_UNSUPPORTED, "%s needs the whole tableau on one GPU (world == 1)", what);
  return LPAMD_OK;
}

#define CHECK_ENGINE(e)                                                        \
  do {                                                                         \
    if (!(e)) return set_err(LPAMD_E_ARG, "null engine");                      \
    HIP_TRY(hipSetDevice((e)->device));                                        \
  } while (0)
#define CHECK_LOADED(e)                                                        \
  do {                                                                         \
    CHECK_ENGINE(e);                                                           \
    if (!(e)->loaded) return set_err(LPAMD_E_STATE, "no tableau uploaded yet"); \
  } while (0)
// entry points that change the tableau or start a pivot: not between lpamd_lookahead_setup and lpamd_lookahead_update
#define CHECK_NO_PENDING_UPDATE(e)                                                                                   \
  do {                                                                                                               \
    if ((e)->update_pending)                                                                                         \
      return set_err(LPAMD_E_STATE, "a pivot set up by lpamd_lookahead_setup still waits for lpamd_lookahead_update"); \
  } while (0)

}  // namespace

extern "C" {

const char* lpamd_last_error(void) { return g_err; }
const char* lpamd_version(void) { return "lpamd 0.1 (gfx950)"; }
int64_t lpamd_exchange_bytes(int64_t m) { return round_up(std::max<int64_t>(m, 1), kSlabRows) * (int64_t)sizeof(double); }

static int create_impl(lpamd_engine* e, int64_t m, int64_t N, int64_t nloc, int32_t rank, int32_t world, void* hip_stream);

// Working buffer of an engine: allocated and CLEARED.  hipMalloc hands out whatever an earlier owner left (a freed array of
// pointers reads as doubles of 1e-310 or 1e300), and a kernel that reads an entry before the first launch that writes it
// must see the same bits in every run; zero is what fresh device memory holds and what every parity test ran on.
// LPAMD_POISON=<field name>|all fills the named buffer with bytes of 0x7F instead (huge doubles and indices): a test that
// changes its outcome under it reads that buffer before anything writes it.
static int dev_alloc(lpamd_engine* e, void** out, size_t bytes, const char* name) {
  HIP_TRY(hipMalloc(out, bytes));
  const char* poison = getenv("LPAMD_POISON");
  const int fill = (poison && (!strcmp(poison, "all") || !strcmp(poison, name))) ? 0x7F : 0;
  HIP_TRY(hipMemsetAsync(*out, fill, bytes, e->stream));
  return LPAMD_OK;
}

int lpamd_create(lpamd_engine** out, int64_t m, int64_t N, int32_t rank, int32_t world, void* hip_stream) {
  if (!out) return set_err(LPAMD_E_ARG, "out is null");
  *out = nullptr;
  if (m <= 0 || N <= 0 || world <= 0 || rank < 0 || rank >= world)
    return set_err(LPAMD_E_ARG, "bad shape m=%lld N=%lld rank=%d world=%d", (long long)m, (long long)N, rank, world);
  const int64_t n_ord = N + 1;  // variables + artificial
  const int64_t nloc = (n_ord - rank + world - 1) / world;
  if (nloc <= 0) return set_err(LPAMD_E_ARG, "more ranks than columns");
  if (nloc + 1 > (int64_t)std::numeric_limits<int32_t>::max() / 2) return set_err(LPAMD_E_ARG, "too many columns per rank");
  // the kernels keep column ids and rows in 32 bits where that saves instructions (reductions, owner and local column of an id)
  if (N + 2 > (int64_t)std::numeric_limits<int32_t>::max() || m > (int64_t)std::numeric_limits<int32_t>::max())
    return set_err(LPAMD_E_ARG, "m and N must fit 31 bits");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return set_err(LPAMD_E_NODEVICE, "no HIP device visible");
  lpamd_engine* e = new lpamd_engine();
  const int rc = create_impl(e, m, N, nloc, rank, world, hip_stream);
  if (rc != LPAMD_OK) {
    char keep[sizeof(g_err)];
    std::memcpy(keep, g_err, sizeof(keep));   // lpamd_destroy must not clobber the cause
    lpamd_destroy(e);
    std::memcpy(g_err, keep, sizeof(keep));
    return rc;
  }
  *out = e;
  return LPAMD_OK;
}

static int create_impl(lpamd_engine* e, int64_t m, int64_t N, int64_t nloc, int32_t rank, int32_t world, void* hip_stream) {
  HIP_TRY(hipGetDevice(&e->device));
  e->sh.m = m;
  e->sh.N = N;
  e->sh.lda = round_up(m, kSlabRows);
  e->sh.nloc = (int32_t)nloc;
  e->sh.rank = rank;
  e->sh.world = world;
  e->ncl = (int32_t)nloc + 1;
  e->n_slabs = (int32_t)(e->sh.lda / kSlabRows);
  // Measured on MI355X (bench.py sweeps, profiles/): shards beyond the 256 MiB Infinity
  // Cache stream 11 % faster with nt loads/stores, shards inside it 13 % slower; one
  // workgroup per column (up to 8192) balances short columns best, long columns do not care.
  e->update_nt = (double)e->sh.lda * (double)e->ncl * sizeof(double) > 512e6;
  e->update_grid = std::max(1, std::min<int32_t>(e->ncl, 8192));
  if (const char* env = getenv("LPAMD_UPDATE_NT")) e->update_nt = atoi(env) != 0;
  // col' loaded together with the tableau batch (before the store burst) pays inside the Infinity
  // Cache (+3 %); HBM-sized shards stream 1.5-3 % faster when it is read after the loads are out
  e->update_prec = e->update_prec_fused = !e->update_nt;
  if (const char* env = getenv("LPAMD_UPDATE_UNROLL")) e->update_unroll_cap = std::max(1, atoi(env));
  if (const char* env = getenv("LPAMD_UPDATE_PREC")) e->update_prec = atoi(env) != 0;
  if (const char* env = getenv("LPAMD_UPDATE_PREC_FUSED")) e->update_prec_fused = atoi(env) != 0;
  if (const char* env = getenv("LPAMD_FUSE_PR")) e->fuse_pr_enabled = atoi(env) != 0;
  if (const char* env = getenv("LPAMD_UPDATE_GRID")) e->update_grid = std::max(1, atoi(env));
  // Measured: the ratio test inside every set-up workgroup beats a third launch on the reference's own
  // inputs (m up to ~1000: israel 14.7 -> 12.6 ms, fit1d 45 -> 41 ms per solve) and loses at m = 4096
  // (9460 against 9780 pivots/s), so it is taken up to 2048 rows
  e->two_launch = world == 1 && m <= 2048;
  if (const char* env = getenv("LPAMD_TWO_LAUNCH")) e->two_launch = world == 1 && atoi(env) != 0;
  // Longer columns: a pivot is bound by the HBM pass, so a chain of pivots shares one (see k_update_chain)
  // (column shards chain through lpamd_chain_step, the caller runs the exchange of every step)
  // Depth: 16 steps still hide under the HBM time of the pass (k_update_slab, col' in registers), and so do about 20
  // with the col' slices in LDS (k_update_deep).  Beyond that the fp64 vector rate (two instructions per entry and
  // step plus the magnitude tracking) shows and the pass grows by 0.1 .. 0.2 ms per step at m=32768 n=65536 while a
  // step's set-up costs 0.03 ms: 22 steps cost 5.8 ms per pass against 5.65 (0.73 of the HBM peak, 3420 pivots/s
  // against 2600 at 16), 23 steps 5.9 .. 6.1 ms (0.70 .. 0.73 from one machine to the next), 32 steps 7.5 ms (0.57 of
  // the peak, 3800 pivots/s).  The default is the deepest chain that keeps the pass clear of 0.70 of the peak on
  // every machine seen.  Shards that fit the Infinity Cache are bound by the set-up launches and stay at 16.
  // Every rank of a deal must take the same depth (a step is a collective), so the size that decides it is that of the
  // LARGEST shard of the deal (rank 0's), not this rank's own: shards differ by one column when N + 1 is no multiple of the world.
  const double deal_bytes = (double)e->sh.lda * (double)((N + 1 + world - 1) / world + 1) * sizeof(double);
  const bool hbm_sized = deal_bytes > 512e6;
  // Column shards of that size chain 26: their steps carry the export and the exchange on top of the set-up (20 us per step
  // beside 34 us of pass per pivot on a shard of 8), so the depth that pays is greater than on one GPU (rank 0 of 8: 18.7 k
  // pivots/s at 22, 19.7 k at 26, 19.8 k at 28 and 32; of 2: 6.35 k, 6.78 k, 6.95 k), at 0.67 of the HBM peak per rank.
  // One GPU, shard inside the Infinity Cache: 24 steps with col' in registers (m=4096 n=8192: 59.5 k pivots/s against 57.9 k at
  // 16; the pass costs 7.0 us per pivot instead of 7.6 and its fixed costs spread over more pivots); MAX_REDUCTION keeps 16
  // there (20.6 k at 24 against 21.4 k: its steps grow dearer with their position in the pass).
  const bool cache_sized_one_gpu = world == 1 && !hbm_sized;
  e->chain_depth = e->two_launch ? 1 : (hbm_sized ? (world > 1 ? 26 : 22) : (cache_sized_one_gpu ? 24 : 16));
  // One GPU, HBM-sized shard: the depth is MEASURED on this device once everything is allocated (probe_chain_depth), from
  // kProbeMaxDepth down; the buffers are sized for that.  Column shards keep their constant: every rank must chain the same number of steps.
  bool probe_depth = world == 1 && hbm_sized && !e->two_launch;
  if (const char* env = getenv("LPAMD_DEPTH_PROBE")) probe_depth = probe_depth && atoi(env) != 0;
  if (const char* env = getenv("LPAMD_CHAIN")) {
    e->chain_depth = std::max(1, std::min(kMaxChain, atoi(env)));
    if (e->chain_depth > 1) e->two_launch = false;
    probe_depth = false;
  }
  if (probe_depth) e->chain_depth = kProbeMaxDepth;
  // MAX_REDUCTION on one GPU, HBM-sized: a step's set-up costs about 65 us against 17 for MAX_COST, and the pass kernel
  // still gains per pivot up to 28 steps (247 us per pivot against 273 at 22, m=32768 n=65536), so those passes go
  // deeper than the depth that keeps the pass kernel at its best fraction of the HBM peak.
  e->mr_depth = (probe_depth && e->sh.lda % (2 * kWave) == 0) ? 28 : e->chain_depth;
  // column shards: a MAX_REDUCTION step's cost grows with its position in the pass (six launches that each walk the steps so
  // far); on a 3 GB shard, where the set-up outweighs the pass, 22 per pass measured faster than 26 (rank 0 of 8 of m=32768
  // n=65536: 10.9 k against 10.0 k pivots/s), on a 21 GB shard slower (rank 0 of 8 of m=65536 n=262144: 2.70 k against 2.95 k)
  if (world > 1 && hbm_sized && !getenv("LPAMD_CHAIN") && deal_bytes < 8e9)
    e->mr_depth = std::min(e->chain_depth, 22);
  if (cache_sized_one_gpu && !getenv("LPAMD_CHAIN")) e->mr_depth = std::min(e->chain_depth, 16);
  if (const char* env = getenv("LPAMD_MR_CHAIN")) { if (e->chain_depth > 1) e->mr_depth = std::max(2, std::min(kMaxChain, atoi(env))); }
  e->alloc_depth = std::max(e->chain_depth, e->mr_depth);
  // Measured (m=32768 n=65536): col' from L2 8.44 ms per pass of 8 pivots; in registers, 1024-row slabs x 8 columns in
  // flight 5.65 ms per pass of 16 pivots (the HBM time of one pass), x 4 columns 5.91 ms, 512-row slabs x 8 columns 6.64 ms
  // shards inside the Infinity Cache (m=4096 n=8192: 0.134 -> 0.116 ms per pass of 16 pivots) take the 512-row slabs as well: twice the workgroups
  e->slab_shape = (e->sh.lda % 1024 == 0 && hbm_sized) ? 4 : 1;
  if (const char* env = getenv("LPAMD_SLAB")) e->slab_shape = atoi(env);
  if (const char* env = getenv("LPAMD_SLAB_GRID")) e->slab_grid = std::max(1, atoi(env));
  // col' in LDS (k_update_deep) for chains beyond 16 steps on HBM-sized shards; inside the Infinity Cache the register kernel
  // serves up to 24 steps and is twice as fast there (0.167 against 0.278 ms per pass of 24 at m=4096 n=8192)
  e->deep_batch = (e->alloc_depth > (hbm_sized ? 16 : 24) && e->sh.lda % (2 * kWave) == 0) ? 8 : 0;
  if (const char* env = getenv("LPAMD_DEEP")) { e->deep_batch = (e->chain_depth > 1 && e->sh.lda % (2 * kWave) == 0) ? atoi(env) : 0; e->deep_any_depth = true; }
  if (e->deep_batch != 4 && e->deep_batch != 8) e->deep_batch = 0;
  if (const char* env = getenv("LPAMD_DEEP_GRID")) e->deep_grid = std::max(0, atoi(env));
  if (hip_stream) {
    e->stream = (hipStream_t)hip_stream;
  } else if (const char* env = getenv("LPAMD_STREAM_CU_RANGE")) {
    // experiments only ("first:last"): the engine's own stream restricted to CUs first .. last-1 of the device
    int first = 0, last = 0;
    if (sscanf(env, "%d:%d", &first, &last) != 2 || first < 0 || last <= first || last > 1024)
      return set_err(LPAMD_E_ARG, "LPAMD_STREAM_CU_RANGE wants first:last");
    std::vector<uint32_t> mask((last + 31) / 32, 0u);
    for (int c = first; c < last; c++) mask[c / 32] |= 1u << (c % 32);
    HIP_TRY(hipExtStreamCreateWithCUMask(&e->stream, (uint32_t)mask.size(), mask.data()));
    e->own_stream = true;
  } else {
    HIP_TRY(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    e->own_stream = true;
  }
  LP_TRY(allocate_tableau(e));
  LP_TRY(dev_alloc(e, (void**)&e->obj, sizeof(double) * e->ncl, "obj"));
  LP_TRY(dev_alloc(e, (void**)&e->is_basic, e->ncl, "is_basic"));
  LP_TRY(dev_alloc(e, (void**)&e->basis, sizeof(int64_t) * m, "basis"));
  e->rstride = std::max<int64_t>(e->ncl, e->sh.lda);
  LP_TRY(dev_alloc(e, (void**)&e->colp, sizeof(double) * e->sh.lda * e->alloc_depth, "colp"));
  LP_TRY(dev_alloc(e, (void**)&e->rowp, sizeof(double) * e->rstride * e->alloc_depth, "rowp"));
  LP_TRY(dev_alloc(e, (void**)&e->active, sizeof(int32_t) * e->ncl, "active"));
  LP_TRY(dev_alloc(e, (void**)&e->cmin, sizeof(double) * e->ncl, "cmin"));
  LP_TRY(dev_alloc(e, (void**)&e->crow, sizeof(int64_t) * e->ncl, "crow"));
  LP_TRY(dev_alloc(e, (void**)&e->pr_stamp, sizeof(int64_t) * e->ncl, "pr_stamp"));
  LP_TRY(dev_alloc(e, (void**)&e->rhs_next, sizeof(double) * e->sh.lda * e->alloc_depth, "rhs_next"));
  LP_TRY(dev_alloc(e, (void**)&e->cand_next, e->ncl, "cand_next"));
  HIP_TRY(hipMemsetAsync(e->pr_stamp, 0xFF, sizeof(int64_t) * e->ncl, e->stream));
  HIP_TRY(hipMemsetAsync(e->rhs_next, 0, sizeof(double) * e->sh.lda * e->alloc_depth, e->stream));
  HIP_TRY(hipMemsetAsync(e->cand_next, 0, e->ncl, e->stream));
  LP_TRY(dev_alloc(e, (void**)&e->steps, sizeof(ChainStep) * kMaxChain, "steps"));
  HIP_TRY(hipMemsetAsync(e->steps, 0, sizeof(ChainStep) * kMaxChain, e->stream));
  LP_TRY(dev_alloc(e, (void**)&e->colv, sizeof(double) * 2 * e->sh.lda, "colv"));   // k_chain_fused keeps two copies that take turns
  // MAX_REDUCTION prices from bounds wherever pivots chain
  e->pr_bounds = e->chain_depth > 1;
  if (const char* env = getenv("LPAMD_PR_BOUNDS")) e->pr_bounds = e->pr_bounds && atoi(env) != 0;
  if (e->pr_bounds) {
    LP_TRY(dev_alloc(e, (void**)&e->prb_info, sizeof(PrbInfo), "prb_info"));
    HIP_TRY(hipMemsetAsync(e->prb_info, 0, sizeof(PrbInfo), e->stream));
    LP_TRY(dev_alloc(e, (void**)&e->prb_c, sizeof(double) * kMaxChain * kPrbRows, "prb_c"));
    LP_TRY(dev_alloc(e, (void**)&e->prb_cache, sizeof(double) * kPrbRows * (size_t)std::max<int64_t>(1, e->sh.nloc), "prb_cache"));
    LP_TRY(dev_alloc(e, (void**)&e->prb_v, sizeof(double) * 3 * kMaxPartials, "prb_v"));
    LP_TRY(dev_alloc(e, (void**)&e->prb_i, sizeof(int64_t) * 3 * kMaxPartials, "prb_i"));
    LP_TRY(dev_alloc(e, (void**)&e->prb_ep_v, sizeof(double) * kPrbSliced * kPrbSlices, "prb_ep_v"));
    LP_TRY(dev_alloc(e, (void**)&e->prb_ep_i, sizeof(int64_t) * kPrbSliced * kPrbSlices, "prb_ep_i"));
    if (const char* env = getenv("LPAMD_PRB_ROWS")) e->prb_rows = std::max(1, std::min(kPrbRows, atoi(env)));
    if (const char* env = getenv("LPAMD_PRB_CACHE")) e->prb_use_cache = atoi(env) != 0;
    if (const char* env = getenv("LPAMD_PRB_RENEW")) e->prb_renew = std::max(1, atoi(env));
    if (const char* env = getenv("LPAMD_PRB_CARRY")) e->prb_carry = atoi(env) != 0;
    if (const char* env = getenv("LPAMD_PRB_BOUND_BLOCKS")) e->prb_bound_blocks = std::max(0, atoi(env));
    LP_TRY(dev_alloc(e, (void**)&e->prb_ub, sizeof(double) * e->ncl, "prb_ub"));
    LP_TRY(dev_alloc(e, (void**)&e->prb_ucand, e->ncl, "prb_ucand"));
    LP_TRY(dev_alloc(e, (void**)&e->prb_elist, sizeof(int32_t) * e->ncl, "prb_elist"));
    LP_TRY(dev_alloc(e, (void**)&e->prb_escore, sizeof(double) * e->ncl, "prb_escore"));
    LP_TRY(dev_alloc(e, (void**)&e->prb_erow, sizeof(int64_t) * e->ncl, "prb_erow"));
    // k_prb_tail: every workgroup of its grid must be resident at the same time (they wait for each other's records); the
    // grid may take three quarters of what the device holds of the kernel, and 512 workgroups at most (record polls)
    e->prb_tail = true;   // (column shards: the launch ends with the candidate packed for the exchange)
    if (const char* env = getenv("LPAMD_PRB_TAIL")) e->prb_tail = e->prb_tail && atoi(env) != 0;
    if (e->prb_tail) {
      int per_cu = 0, cus = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_prb_tail<8, false>, kPartialThreads, 0) != hipSuccess ||
          hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device) != hipSuccess) { (void)hipGetLastError(); per_cu = 0; }
      const int64_t room = std::min<int64_t>(kTailMaxGroups, (int64_t)per_cu * cus * 3 / 4);
      e->tail_rows = (int32_t)std::min<int64_t>(128, std::max<int64_t>(1, (m + kPartialThreads - 1) / kPartialThreads));
      e->tail_cols = (int32_t)std::min<int64_t>(room - e->tail_rows, std::max<int64_t>(1, (e->ncl + kPartialThreads - 1) / kPartialThreads));
      if (const char* env = getenv("LPAMD_TAIL_ROWS")) e->tail_rows = std::max(1, std::min(128, atoi(env)));
      if (const char* env = getenv("LPAMD_TAIL_COLS")) e->tail_cols = std::max(1, std::min<int>((int)room - e->tail_rows, atoi(env)));
      if (e->tail_cols < 1 || e->tail_rows + e->tail_cols < 2 * kTailCols) e->prb_tail = false;
    }
    if (e->prb_tail) {
      const int grid = e->tail_rows + e->tail_cols;
      const size_t words = (size_t)kFusedReplicas * (fused_region_words(grid, kTailEvalWords) + fused_region_words(grid, kTailBackWords));
      LP_TRY(dev_alloc(e, (void**)&e->prb_top_u, sizeof(double) * 2 * kMaxPartials, "prb_top_u"));
      LP_TRY(dev_alloc(e, (void**)&e->prb_top_l, sizeof(int32_t) * kMaxPartials, "prb_top_l"));
      LP_TRY(dev_alloc(e, (void**)&e->rec_tail, sizeof(uint64_t) * words, "rec_tail"));
      HIP_TRY(hipMemsetAsync(e->rec_tail, 0, sizeof(uint64_t) * words, e->stream));
    }
  }
  // Steps 1.. of a MAX_COST chain in one launch (k_chain_fused): one GPU, chains only.  Preferred shape: one row / one column
  // per thread (history in registers) with the smallest workgroups that keep the grid within 512 workgroups, which the
  // chip holds at once with room to spare; shards too wide for that give a thread several columns (history from memory).
  e->fused = world == 1 && e->chain_depth > 1;
  if (const char* env = getenv("LPAMD_FUSED")) e->fused = e->fused && atoi(env) != 0;
  if (e->fused) {
    // limits of the kernel: at most 128 row workgroups (their records are folded by one wavefront, two per lane), at most two
    // column workgroups per thread of a workgroup (their records are folded by a whole row workgroup), 32 columns per thread
    auto blocks = [](int64_t items, int64_t per) { return (items + per - 1) / per; };
    int threads = 256;
    while (threads < 1024 && (blocks(m, threads) > 128 || blocks(m, threads) + blocks(e->ncl, threads) > 520)) threads *= 2;
    if (const char* env = getenv("LPAMD_FUSED_THREADS")) { const int t = atoi(env); if (t == 256 || t == 512 || t == 1024) threads = t; }
    e->fused_threads = threads;
    e->fused_rows = (int32_t)std::min<int64_t>(128, blocks(m, threads));
    e->fused_cols = (int32_t)std::min<int64_t>(392, blocks(e->ncl, threads));
    if (const char* env = getenv("LPAMD_FUSED_ROWS")) e->fused_rows = std::max(1, std::min(128, atoi(env)));
    if (const char* env = getenv("LPAMD_FUSED_COLS")) e->fused_cols = std::max(1, std::min(2 * threads, atoi(env)));
    e->fused_reg = threads < 1024 && (int64_t)e->fused_rows * threads >= m && (int64_t)e->fused_cols * threads >= e->ncl;   // 1024 threads leave 128 registers each: the history would spill
    if (const char* env = getenv("LPAMD_FUSED_REG")) e->fused_reg = e->fused_reg && atoi(env) != 0;
    // Every workgroup of the grid must be resident at the same time (they wait for each other): the grid may take three
    // quarters of what the device holds of this kernel; otherwise the leaner variant (history from memory), and if that does
    // not fit either, the step-by-step kernels.
    auto capacity = [&](bool reg) -> int64_t {
      int per_cu = 0, cus = 0;
      hipError_t err = hipSuccess;
#define LPAMD_OCC(THREADS, REG, HIST) err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_chain_fused<THREADS, REG, HIST>, THREADS, 0)
      const bool h24 = e->chain_depth <= 24;
      if (threads == 1024) LPAMD_OCC(1024, false, 32);
      else if (threads == 512) { if (!reg) LPAMD_OCC(512, false, 32); else if (h24) LPAMD_OCC(512, true, 24); else LPAMD_OCC(512, true, 32); }
      else { if (!reg) LPAMD_OCC(256, false, 32); else if (h24) LPAMD_OCC(256, true, 24); else LPAMD_OCC(256, true, 32); }
#undef LPAMD_OCC
      if (err != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device) != hipSuccess) { (void)hipGetLastError(); return 0; }
      return (int64_t)per_cu * cus;
    };
    const int64_t grid = e->fused_rows + e->fused_cols;
    if (e->fused_reg && grid * 4 > capacity(true) * 3) e->fused_reg = false;
    if (!e->fused_reg && grid * 4 > capacity(false) * 3) e->fused = false;
    if (blocks(e->ncl, (int64_t)e->fused_cols * threads) > 32) e->fused = false;
  }
  if (e->fused) {
    LP_TRY(dev_alloc(e, (void**)&e->rec_ratio, sizeof(uint64_t) * 2 * kFusedReplicas * fused_region_words(e->fused_rows, kFusedRatioWords), "rec_ratio"));
    LP_TRY(dev_alloc(e, (void**)&e->rec_price, sizeof(uint64_t) * 2 * kFusedReplicas * fused_region_words(e->fused_cols, kFusedPriceWords), "rec_price"));
    HIP_TRY(hipMemsetAsync(e->rec_ratio, 0, sizeof(uint64_t) * 2 * kFusedReplicas * fused_region_words(e->fused_rows, kFusedRatioWords), e->stream));
    HIP_TRY(hipMemsetAsync(e->rec_price, 0, sizeof(uint64_t) * 2 * kFusedReplicas * fused_region_words(e->fused_cols, kFusedPriceWords), e->stream));
  }
  LP_TRY(dev_alloc(e, (void**)&e->pass, sizeof(PassInfo), "pass"));
  HIP_TRY(hipMemsetAsync(e->pass, 0, sizeof(PassInfo), e->stream));
  if (e->chain_depth > 1) {
    LP_TRY(dev_alloc(e, (void**)&e->list_full, sizeof(int32_t) * e->ncl, "list_full"));
    LP_TRY(dev_alloc(e, (void**)&e->list_part, sizeof(int32_t) * e->ncl, "list_part"));
    LP_TRY(dev_alloc(e, (void**)&e->rsl_full, sizeof(double) * (size_t)(e->ncl + 8) * e->alloc_depth, "rsl_full"));   // whole batches of up to 8 columns
    LP_TRY(dev_alloc(e, (void**)&e->rsl_part, sizeof(double) * (size_t)e->ncl * e->alloc_depth, "rsl_part"));
    LP_TRY(dev_alloc(e, (void**)&e->ptr_full, sizeof(uint64_t) * (size_t)(e->ncl + 8), "ptr_full"));
  }
  HIP_TRY(hipMemsetAsync(e->colv, 0, sizeof(double) * 2 * e->sh.lda, e->stream));    // rows beyond m stay zero
  LP_TRY(dev_alloc(e, (void**)&e->cand, sizeof(Candidate), "cand"));
  LP_TRY(dev_alloc(e, (void**)&e->cand_la, sizeof(Candidate), "cand_la"));
  HIP_TRY(hipMemsetAsync(e->cand_la, 0xFF, sizeof(Candidate), e->stream));   // gcol = -1: no candidate
  LP_TRY(dev_alloc(e, (void**)&e->part_v, sizeof(double) * 2 * kMaxPartials, "part_v"));
  LP_TRY(dev_alloc(e, (void**)&e->part_i, sizeof(int64_t) * 3 * kMaxPartials, "part_i"));
  // partial reductions: 256 columns / 256 rows per workgroup, strided beyond kMaxPartials workgroups
  e->n_price_part = (int32_t)std::min<int64_t>(kMaxPartials, std::max<int64_t>(1, (nloc + 1 + kPartialThreads - 1) / kPartialThreads));
  e->n_ratio_part = (int32_t)std::min<int64_t>(kMaxPartials, std::max<int64_t>(1, (m + kPartialThreads - 1) / kPartialThreads));
  e->n_row_blocks = (int32_t)std::min<int64_t>(256, std::max<int64_t>(1, (m + kPartialThreads - 1) / kPartialThreads));   // one row per thread up to 65536 rows
  if (const char* env = getenv("LPAMD_ROW_BLOCKS")) e->n_row_blocks = std::max(1, std::min(1024, atoi(env)));
  // k_shard_step: one row per thread in the row workgroups, few enough column workgroups for one poll, and the whole grid
  // resident at once (the row workgroups wait for the column workgroups)
  e->shard_step = world > 1 && e->chain_depth > 1 && (int64_t)e->n_row_blocks * kPartialThreads >= m && e->n_price_part <= kShardMaxColGroups;
  if (const char* env = getenv("LPAMD_SHARD_STEP")) e->shard_step = e->shard_step && atoi(env) != 0;
  if (e->shard_step) {
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_shard_step<true, true>, kPartialThreads, 0) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device) != hipSuccess) { (void)hipGetLastError(); per_cu = 0; }
    if ((int64_t)(e->n_row_blocks + e->n_price_part) * 4 > (int64_t)per_cu * cus * 3) e->shard_step = false;
  }
  if (e->shard_step) {
    const size_t words = (size_t)kFusedReplicas * fused_region_words(e->n_price_part, kShardRecWords);
    LP_TRY(dev_alloc(e, (void**)&e->rec_shard, sizeof(uint64_t) * words, "rec_shard"));
    HIP_TRY(hipMemsetAsync(e->rec_shard, 0, sizeof(uint64_t) * words, e->stream));
  }
  LP_TRY(dev_alloc(e, (void**)&e->st, sizeof(PivotState), "st"));
  HIP_TRY(hipMemsetAsync(e->colp, 0, sizeof(double) * e->sh.lda * e->alloc_depth, e->stream));
  HIP_TRY(hipMemsetAsync(e->st, 0, sizeof(PivotState), e->stream));
  std::memset(&e->h_st, 0, sizeof(PivotState));
  LP_TRY(ensure_trace(e, 4096));
  HIP_TRY(hipStreamSynchronize(e->stream));
  if (probe_depth && e->deep_batch > 0) LP_TRY(probe_chain_depth(e));
  return LPAMD_OK;
}

int lpamd_destroy(lpamd_engine* e) {
  if (!e) return LPAMD_OK;
  (void)hipSetDevice(e->device);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  if (getenv("LPAMD_STATE_VERBOSE") && e->st && pull_state(e) == LPAMD_OK)
    fprintf(stderr, "lpamd: %lld pivots, status %d, basic columns %s\n", (long long)e->h_st.iter, (int)e->h_st.status,
            e->h_st.dirty ? "no longer unit columns (the loop that scans rows like the reference took over)" : "unit columns");
  if (e->prb_info && getenv("LPAMD_PRB_VERBOSE")) {
    PrbInfo pi;
    if (hipMemcpy(&pi, e->prb_info, sizeof(pi), hipMemcpyDeviceToHost) == hipSuccess)
      fprintf(stderr, "lpamd: %lld pricings from bounds, %lld columns evaluated besides the first, %lld second rounds in k_prb_tail\n",
              (long long)pi.priced, (long long)pi.listed, (long long)pi.tail_back);
  }
  for (hipEvent_t ev : e->ev) (void)hipEventDestroy(ev);
  void* ptrs[] = {e->T.half[0], e->T.half[1], e->obj, e->is_basic, e->basis, e->colp, e->rowp, e->active, e->cmin, e->crow, e->pr_stamp, e->rhs_next, e->cand_next, e->cand, e->cand_la, e->part_v, e->part_i, e->st, e->trace, e->steps, e->colv, e->pass, e->list_full, e->list_part, e->rsl_full, e->rsl_part, e->ptr_full,
                  e->rec_ratio, e->rec_price, e->first_basic, e->basis_disc, e->prb_info, e->prb_c, e->prb_cache, e->prb_v, e->prb_i, e->prb_ep_v, e->prb_ep_i, e->prb_ub, e->prb_ucand, e->prb_elist, e->prb_escore, e->prb_erow, e->prb_top_u, e->prb_top_l, e->rec_tail, e->rec_shard};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (e->own_stream) (void)hipStreamDestroy(e->stream);
  delete e;
  return LPAMD_OK;
}

int lpamd_upload(lpamd_engine* e, const double* T, int64_t ld, const double* obj, const uint8_t* is_basic) {
  CHECK_ENGINE(e);
  CHECK_NO_PENDING_UPDATE(e);
  const Shape& sh = e->sh;
  if (!T || !obj || !is_basic || ld < sh.m) return set_err(LPAMD_E_ARG, "upload: null buffer or ld < m");
  std::vector<double> hT[2], hobj(e->ncl, 0.0);
  for (int h = 0; h < 2; h++) hT[h].assign((size_t)e->T.lda * e->half_cols[h], 0.0);
  std::vector<uint8_t> hb(e->ncl, 0);
  for (int64_t l = 0; l < e->ncl; l++) {
    const int64_t g = global_of(e, l);
    std::memcpy(&hT[l & 1][(size_t)(l >> 1) * e->T.lda], T + g * ld, sizeof(double) * sh.m);
    hobj[l] = obj[g];
    hb[l] = (g == sh.N || g == sh.N + 1) ? 0 : (is_basic[g] ? 1 : 0);
  }
  // basic variable of every row: the discovery loop of reference
  // lp_simplex_tableau.cc:240-251 (ascending basic columns, first non-zero row,
  // later columns overwrite earlier ones), done once on the full host matrix.
  std::vector<int64_t> basis(sh.m, -1);
  for (int64_t g = 0; g < sh.N; g++) {
    if (!is_basic[g]) continue;
    for (int64_t i = 0; i < sh.m; i++)
      if (!h_is_zero(T[g * ld + i])) { basis[i] = g; break; }
  }
  // Stored entries the zero drop would not have let through (0 < |x| < 1e-6; the reference keeps right-hand sides however
  // small, reference lp.cc:180): the first pivot then reaches every column, see PivotState::dense_next.  Looked for in the
  // WHOLE matrix, not only this rank's columns, so that all ranks of a group agree.
  bool tiny = false;
  for (int64_t g = 0; g < sh.N + 2 && !tiny; g++)
    for (int64_t i = 0; i < sh.m; i++) {
      const double x = T[g * ld + i];
      if (x != 0.0 && std::fabs(x) < (double)1e-6f) { tiny = true; break; }
    }
  e->tab_version++;
  for (int h = 0; h < 2; h++)
    if (!hT[h].empty())
      HIP_TRY(hipMemcpyAsync(e->T.half[h], hT[h].data(), sizeof(double) * hT[h].size(), hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipMemcpyAsync(e->obj, hobj.data(), sizeof(double) * e->ncl, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipMemcpyAsync(e->is_basic, hb.data(), e->ncl, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipMemcpyAsync(e->basis, basis.data(), sizeof(int64_t) * sh.m, hipMemcpyHostToDevice, e->stream));
  // The device loops rely on every basic column being a unit column, one per row (PivotState::dirty); a tableau that does
  // not start that way is solved by the loop that scans like the reference from the first pivot on.
  bool unit_basis = true;
  {
    std::vector<uint8_t> covered(sh.m, 0);
    int64_t n_basic = 0;
    for (int64_t g = 0; g < sh.N && unit_basis; g++) {
      if (!is_basic[g]) continue;
      n_basic++;
      int64_t nz = 0, at = -1;
      for (int64_t i = 0; i < sh.m; i++)
        if (!h_is_zero(T[g * ld + i])) { nz++; at = i; }
      if (nz != 1 || covered[at]) unit_basis = false;
      else covered[at] = 1;
    }
    if (n_basic != sh.m) unit_basis = false;
  }
  reset_state(e);
  e->dirty_basis = false;
  e->h_st.dirty = unit_basis ? 0 : 1;
  e->h_st.dense_next = tiny ? 1 : 0;
  e->dense_first = tiny;
  LP_TRY(invalidate_pr(e));
  LP_TRY(push_state(e));
  e->loaded = true;
  return LPAMD_OK;
}

int lpamd_fill_synthetic(lpamd_engine* e, int64_t n_struct, uint64_t seed) {
  CHECK_ENGINE(e);
  CHECK_NO_PENDING_UPDATE(e);
  const Shape& sh = e->sh;
  if (n_struct <= 0 || n_struct + sh.m != sh.N)
    return set_err(LPAMD_E_ARG, "synthetic LP needs N == n_struct + m (N=%lld n=%lld m=%lld)", (long long)sh.N, (long long)n_struct, (long long)sh.m);
  e->tab_version++;
  hipLaunchKernelGGL(k_fill_synthetic, dim3((unsigned)e->ncl, (unsigned)((sh.lda + 255) / 256)), dim3(256), 0, e->stream, e->T, e->obj,
                     e->is_basic, e->basis, sh, n_struct, seed);
  LP_TRY(check_launch("k_fill_synthetic"));
  reset_state(e);
  e->dense_first = false;
  e->dirty_basis = false;
  LP_TRY(invalidate_pr(e));
  LP_TRY(push_state(e));
  e->loaded = true;
  return LPAMD_OK;
}

int lpamd_download(const lpamd_engine* e, double* T, int64_t ld, double* obj, uint8_t* is_basic) {
  CHECK_LOADED(e);
  const Shape& sh = e->sh;
  if (T) {
    if (ld < sh.m) return set_err(LPAMD_E_ARG, "download: ld < m");
    for (int h = 0; h < 2; h++) {
      if (e->half_cols[h] == 0) continue;
      std::vector<double> hT((size_t)e->T.lda * e->half_cols[h]);
      LP_TRY(read_local(e, e->T.half[h], hT.data(), sizeof(double) * hT.size()));
      for (int64_t l = h; l < e->ncl; l += 2)
        std::memcpy(T + global_of(e, l) * ld, &hT[(size_t)(l >> 1) * e->T.lda], sizeof(double) * sh.m);
    }
  }
  if (obj) {
    std::vector<double> h(e->ncl);
    LP_TRY(read_local(e, e->obj, h.data(), sizeof(double) * e->ncl));
    for (int64_t l = 0; l < e->ncl; l++) obj[global_of(e, l)] = h[l];
  }
  if (is_basic) {
    std::vector<uint8_t> h(e->ncl);
    LP_TRY(read_local(e, e->is_basic, h.data(), e->ncl));
    for (int64_t l = 0; l < e->sh.nloc; l++) is_basic[global_of(e, l)] = h[l];
  }
  return LPAMD_OK;
}

int lpamd_read_column(const lpamd_engine* e, int64_t col, double* out) {
  CHECK_LOADED(e);
  if (col < 0 || col > e->sh.N + 1 || !out) return set_err(LPAMD_E_ARG, "read_column: bad column %lld", (long long)col);
  if (!owns(e, col)) return set_err(LPAMD_E_ARG, "read_column: column %lld lives on another rank", (long long)col);
  return read_local(e, e->T.col(local_of(e, col)), out, sizeof(double) * e->sh.m);
}

int lpamd_read_row(const lpamd_engine* e, int64_t row, double* out) {
  CHECK_LOADED(e);
  if (row < 0 || row >= e->sh.m || !out) return set_err(LPAMD_E_ARG, "read_row: bad row %lld", (long long)row);
  std::vector<double> h;
  LP_TRY(gather_row_host(e, row, h));
  for (int64_t l = 0; l < e->ncl; l++) out[global_of(e, l)] = h[l];
  return LPAMD_OK;
}

int lpamd_write_objective(lpamd_engine* e, const double* obj) {
  CHECK_LOADED(e);
  CHECK_NO_PENDING_UPDATE(e);
  e->price_partials_fresh = false;
  if (!obj) return set_err(LPAMD_E_ARG, "write_objective: null");
  std::vector<double> h(e->ncl);
  for (int64_t l = 0; l < e->ncl; l++) h[l] = obj[global_of(e, l)];
  HIP_TRY(hipMemcpyAsync(e->obj, h.data(), sizeof(double) * e->ncl, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return LPAMD_OK;
}

int lpamd_basis_of_row(const lpamd_engine* e, int64_t* out) {
  CHECK_LOADED(e);
  if (!out) return set_err(LPAMD_E_ARG, "basis_of_row: null");
  return read_local(e, e->basis, out, sizeof(int64_t) * e->sh.m);
}

int lpamd_need_init(lpamd_engine* e, int32_t* out) {
  CHECK_LOADED(e);
  if (!out) return set_err(LPAMD_E_ARG, "need_init: null");
  std::vector<double> rhs(e->sh.m);
  LP_TRY(read_local(e, e->T.col(e->sh.nloc), rhs.data(), sizeof(double) * e->sh.m));
  *out = 0;
  for (double b : rhs)
    if (h_is_neg(b)) { *out = 1; break; }
  return LPAMD_OK;
}

static int read_candidate(lpamd_engine* e, Candidate* c) { return read_local(e, e->cand, c, sizeof(Candidate)); }

static int begin_step(lpamd_engine* e) {
  LP_TRY(pull_state(e));
  e->h_st.status = ST_RUNNING;
  e->h_st.max_iter = std::numeric_limits<int64_t>::max();
  e->h_st.n_active = 0;
  e->price_partials_fresh = false;
  return push_state(e);
}

int lpamd_price_max_cost(lpamd_engine* e, int64_t* col) {
  CHECK_LOADED(e);
  LP_TRY(single_gpu_only(e, "lpamd_price_max_cost"));
  LP_TRY(begin_step(e));
  LP_TRY(enqueue_pricing(e, LPAMD_MAX_COST, e->cand));
  Candidate c;
  LP_TRY(read_candidate(e, &c));
  *col = c.gcol;
  return LPAMD_OK;
}

int lpamd_price_max_reduction(lpamd_engine* e, int64_t* col, int64_t* r, int64_t* d) {
  CHECK_LOADED(e);
  LP_TRY(single_gpu_only(e, "lpamd_price_max_reduction"));
  LP_TRY(begin_step(e));
  LP_TRY(enqueue_pricing(e, LPAMD_MAX_REDUCTION, e->cand));
  Candidate c;
  LP_TRY(read_candidate(e, &c));
  *col = c.gcol;
  *r = c.row;
  *d = -1;
  if (c.row >= 0) LP_TRY(read_local(e, e->basis + c.row, d, sizeof(int64_t)));
  return LPAMD_OK;
}

int lpamd_ratio_test(lpamd_engine* e, int64_t col, int64_t* r, int64_t* d) {
  CHECK_LOADED(e);
  LP_TRY(single_gpu_only(e, "lpamd_ratio_test"));
  if (col < 0 || col > e->sh.N + 1 || col == e->sh.N) return set_err(LPAMD_E_ARG, "ratio_test: bad column %lld", (long long)col);
  LP_TRY(begin_step(e));
  Candidate c{0.0, col, -1, 0.0};
  HIP_TRY(hipMemcpyAsync(e->cand, &c, sizeof(c), hipMemcpyHostToDevice, e->stream));
  LP_TRY(enqueue_ratio_partial(e, false, e->cand, nullptr));
  std::vector<double> rv(e->n_ratio_part);
  std::vector<int64_t> ri(e->n_ratio_part);
  LP_TRY(read_local(e, e->part_v + kMaxPartials, rv.data(), sizeof(double) * rv.size()));
  LP_TRY(read_local(e, e->part_i + kMaxPartials, ri.data(), sizeof(int64_t) * ri.size()));
  double bv = 0.0;
  int64_t bi = -1;
  for (size_t k = 0; k < rv.size(); k++)   // same rule as the device fold: smaller ratio, then lower row
    if (ri[k] >= 0 && (bi < 0 || rv[k] < bv || (rv[k] == bv && ri[k] < bi))) { bv = rv[k]; bi = ri[k]; }
  *r = bi;
  *d = -1;
  if (bi >= 0) LP_TRY(read_local(e, e->basis + bi, d, sizeof(int64_t)));
  return LPAMD_OK;
}

int lpamd_pivot(lpamd_engine* e, int64_t d, int64_t col, int64_t r) {
  CHECK_LOADED(e);
  CHECK_NO_PENDING_UPDATE(e);
  LP_TRY(single_gpu_only(e, "lpamd_pivot"));
  const Shape& sh = e->sh;
  if (col < 0 || col > sh.N + 1 || col == sh.N || r < 0 || r >= sh.m || d > sh.N + 1)
    return set_err(LPAMD_E_ARG, "pivot: bad arguments d=%lld e=%lld r=%lld", (long long)d, (long long)col, (long long)r);
  LP_TRY(begin_step(e));
  LP_TRY(ensure_trace(e, e->h_st.iter + 1));
  Candidate c{0.0, col, r, 0.0};
  LP_TRY(read_local(e, e->obj + local_of(e, col), &c.ce, sizeof(double)));
  HIP_TRY(hipMemcpyAsync(e->cand, &c, sizeof(c), hipMemcpyHostToDevice, e->stream));
  LP_TRY(enqueue_pivot(e, false, e->cand, nullptr, d >= 0 ? d : (int64_t)(sh.N)));  // d<0: nothing to clear
  return pull_state(e);
}

int lpamd_pivot_shard(lpamd_engine* e, int64_t d, int64_t col, int64_t r, double ce, const void* xbuf_dev) {
  CHECK_LOADED(e);
  CHECK_NO_PENDING_UPDATE(e);
  const Shape& sh = e->sh;
  if (col < 0 || col > sh.N + 1 || col == sh.N || r < 0 || r >= sh.m || d > sh.N + 1)
    return set_err(LPAMD_E_ARG, "pivot_shard: bad arguments d=%lld e=%lld r=%lld", (long long)d, (long long)col, (long long)r);
  if (!xbuf_dev && !owns(e, col)) return set_err(LPAMD_E_ARG, "pivot_shard: column %lld lives on another rank and no exchange buffer was given", (long long)col);
  LP_TRY(begin_step(e));
  LP_TRY(ensure_trace(e, e->h_st.iter + 1));
  const Candidate c{0.0, col, r, ce};
  HIP_TRY(hipMemcpyAsync(e->cand, &c, sizeof(c), hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));   // c is a stack object
  LP_TRY(enqueue_pivot(e, false, e->cand, (const double*)xbuf_dev, d >= 0 ? d : (int64_t)(sh.N)));
  return pull_state(e);
}

int lpamd_fill_column(lpamd_engine* e, int64_t col, double value) {
  CHECK_LOADED(e);
  CHECK_NO_PENDING_UPDATE(e);
  if (col < 0 || col > e->sh.N + 1) return set_err(LPAMD_E_ARG, "fill_column: bad column %lld", (long long)col);
  e->price_partials_fresh = false;
  LP_TRY(invalidate_pr(e));
  if (!owns(e, col)) return LPAMD_OK;
  e->tab_version++;
  hipLaunchKernelGGL(k_fill, dim3((unsigned)((e->sh.m + 255) / 256)), dim3(256), 0, e->stream, e->T.col(local_of(e, col)), e->sh.m, value);
  return check_launch("k_fill");
}

int lpamd_set_basic_flag(lpamd_engine* e, int64_t col, int32_t flag) {
  CHECK_LOADED(e);
  if (col < 0 || col > e->sh.N + 1 || col == e->sh.N) return set_err(LPAMD_E_ARG, "set_basic_flag: bad column %lld", (long long)col);
  e->price_partials_fresh = false;
  LP_TRY(invalidate_pr(e));
  if (!owns(e, col)) return LPAMD_OK;
  const uint8_t v = flag ? 1 : 0;
  HIP_TRY(hipMemcpyAsync(e->is_basic + local_of(e, col), &v, 1, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return LPAMD_OK;
}

int lpamd_run_pivots(lpamd_engine* e, int32_t strategy, int64_t k, int32_t* result, int64_t* done) {
  CHECK_LOADED(e);
  CHECK_NO_PENDING_UPDATE(e);
  LP_TRY(single_gpu_only(e, "lpamd_run_pivots"));
  if (!result) return set_err(LPAMD_E_ARG, "run_pivots: null result");
  LP_TRY(pull_state(e));
  const int64_t it0 = e->h_st.iter;
  LP_TRY(run_loop(e, strategy, k, result));
  if (done) *done = e->h_st.iter - it0;
  return LPAMD_OK;
}

// LPModel::TableauSimplexInitialize (reference lp_simplex_tableau.cc:64-171).
// The pivots and the inner solve run on the device; the one-off scans around
// them read single rows / columns back.
}  // extern "C"

namespace {
// max_pivots >= 0: at most that many pivots, the explicit ones included; *result = LPAMD_LIMIT when they are used up
// before the inner solve ends (the engine is then in the middle of the phase: auxiliary objective, artificial column).
int phase1_budgeted(lpamd_engine* e, int32_t strategy, int64_t max_pivots, int32_t* result);
}  // namespace

extern "C" {

int lpamd_phase1(lpamd_engine* e, int32_t strategy, int32_t* result) {
  CHECK_LOADED(e);
  CHECK_NO_PENDING_UPDATE(e);
  LP_TRY(single_gpu_only(e, "lpamd_phase1"));
  if (!result) return set_err(LPAMD_E_ARG, "phase1: null result");
  return phase1_budgeted(e, strategy, -1, result);
}

}  // extern "C"

namespace {
int phase1_budgeted(lpamd_engine* e, int32_t strategy, int64_t max_pivots, int32_t* result) {
  LP_TRY(pull_state(e));
  const int64_t iter0 = e->h_st.iter;
  const Shape& sh = e->sh;
  const int64_t m = sh.m, N = sh.N, A = sh.N + 1, W = N + 2;
  int32_t need = 0;
  LP_TRY(lpamd_need_init(e, &need));
  if (!need) { *result = LPAMD_SOLVED; return LPAMD_OK; }
  *result = LPAMD_ERROR;
  std::vector<double> saved(W, 0.0), tmp(W, 0.0), col(m), row(W, 0.0);
  std::vector<uint8_t> basic(W, 0);
  LP_TRY(lpamd_download(e, nullptr, 0, saved.data(), nullptr));
  tmp[A] = -1.0;                                                       // :79
  LP_TRY(lpamd_write_objective(e, tmp.data()));
  const uint8_t zero8 = 0;
  HIP_TRY(hipMemcpyAsync(e->is_basic + local_of(e, A), &zero8, 1, hipMemcpyHostToDevice, e->stream));   // :71
  e->tab_version++;
  hipLaunchKernelGGL(k_fill, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, e->stream, e->T.col(local_of(e, A)), m, 1.0);  // :82-86
  LP_TRY(check_launch("k_fill"));
  LP_TRY(invalidate_pr(e));
  LP_TRY(lpamd_read_column(e, N, col.data()));
  int64_t r = -1;
  double mv = 0.0;
  for (int64_t i = 0; i < m; i++)                                      // MinReduce, ties to the later row (:88-91)
    if (!(mv < col[i])) { mv = col[i]; r = i; }
  if (r < 0) return LPAMD_OK;
  LP_TRY(lpamd_read_row(e, r, row.data()));
  LP_TRY(lpamd_download(e, nullptr, 0, nullptr, basic.data()));
  int64_t d = -1;
  for (int64_t j = 0; j < W; j++) {                                    // :94-104
    if (j == N || h_is_zero(row[j])) continue;
    if (basic[j]) { d = j; break; }
  }
  if (d < 0) return LPAMD_OK;
  if (max_pivots == 0) { *result = LPAMD_LIMIT; return LPAMD_OK; }
  LP_TRY(lpamd_pivot(e, d, A, r));
  LP_TRY(lpamd_read_column(e, N, col.data()));
  for (int64_t i = 0; i < m; i++)                                      // :106-108 (live assert upstream)
    if (!h_is_nonneg(col[i])) return LPAMD_OK;
  int32_t inner = LPAMD_ERROR;
  LP_TRY(pull_state(e));
  const int64_t left = max_pivots < 0 ? -1 : std::max<int64_t>(0, max_pivots - (e->h_st.iter - iter0));
  LP_TRY(run_loop(e, strategy, left, &inner));                         // :110
  if (inner == LPAMD_LIMIT) { *result = LPAMD_LIMIT; return LPAMD_OK; }
  if (inner != LPAMD_SOLVED) return LPAMD_OK;                          // :111 (live assert upstream)
  double oc = 0.0;
  LP_TRY(lpamd_objective_constant(e, &oc));
  if (h_is_neg(oc)) { *result = LPAMD_NOSOLUTION; return LPAMD_OK; }   // :113-115
  LP_TRY(lpamd_download(e, nullptr, 0, nullptr, basic.data()));
  if (basic[A]) {                                                      // :117-135
    LP_TRY(lpamd_read_column(e, A, col.data()));
    bool pivoted = false;
    for (int64_t i = 0; i < m && !pivoted; i++) {
      if (h_is_zero(col[i])) continue;
      LP_TRY(lpamd_read_row(e, i, row.data()));
      for (int64_t j = 0; j < W; j++) {
        if (j != N && basic[j]) continue;
        if (h_is_zero(row[j])) continue;
        if (j == N) return LPAMD_OK;  // upstream would pivot on the constant column and abort on its assert
        LP_TRY(lpamd_pivot(e, A, j, i));
        pivoted = true;
        break;
      }
    }
    if (!pivoted) return LPAMD_OK;
    HIP_TRY(hipMemcpyAsync(e->is_basic + local_of(e, A), &zero8, 1, hipMemcpyHostToDevice, e->stream));
    LP_TRY(lpamd_download(e, nullptr, 0, nullptr, basic.data()));
  }
  e->tab_version++;
  hipLaunchKernelGGL(k_fill, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, e->stream, e->T.col(local_of(e, A)), m, 0.0);  // :143
  LP_TRY(check_launch("k_fill"));
  LP_TRY(invalidate_pr(e));
  saved[A] = 0.0;
  // objective re-expression over the basic columns (:149-166)
  std::vector<double> acc(W, 0.0);
  for (int64_t j = 0; j < W; j++) {
    if (j == N || !basic[j]) continue;
    if (h_is_zero(saved[j])) continue;
    LP_TRY(lpamd_read_column(e, j, col.data()));
    int64_t rho = -1;
    double bv = (double)1e-6f;                                         // MaxAbsReduce init {-1, kEpsilonF}
    for (int64_t i = 0; i < m; i++)
      if (!(std::fabs(bv) > std::fabs(col[i]))) { bv = col[i]; rho = i; }
    if (rho < 0) return LPAMD_OK;
    LP_TRY(lpamd_read_row(e, rho, row.data()));
    const double s = -saved[j] / row[j];                               // :162
    for (int64_t k = 0; k < W; k++) {
      const double prod = row[k] * s;
      acc[k] = h_flushz(acc[k] + prod);
    }
  }
  for (int64_t k = 0; k < W; k++) saved[k] = h_flushz(saved[k] + acc[k]);  // :166
  LP_TRY(lpamd_write_objective(e, saved.data()));
  *result = LPAMD_SOLVED;
  return LPAMD_OK;
}
}  // namespace

extern "C" {

int lpamd_solve(lpamd_engine* e, int32_t strategy, int64_t max_pivots, int32_t* result, int64_t* n_pivots) {
  CHECK_LOADED(e);
  CHECK_NO_PENDING_UPDATE(e);
  LP_TRY(single_gpu_only(e, "lpamd_solve"));
  if (!result) return set_err(LPAMD_E_ARG, "solve: null result");
  if (strategy != LPAMD_MAX_COST && strategy != LPAMD_MAX_REDUCTION)
    return set_err(LPAMD_E_UNSUPPORTED, "pivoting strategy %d is not implemented for the tableau path", strategy);
  int32_t need = 0;
  LP_TRY(lpamd_need_init(e, &need));
  int64_t left = max_pivots;          // the budget covers the feasibility phase as well
  if (need) {                                                          // reference lp_simplex_tableau.cc:174-178
    LP_TRY(pull_state(e));
    const int64_t iter0 = e->h_st.iter;
    int32_t r1 = LPAMD_ERROR;
    LP_TRY(phase1_budgeted(e, strategy, max_pivots, &r1));
    if (r1 != LPAMD_SOLVED) {
      *result = (r1 == LPAMD_NOSOLUTION || r1 == LPAMD_LIMIT) ? r1 : LPAMD_ERROR;
      LP_TRY(pull_state(e));
      if (n_pivots) *n_pivots = e->h_st.iter;
      return LPAMD_OK;
    }
    LP_TRY(lpamd_need_init(e, &need));
    if (need) { *result = LPAMD_ERROR; return LPAMD_OK; }              // :179
    LP_TRY(pull_state(e));
    if (max_pivots >= 0) left = std::max<int64_t>(0, max_pivots - (e->h_st.iter - iter0));
  }
  LP_TRY(run_loop(e, strategy, left, result));
  if (n_pivots) *n_pivots = e->h_st.iter;
  return LPAMD_OK;
}

int lpamd_column_values(lpamd_engine* e, double* x) {
  CHECK_LOADED(e);
  if (!x) return set_err(LPAMD_E_ARG, "column_values: null");
  hipLaunchKernelGGL(k_column_values, dim3(e->sh.nloc), dim3(256), 0, e->stream, e->T, e->is_basic, e->sh, e->rowp);
  LP_TRY(check_launch("k_column_values"));
  std::vector<double> h(e->sh.nloc);
  LP_TRY(read_local(e, e->rowp, h.data(), sizeof(double) * e->sh.nloc));
  for (int64_t j = 0; j < e->sh.N + 2; j++) x[j] = 0.0;
  for (int64_t l = 0; l < e->sh.nloc; l++) x[global_of(e, l)] = h[l];
  return LPAMD_OK;
}

int lpamd_objective_constant(lpamd_engine* e, double* out) {
  CHECK_LOADED(e);
  if (!out) return set_err(LPAMD_E_ARG, "objective_constant: null");
  return read_local(e, e->obj + e->sh.nloc, out, sizeof(double));
}

int lpamd_pivot_count(const lpamd_engine* e, int64_t* n) {
  CHECK_LOADED(e);
  LP_TRY(pull_state(const_cast<lpamd_engine*>(e)));
  *n = e->h_st.iter;
  return LPAMD_OK;
}

int lpamd_trace(const lpamd_engine* e, int64_t* te, int64_t* tr, int64_t cap, int64_t* n) {
  CHECK_LOADED(e);
  LP_TRY(pull_state(const_cast<lpamd_engine*>(e)));
  const int64_t k = std::min<int64_t>(std::min<int64_t>(e->h_st.iter, e->trace_cap), cap);
  std::vector<int64_t> h(2 * std::max<int64_t>(k, 1));
  if (k > 0) LP_TRY(read_local(e, e->trace, h.data(), sizeof(int64_t) * 2 * k));
  for (int64_t i = 0; i < k; i++) { te[i] = h[2 * i]; tr[i] = h[2 * i + 1]; }
  if (n) *n = e->h_st.iter;
  return LPAMD_OK;
}

int lpamd_set_profiling(lpamd_engine* e, int32_t on) {
  CHECK_ENGINE(e);
  HIP_TRY(hipStreamSynchronize(e->stream));
  e->profiling = on != 0;
  e->ev_used = 0;
  e->prof_ms = 0.0;
  e->prof_launches = 0;
  e->prof_active = 0;
  HIP_TRY(hipMemsetAsync(e->steps, 0, sizeof(ChainStep), e->stream));   // running totals of the chained passes
  HIP_TRY(hipStreamSynchronize(e->stream));
  return LPAMD_OK;
}

// Folds the HIP events recorded since the last call into the running totals.
static int drain_events(lpamd_engine* e) {
  HIP_TRY(hipStreamSynchronize(e->stream));
  for (size_t k = 0; k + 1 < e->ev_used; k += 2) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e->ev[k], e->ev[k + 1]));
    e->prof_ms += ms;
    e->prof_launches++;
  }
  e->ev_used = 0;
  return LPAMD_OK;
}

int lpamd_update_stats(lpamd_engine* e, double* total_ms, int64_t* launches, int64_t* active_columns) {
  CHECK_ENGINE(e);
  LP_TRY(drain_events(e));
  LP_TRY(pull_state(e));
  e->prof_active = e->h_st.n_active;
  if (e->loop_depth > 1) {   // a pass that applied several pivots counts the columns of each
    PassInfo pi;
    ChainStep c0;
    LP_TRY(read_local(e, e->pass, &pi, sizeof(pi)));
    LP_TRY(read_local(e, e->steps, &c0, sizeof(c0)));
    if (pi.n_steps > 0) e->prof_active = pi.touched;
    // passes of different lengths (a budget that ends inside a chain): the average per pass since profiling began
    if (c0.r > 0) e->prof_active = c0.e / c0.r;
  }
  if (total_ms) *total_ms = e->prof_ms;
  if (launches) *launches = e->prof_launches;
  if (active_columns) *active_columns = e->prof_active;
  return LPAMD_OK;
}

int lpamd_profile_totals(lpamd_engine* e, double* total_ms, int64_t* passes, int64_t* pivots, int64_t* streamed_columns,
                         int64_t* touched_columns) {
  CHECK_ENGINE(e);
  LP_TRY(drain_events(e));
  LP_TRY(pull_state(e));
  int64_t np = e->prof_launches, nv = e->prof_launches;
  int64_t streamed = e->prof_launches * (int64_t)e->h_st.n_active, touched = streamed;   // one pivot per pass
  if (e->loop_depth > 1) {   // chained passes keep their totals on the device (record 0 of the step array)
    ChainStep c0;
    LP_TRY(read_local(e, e->steps, &c0, sizeof(c0)));
    np = c0.r; nv = (int64_t)c0.a; streamed = c0.d; touched = c0.e;
  }
  if (total_ms) *total_ms = e->prof_ms;
  if (passes) *passes = np;
  if (pivots) *pivots = nv;
  if (streamed_columns) *streamed_columns = streamed;
  if (touched_columns) *touched_columns = touched;
  return LPAMD_OK;
}

int lpamd_pricing_stats(lpamd_engine* e, int64_t* pricings, int64_t* columns_evaluated) {
  CHECK_ENGINE(e);
  PrbInfo pi;
  std::memset(&pi, 0, sizeof(pi));
  if (e->prb_info) LP_TRY(read_local(e, e->prb_info, &pi, sizeof(pi)));
  if (pricings) *pricings = pi.priced;
  if (columns_evaluated) *columns_evaluated = pi.priced + pi.listed;   // the column with the largest bound is always evaluated
  return LPAMD_OK;
}

int lpamd_loop_depth(const lpamd_engine* e, int32_t strategy, int32_t* depth) {
  if (!e || !depth) return set_err(LPAMD_E_ARG, "loop_depth: null argument");
  *depth = (strategy == LPAMD_MAX_COST && !e->two_launch) ? e->chain_depth : (strategy == LPAMD_MAX_REDUCTION && e->pr_bounds) ? e->mr_depth : 1;
  return LPAMD_OK;
}

int lpamd_pass_stats(lpamd_engine* e, int32_t* pivots, int64_t* streamed_columns, int64_t* touched_columns) {
  CHECK_ENGINE(e);
  LP_TRY(pull_state(e));
  int32_t np = 1;
  int64_t streamed = e->h_st.n_active, touched = e->h_st.n_active;
  if (e->loop_depth > 1) {
    PassInfo pi;
    LP_TRY(read_local(e, e->pass, &pi, sizeof(pi)));
    if (pi.n_steps > 0) { np = pi.n_steps; streamed = pi.n_full + pi.n_part; touched = pi.touched; }
  }
  if (pivots) *pivots = np;
  if (streamed_columns) *streamed_columns = streamed;
  if (touched_columns) *touched_columns = touched;
  return LPAMD_OK;
}

// ------------------------------------------------------------- multi-GPU steps
int lpamd_local_candidate(lpamd_engine* e, int32_t strategy, void* cand_dev) {
  CHECK_LOADED(e);
  if (!cand_dev) return set_err(LPAMD_E_ARG, "local_candidate: null buffer");
  return enqueue_pricing(e, strategy, (Candidate*)cand_dev);
}

// Pure host fold over the gathered records (no device access, callable without a
// GPU): MAX_COST keeps the largest reduced cost (reference
// lp_simplex_tableau.cc:190-198), MAX_REDUCTION the largest c_j * min-ratio among
// columns that have an eligible row (reference lp_simplex_tableau.cc:310-325 /
// 353-362); ties go to the lowest global column id in both.
int lpamd_fold_candidates(int32_t strategy, const void* records_host, int32_t world, int64_t N, int64_t* col, int64_t* r,
                          int32_t* owner_rank, int32_t* result, void* winner_out) {
  if (!records_host || world <= 0 || !col || !r || !owner_rank || !result)
    return set_err(LPAMD_E_ARG, "fold_candidates: bad arguments");
  if (strategy != LPAMD_MAX_COST && strategy != LPAMD_MAX_REDUCTION)
    return set_err(LPAMD_E_UNSUPPORTED, "pivoting strategy %d is not implemented for the tableau path", strategy);
  const Candidate* c = (const Candidate*)records_host;
  *col = -1;
  *r = -1;
  *owner_rank = -1;
  Candidate best{-DBL_MAX, -1, -1, 0.0};
  bool any = false;
  for (int32_t k = 0; k < world; k++) {
    if (c[k].gcol < 0) continue;
    any = true;
    if (strategy == LPAMD_MAX_REDUCTION && c[k].row < 0) continue;
    if (best.gcol < 0 || c[k].key > best.key || (c[k].key == best.key && c[k].gcol < best.gcol)) best = c[k];
  }
  if (best.gcol < 0) {
    *result = any ? LPAMD_UNBOUNDED : LPAMD_SOLVED;   // reference lp_simplex_tableau.cc:365-373 / 201-205
    return LPAMD_OK;
  }
  if (winner_out) std::memcpy(winner_out, &best, sizeof(Candidate));
  *col = best.gcol;
  *r = best.row;
  *owner_rank = (int32_t)(ordinal_of(best.gcol, N) % world);
  *result = LPAMD_LIMIT;
  return LPAMD_OK;
}

int lpamd_select_candidate(lpamd_engine* e, int32_t strategy, const void* cands_dev, int32_t world, int64_t* col, int64_t* r,
                           int32_t* owner_rank, int32_t* result) {
  CHECK_LOADED(e);
  if (!cands_dev || world != e->sh.world || !col || !r || !owner_rank || !result)
    return set_err(LPAMD_E_ARG, "select_candidate: bad arguments");
  std::vector<Candidate> c(world);
  LP_TRY(read_local(e, cands_dev, c.data(), sizeof(Candidate) * world));
  LP_TRY(pull_state(e));
  *col = -1;
  *r = -1;
  *owner_rank = -1;
  if (e->h_st.status != ST_RUNNING) { *result = status_to_result(e->h_st.status); return LPAMD_OK; }
  Candidate best;
  LP_TRY(lpamd_fold_candidates(strategy, c.data(), world, e->sh.N, col, r, owner_rank, result, &best));
  if (*result != LPAMD_LIMIT) return LPAMD_OK;
  HIP_TRY(hipMemcpyAsync(e->cand, &best, sizeof(Candidate), hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));   // `best` is a stack object
  return LPAMD_OK;
}

int lpamd_select_candidate_device(lpamd_engine* e, int32_t strategy, const void* cands_dev, int32_t world) {
  CHECK_LOADED(e);
  if (!cands_dev || world != e->sh.world) return set_err(LPAMD_E_ARG, "select_candidate_device: bad arguments");
  if (strategy != LPAMD_MAX_COST && strategy != LPAMD_MAX_REDUCTION)
    return set_err(LPAMD_E_UNSUPPORTED, "pivoting strategy %d is not implemented for the tableau path", strategy);
  hipLaunchKernelGGL(k_fold_candidates, dim3(1), dim3(kWave), 0, e->stream, (const Candidate*)cands_dev, (int)world,
                     strategy == LPAMD_MAX_REDUCTION ? 1 : 0, e->st, e->cand);
  return check_launch("k_fold_candidates");
}

int lpamd_export_column_or_zero(lpamd_engine* e, void* xbuf_dev) {
  CHECK_LOADED(e);
  if (!xbuf_dev) return set_err(LPAMD_E_ARG, "export_column_or_zero: null exchange buffer");
  hipLaunchKernelGGL(k_export_or_zero, dim3((unsigned)((e->sh.lda + 255) / 256)), dim3(256), 0, e->stream, e->T, e->sh, e->st, e->cand,
                     (double*)xbuf_dev);
  return check_launch("k_export_or_zero");
}

int lpamd_set_pivot_budget(lpamd_engine* e, int64_t max_pivots) {
  CHECK_LOADED(e);
  LP_TRY(pull_state(e));
  e->h_st.status = ST_RUNNING;
  e->h_st.max_iter = max_pivots < 0 ? std::numeric_limits<int64_t>::max() : e->h_st.iter + max_pivots;
  e->price_partials_fresh = false;
  return push_state(e);
}

int lpamd_export_column(lpamd_engine* e, int64_t col, void* xbuf_dev) {
  CHECK_LOADED(e);
  if (!xbuf_dev || col < 0 || col > e->sh.N + 1 || col == e->sh.N) return set_err(LPAMD_E_ARG, "export_column: bad arguments");
  if (!owns(e, col)) return set_err(LPAMD_E_ARG, "export_column: column %lld lives on another rank", (long long)col);
  HIP_TRY(hipMemcpyAsync(xbuf_dev, e->T.col(local_of(e, col)), sizeof(double) * e->sh.lda, hipMemcpyDeviceToDevice, e->stream));
  return LPAMD_OK;
}

int lpamd_apply_pivot(lpamd_engine* e, int32_t strategy, int64_t col, int64_t r, const void* xbuf_dev) {
  CHECK_LOADED(e);
  CHECK_NO_PENDING_UPDATE(e);
  (void)col;
  (void)r;
  if (!xbuf_dev) return set_err(LPAMD_E_ARG, "apply_pivot: null exchange buffer");
  // h_st.iter is refreshed by lpamd_select_candidate / lpamd_poll; between two polls the caller
  // may enqueue many pivots, so keep generous head-room (the trace stops recording when full).
  e->apply_since_pull++;
  LP_TRY(ensure_trace(e, e->h_st.iter + e->apply_since_pull + 2));
  return enqueue_pivot(e, strategy == LPAMD_MAX_COST, e->cand, (const double*)xbuf_dev, -1, strategy == LPAMD_MAX_REDUCTION);
}

int lpamd_lookahead_export(lpamd_engine* e, void* cand_dev, void* xlocal_dev) {
  CHECK_LOADED(e);
  if (!cand_dev || !xlocal_dev) return set_err(LPAMD_E_ARG, "lookahead_export: null buffer");
  if (!e->price_partials_fresh) LP_TRY(enqueue_price_partial(e));
  hipLaunchKernelGGL(k_export_lookahead<false>, dim3((unsigned)((e->sh.lda + kPartialThreads - 1) / kPartialThreads)), dim3(kPartialThreads), 0,
                     e->stream, e->T, e->sh, e->st, e->part_v, e->part_i, e->n_price_part, (Candidate*)cand_dev, e->colp, e->rowp,
                     e->rstride, e->steps, e->update_pending ? e->pending_steps : 0, (double*)xlocal_dev, nullptr, nullptr, nullptr);
  return check_launch("k_export_lookahead");
}

int lpamd_lookahead_select(lpamd_engine* e, const void* cands_dev, int32_t world, const void* xlocal_dev, void* xbuf_dev,
                           void* hip_stream) {
  CHECK_LOADED(e);
  if (!cands_dev || !xlocal_dev || !xbuf_dev || world != e->sh.world) return set_err(LPAMD_E_ARG, "lookahead_select: bad arguments");
  hipStream_t s = hip_stream ? (hipStream_t)hip_stream : e->stream;
  hipLaunchKernelGGL(k_lookahead_select, dim3((unsigned)((e->sh.lda + 255) / 256)), dim3(256), 0, s, (const Candidate*)cands_dev,
                     (int)world, e->sh, (const double*)xlocal_dev, e->cand_la, (double*)xbuf_dev);
  return check_launch("k_lookahead_select");
}

int lpamd_lookahead_setup(lpamd_engine* e, const void* xbuf_dev) {
  CHECK_LOADED(e);
  if (!xbuf_dev) return set_err(LPAMD_E_ARG, "lookahead_setup: null exchange buffer");
  if (e->update_pending) return set_err(LPAMD_E_STATE, "lookahead_setup: the previous pivot still waits for lpamd_lookahead_update");
  e->apply_since_pull++;
  LP_TRY(ensure_trace(e, e->h_st.iter + e->apply_since_pull + 2));
  LP_TRY(enqueue_ratio_partial(e, false, e->cand_la, (const double*)xbuf_dev));
  LP_TRY(enqueue_finalize(e, true, e->cand_la, (const double*)xbuf_dev, -1));
  e->update_pending = true;
  e->pending_steps = 1;
  return LPAMD_OK;
}

int lpamd_chain_depth(const lpamd_engine* e, int32_t* depth) {
  if (!e || !depth) return set_err(LPAMD_E_ARG, "chain_depth: null argument");
  *depth = e->chain_depth;
  return LPAMD_OK;
}

int lpamd_chain_step(lpamd_engine* e, void* xbuf_dev) {
  CHECK_LOADED(e);
  if (!xbuf_dev) return set_err(LPAMD_E_ARG, "chain_step: null exchange buffer");
  if (!e->update_pending) return set_err(LPAMD_E_STATE, "chain_step: no pivot has been set up by lpamd_lookahead_setup");
  if (e->dense_first) return set_err(LPAMD_E_STATE, "chain_step: the first pivot after this upload must be applied alone (entries below the drop threshold)");
  if (e->pending_steps >= e->chain_depth)
    return set_err(LPAMD_E_STATE, "chain_step: the pass already holds %d pivots", e->pending_steps);
  e->apply_since_pull++;
  LP_TRY(ensure_trace(e, e->h_st.iter + e->apply_since_pull + 2));
  LP_TRY(enqueue_chain_step(e, e->pending_steps, e->cand_la, (double*)xbuf_dev));
  e->pending_steps++;
  return LPAMD_OK;
}

}  // extern "C"

namespace {

// Ratio-test partials a slot carries behind the column: one (value, row) pair per 256 rows.
int64_t slot_ratio_parts(const lpamd_engine* e) { return (e->sh.lda + kPartialThreads - 1) / kPartialThreads; }

// Column shards, MAX_COST, a step inside a chain: this rank's candidate for the NEXT pivot packed into `slot` (record,
// column as the pivots set up so far will leave it, ratio-test partials on that column).
int export_lookahead_ratio(lpamd_engine* e, uint64_t* slot, int header_words) {
  if (!e->price_partials_fresh) LP_TRY(enqueue_price_partial(e));
  const int64_t n_parts = slot_ratio_parts(e);
  double* col = reinterpret_cast<double*>(slot + header_words);
  double* rpv = col + e->sh.lda;
  hipLaunchKernelGGL(k_export_lookahead<true>, dim3((unsigned)n_parts), dim3(kPartialThreads), 0, e->stream, e->T, e->sh, e->st, e->part_v, e->part_i,
                     e->n_price_part, reinterpret_cast<Candidate*>(slot), e->colp, e->rowp, e->rstride, e->steps,
                     e->update_pending ? e->pending_steps : 0, col, e->rhs_next, rpv, reinterpret_cast<int64_t*>(rpv + n_parts));
  return check_launch("k_export_lookahead");
}

// export_lookahead_ratio with the slot stored into every rank's receive buffer (link.tag_out) instead of a send buffer.
int export_lookahead_ratio_link(lpamd_engine* e, int world, int header_words, const ShardLink& link) {
  if (!link.tag_out || !link.peers || world > kWave) return set_err(LPAMD_E_ARG, "export: bad link");
  if (!e->price_partials_fresh) LP_TRY(enqueue_price_partial(e));
  hipLaunchKernelGGL(k_export_lookahead_link, dim3((unsigned)slot_ratio_parts(e)), dim3(kPartialThreads), 0, e->stream, e->T, e->sh, e->st, e->part_v,
                     e->part_i, e->n_price_part, e->colp, e->rowp, e->rstride, e->steps, e->update_pending ? e->pending_steps : 0, e->rhs_next, world, header_words, link);
  return check_launch("k_export_lookahead_link");
}

// The two together: set-up of step `pending_steps` from the gathered slots and, into `next_slot` (nullptr: none), this rank's
// slot for the step after it (k_shard_step).  link (nullptr: none): the exchange inside the launch; tag_in != 0: the slots
// are read from this rank's receive buffer, tag_out != 0: the slot goes straight into every rank's receive buffer.
int chain_step_export_from_slots(lpamd_engine* e, const uint64_t* slots, int world, int64_t slot_words, int header_words, uint64_t* next_slot,
                                 const ShardLink* link = nullptr) {
  if (!e->update_pending) return set_err(LPAMD_E_STATE, "chain step: no pivot has been set up by lpamd_lookahead_setup");
  if (e->pending_steps >= e->chain_depth) return set_err(LPAMD_E_STATE, "chain step: the pass already holds %d pivots", e->pending_steps);
  ShardLink lk{};
  if (link) lk = *link;
  const bool in = lk.tag_in != 0, out = lk.tag_out != 0;
  if ((in || out) && (world > kWave || !lk.peers || !lk.in)) return set_err(LPAMD_E_ARG, "chain step: bad link");
  if (out && !next_slot) return set_err(LPAMD_E_ARG, "chain step: a slot is exported only when a next step follows");
  e->apply_since_pull++;
  LP_TRY(ensure_trace(e, e->h_st.iter + e->apply_since_pull + 2));
  e->shard_epoch = (e->shard_epoch + 1) & 0xFFFFFFu;
  if (e->shard_epoch == 0) {   // the launch number wrapped: no record of an old launch may carry a tag that comes round again
    HIP_TRY(hipMemsetAsync(e->rec_shard, 0, sizeof(uint64_t) * (size_t)kFusedReplicas * fused_region_words(e->n_price_part, kShardRecWords), e->stream));
    e->shard_epoch = 1;
  }
  auto kernel = in ? (out ? k_shard_step<true, true> : k_shard_step<true, false>) : (out ? k_shard_step<false, true> : k_shard_step<false, false>);
  hipLaunchKernelGGL(kernel, dim3(e->n_row_blocks + e->n_price_part), dim3(kPartialThreads), 0, e->stream, e->T, e->sh, e->st, e->steps,
                     e->pending_steps, (int)slot_ratio_parts(e), e->n_row_blocks, e->colp, e->rowp, e->rstride, e->rhs_next, e->obj, e->is_basic,
                     e->basis, e->trace, e->trace_cap, e->part_v, e->part_i, slots, world, slot_words, header_words, e->chain_depth, next_slot,
                     e->rec_shard, e->shard_epoch, lk);
  LP_TRY(check_launch("k_shard_step"));
  e->price_partials_fresh = true;
  e->pending_steps++;
  return LPAMD_OK;
}

// ... and the set-up of that pivot on every rank straight from the gathered slots (no fold kernel, no ratio-test launch).
// rows_given: MAX_REDUCTION, the records carry the pivot rows and no ratio-test partials.
int chain_step_from_slots(lpamd_engine* e, const uint64_t* slots, int world, int64_t slot_words, int header_words, bool rows_given = false) {
  if (!e->update_pending) return set_err(LPAMD_E_STATE, "chain step: no pivot has been set up by lpamd_lookahead_setup");
  if (e->pending_steps >= e->chain_depth) return set_err(LPAMD_E_STATE, "chain step: the pass already holds %d pivots", e->pending_steps);
  e->apply_since_pull++;
  LP_TRY(ensure_trace(e, e->h_st.iter + e->apply_since_pull + 2));
  if (rows_given)
    hipLaunchKernelGGL((k_chain_finalize<false, false, true>), dim3(e->n_row_blocks + e->n_price_part), dim3(kPartialThreads), 0, e->stream, e->T, e->sh,
                       e->st, e->steps, e->pending_steps, nullptr, nullptr, 0, e->n_row_blocks, e->colp, e->rowp, e->rstride,
                       e->rhs_next, nullptr, nullptr, e->obj, e->is_basic, e->basis, e->trace, e->trace_cap, e->part_v, e->part_i, slots, world,
                       slot_words, header_words, e->chain_depth);
  else
    hipLaunchKernelGGL((k_chain_finalize<true, false, true>), dim3(e->n_row_blocks + e->n_price_part), dim3(kPartialThreads), 0, e->stream, e->T, e->sh,
                       e->st, e->steps, e->pending_steps, nullptr, nullptr, (int)slot_ratio_parts(e), e->n_row_blocks, e->colp, e->rowp, e->rstride,
                       e->rhs_next, nullptr, nullptr, e->obj, e->is_basic, e->basis, e->trace, e->trace_cap, e->part_v, e->part_i, slots, world,
                       slot_words, header_words, e->chain_depth);
  LP_TRY(check_launch("k_chain_finalize"));
  e->price_partials_fresh = true;
  e->pending_steps++;
  return LPAMD_OK;
}

}  // namespace

extern "C" {

int lpamd_lookahead_update(lpamd_engine* e) {
  CHECK_LOADED(e);
  if (!e->update_pending) return set_err(LPAMD_E_STATE, "lookahead_update: no pivot has been set up");
  e->update_pending = false;
  const int steps = e->pending_steps;
  e->pending_steps = 0;
  if (steps > 1) {
    e->loop_depth = steps;
    return enqueue_chain_update(e);
  }
  return enqueue_update(e, false);
}

int lpamd_poll(lpamd_engine* e, int32_t* result) {
  CHECK_LOADED(e);
  LP_TRY(pull_state(e));
  *result = status_to_result(e->h_st.status);
  return LPAMD_OK;
}

}  // extern "C"

#include "lpamd_revised.hip.h"
#include "lpamd_noalloc.hip.h"
#include "lpamd_sharded.hip.h"

#ifdef LPAMD_DBG_CLOCK
// experiments only: the checkpoint clocks of the last pass (kMaxChain x 4 x 8 values of the 100 MHz wall clock)
extern "C" int lpamd_dbg_clock(unsigned long long* out) {
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(lpamd::g_dbg_clock), sizeof(unsigned long long) * lpamd::kMaxChain * 4 * 8));
  return LPAMD_OK;
}
#endif

#ifdef LPAMD_DEVICE_CHECKS
// experiments only: how often each index check of the kernels failed since the last call (32 counters, cleared by the call)
extern "C" int lpamd_dbg_check_failures(unsigned int* out) {
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(lpamd::g_check_fail), sizeof(unsigned int) * 32));
  unsigned int zero[32] = {0};
  HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(lpamd::g_check_fail), zero, sizeof(zero)));
  return LPAMD_OK;
}
#endif
