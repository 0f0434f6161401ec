// Device kernels of the dense simplex-tableau engine (gfx950 / CDNA4, wave64).
//
// All arithmetic that reaches the tableau is written with __dmul_rn/__dadd_rn
// so that no multiply-add is contracted into an FMA: the reference build has
// no FMA (x86-64 baseline) and its pivot sequence depends on every product and
// sum being rounded separately.  The file is also compiled with
// -ffp-contract=off.
//
// Data layout (per engine / GPU):
//   T        column-major fp64, leading dimension lda (multiple of 512 rows,
//            padding rows are zero and stay zero), ncl = nloc + 1 columns, kept as
//            two buffers (even / odd local columns, struct Tab) so that the halves
//            can sit in different HBM placement classes (see lpamd_engine.hip):
//            local column l < nloc holds the variable column whose ordinal is
//            l * world + rank; local column nloc is the replica of the
//            right-hand-side column.
//   obj      ncl doubles, the objective row laid out like the columns.
//   is_basic ncl bytes.
//   basis    m int64: global column id of the basic variable of every row.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>

namespace lpamd {

constexpr int kSlabRows = 512;        // rows covered by one 256-thread block with 16 B per lane
constexpr int kUpdateThreads = 256;
constexpr int kSelectThreads = 1024;  // single-workgroup selection kernels
constexpr int kPartialThreads = 256;  // multi-workgroup partial reductions
constexpr int kMaxPartials = 2048;
constexpr int kWave = 64;

// device-side status: 0 while running, otherwise (reference Result value + 1)
// ST_DIRTY: the device loops stop because basic columns are no longer unit columns (see PivotState::dirty); the host goes on
// with the loop that scans rows and columns the way the reference does.
enum : int32_t { ST_RUNNING = 0, ST_ERROR = 1, ST_UNBOUNDED = 2, ST_NOSOLUTION = 3, ST_SOLVED = 4, ST_LIMIT = 5, ST_DIRTY = 6 };

struct Candidate {   // 32 bytes, also the multi-GPU exchange record
  double key;        // MAX_COST: reduced cost c_e.  MAX_REDUCTION: c_e * min ratio
  int64_t gcol;      // global column id, -1 = no candidate
  int64_t row;       // MAX_REDUCTION: pivot row (or -1 = column has no eligible row)
  double ce;         // objective coefficient of gcol
};

struct PivotState {  // device resident, one per engine
  int32_t status;
  // 1: the tableau may hold stored entries below the drop threshold (an uploaded right-hand side of 5e-7, say).  The
  // reference's update adds to EVERY entry, T = flush(T + col' * row'), so its first pivot drops them all, also in
  // columns whose row' entry is 0; from then on every stored entry is a fixed point of the drop and leaving those
  // columns alone changes no bit.  While this flag is set the next pivot is applied the reference's way, to every
  // column (set by the upload when it sees such an entry, cleared by the update kernel of that pivot).
  int32_t dense_next;
  int64_t iter;      // pivots applied so far (index into the trace)
  int64_t max_iter;  // pricing reports ST_LIMIT once iter reaches this
  int64_t e, r, d;   // global ids of entering / leaving column and the pivot row
  double a, ce, f, g;
  int32_t n_active;  // local columns whose pivot-row entry is non-zero
  int32_t n_active2; // second counter of the two-launch loop (the counters alternate from pivot to pivot)
  int32_t l_e, l_d;  // local columns of e and d on this rank (-1: another rank's, or none)
  // The device loops take the leaving variable of a pivot row from basis[] and test no row for "has a basic column with a
  // non-zero entry" (reference lp_simplex_tableau.cc:215-226, 240-251), which is the reference's answer exactly as long as
  // every basic column is a unit column.  The entering column of a pivot becomes one because its other entries cancel,
  // T[i][e] + T[i][e] * (a * (-1/a)): exactly when a * (-1/a) == -1, else up to |T[i][e]| * 2.3e-16, which the zero drop
  // removes unless |T[i][e]| reaches about 4e9 (netlib's 25fv47 gets there: 9.5e10 at pivot 9724, and from then on the
  // reference's scans and basis[] disagree).  Every set-up looks at the col' it writes: an entry of kDirtyBound or more
  // marks the step (dirty_step = step of the pass + 1); no later step of that pass is set up, the pass kernel turns the
  // mark into `dirty`, and the first kernel of the next iteration ends the device loop with ST_DIRTY.  The host then goes
  // on with the loop that scans like the reference (k_basic_scan_rows, k_basis_discovery) and sets dirty = 2: known.
  int32_t dirty;
  int32_t dirty_step;
};
constexpr double kDirtyBound = 4.0e9;

constexpr int kMaxChain = 32;   // pivots one pass over the tableau may apply

struct ChainStep {   // pivots 1.. of the pass (pivot 0 is described by PivotState)
  int32_t done;      // 1 once k_chain_finalize has set the step up
  int32_t slot;      // column shards: rank whose slot delivered the entering column
  int32_t l_in;      // local column of the entering variable (-1: another rank's)
  int32_t l_out;     // local column of the leaving variable (-1: another rank's, or none)
  int64_t e, r, d;
  double a, ce, f, g;
};

// What k_chain_list leaves for the pass kernel.  Zeroed by the set-up of pivot 0 of every pass.
struct PassInfo {
  int32_t n_steps;   // pivots the pass applies
  int32_t n_full;    // columns every step touches (row'_s != 0 for all s): list_full, no tests inside the pass
  int32_t n_part;    // columns some steps leave alone: list_part
  int32_t pad;
  int64_t touched;   // sum over the steps of the columns each touches (the algorithmic count)
};

// The tableau shard: local column l lives in half[l & 1] at column slot l >> 1.
struct Tab {
  double* half[2];
  int64_t lda;
  __host__ __device__ __forceinline__ double* col(int64_t l) const { return half[l & 1] + (l >> 1) * lda; }
};

struct Shape {
  int64_t m, lda, N;
  int32_t nloc;      // owned variable columns (RHS replica is local column nloc)
  int32_t rank, world;
};

// Timeline of the set-up kernels of a chain (experiments only, -DLPAMD_DBG_CLOCK): the 100 MHz wall clock at
// checkpoints of step j, kept per step for the last pass; who = 0 k_chain_ratio, 1 k_chain_finalize row block 0,
// 2 k_chain_finalize column block 0, 3 the last workgroup of k_chain_finalize.  Read back with lpamd_dbg_clock.
#ifdef LPAMD_DBG_CLOCK
__device__ unsigned long long g_dbg_clock[kMaxChain][4][8];
#define LPAMD_CLK(j, who, cp, cond) do { if ((cond) && threadIdx.x == 0) g_dbg_clock[j][who][cp] = wall_clock64(); } while (0)
// the same with the clocks held in registers until the end of the kernel (a store between the loads of a kernel is
// one more thing their waits wait for)
#define LPAMD_CLKR_DECL unsigned long long clk_r[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define LPAMD_CLKR(cp, cond) do { if (cond) clk_r[cp] = wall_clock64(); } while (0)
#define LPAMD_CLKR_FLUSH(j, who, cond) do { if ((cond) && threadIdx.x == 0) for (int q_ = 0; q_ < 8; q_++) g_dbg_clock[j][who][q_] = clk_r[q_]; } while (0)
#else
#define LPAMD_CLK(j, who, cp, cond) do { } while (0)
#define LPAMD_CLKR_DECL do { } while (0)
#define LPAMD_CLKR(cp, cond) do { } while (0)
#define LPAMD_CLKR_FLUSH(j, who, cond) do { } while (0)
#endif

// Index checks of the kernels (experiments only, -DLPAMD_DEVICE_CHECKS; `make dbg-checks`): a violated bound is counted
// per check id instead of trapping, read back with lpamd_dbg_check_failures.  Off in the product build.
#ifdef LPAMD_DEVICE_CHECKS
__device__ unsigned int g_check_fail[32];
#define LPAMD_CHECK(id, cond) do { if (!(cond)) atomicAdd(&g_check_fail[id], 1u); } while (0)
#else
#define LPAMD_CHECK(id, cond) do { } while (0)
#endif

// reference base.h:61-66: the predicates take float, so the double is narrowed first.
__device__ __forceinline__ bool is_pos(double x) { return (float)x >= 1e-6f; }
__device__ __forceinline__ bool is_neg(double x) { return (float)x <= -1e-6f; }
__device__ __forceinline__ bool is_zero(double x) { return fabsf((float)x) < 1e-6f; }
// reference lp_simplex_tableau.cc:5-8: entries below 1e-6f (as a double) are dropped after every add.
__device__ __forceinline__ double flushz(double v) { return fabs(v) < (double)1e-6f ? 0.0 : v; }

__host__ __device__ __forceinline__ int64_t ordinal_of(int64_t gcol, int64_t N) { return gcol < N ? gcol : N; }
__host__ __device__ __forceinline__ int64_t gcol_of(int64_t ord, int64_t N) { return ord < N ? ord : N + 1; }
// Owner rank and local column of ordinal o under the round-robin deal.  Ordinals fit 31 bits (checked at create), so
// the division is a 32-bit one (a 64-bit division is some 150 instructions on this machine), and none on one GPU.
__device__ __forceinline__ int ord_rank(int64_t o, int world) { return world == 1 ? 0 : (int)((uint32_t)o % (uint32_t)world); }
__device__ __forceinline__ int ord_local(int64_t o, int world) { return world == 1 ? (int)o : (int)((uint32_t)o / (uint32_t)world); }

// ---------------------------------------------------------------- reductions
// "larger key wins, lower index breaks ties"; index -1 = empty.
__device__ __forceinline__ void take_max(double& bv, int64_t& bi, double v, int64_t i) {
  if (i >= 0 && (bi < 0 || v > bv || (v == bv && i < bi))) { bv = v; bi = i; }
}
__device__ __forceinline__ void take_min(double& bv, int64_t& bi, double v, int64_t i) {
  if (i >= 0 && (bi < 0 || v < bv || (v == bv && i < bi))) { bv = v; bi = i; }
}

// Lane exchange inside a row of 16 lanes through the data-parallel-primitive path of the vector ALU (a few cycles;
// a shuffle through the LDS crossbar costs a hundred and more, and the arg-reductions below sit on the critical
// path of every short set-up kernel): quad_perm [1,0,3,2] and [2,3,0,1], row_half_mirror, row_mirror.
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141, kDppMirror = 0x140;
template <int kCtrl> __device__ __forceinline__ int dpp_i32(int x) { return __builtin_amdgcn_update_dpp(x, x, kCtrl, 0xF, 0xF, false); }
__device__ __forceinline__ int64_t readlane_i64(int64_t x, int lane) {
  const int lo = __builtin_amdgcn_readlane((int)(uint32_t)x, lane), hi = __builtin_amdgcn_readlane((int)(x >> 32), lane);
  return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}

// The arg-reductions run on an integer image of the pair: a 64-bit key that grows with "better" (the bits of the
// double made order-preserving, complemented for a minimum; both zeros share a key; 0 = no entry, which no number
// maps to) and the index as 32 bits (rows, local and global columns all fit; ~0 = none).  "Better key, lower index on
// ties" is then a 64-bit maximum followed by a 32-bit minimum among the lanes that hold that maximum: straight-line
// integer code, where the same rule written on (double, int64) compiles to a ladder of branches several times as long,
// and these reductions sit on the critical path of every short set-up kernel (a lone wavefront issues an instruction
// every four to five cycles).
constexpr uint64_t kSignBit = 0x8000000000000000ull;
constexpr uint32_t kNoIndex = 0xFFFFFFFFu;
template <bool kMax>
__device__ __forceinline__ uint64_t arg_key(double v, int64_t i) {
  uint64_t b = (uint64_t)__double_as_longlong(v);
  if ((b << 1) == 0) b = 0;                                   // -0.0 ranks with +0.0
  uint64_t key = (b & kSignBit) ? ~b : (b | kSignBit);         // ascending with v
  if (!kMax) key = ~key;
  return i >= 0 ? key : 0;
}
template <bool kMax>
__device__ __forceinline__ double arg_value(uint64_t key) {    // the number behind a key (+0.0 for either zero)
  if (!kMax) key = ~key;
  const uint64_t b = (key & kSignBit) ? (key & ~kSignBit) : ~key;
  return __longlong_as_double((long long)b);
}
template <int kCtrl> __device__ __forceinline__ uint64_t dpp_u64(uint64_t x) {
  const int lo = dpp_i32<kCtrl>((int)(uint32_t)x), hi = dpp_i32<kCtrl>((int)(x >> 32));
  return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
}
__device__ __forceinline__ uint64_t readlane_u64(uint64_t x, int lane) {
  const int lo = __builtin_amdgcn_readlane((int)(uint32_t)x, lane), hi = __builtin_amdgcn_readlane((int)(x >> 32), lane);
  return ((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo;
}
__device__ __forceinline__ uint64_t max_u64(uint64_t a, uint64_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t min_u32(uint32_t a, uint32_t b) { return a < b ? a : b; }

// Best (key, index) over the 64 lanes of a wavefront; EVERY lane gets it.  All lanes must be active.  Butterfly
// inside the rows of 16 lanes, then the four rows through lane reads.
__device__ __forceinline__ void wave_best(uint64_t& key, uint32_t& idx) {
  uint64_t k = key;
  k = max_u64(k, dpp_u64<kDppXor1>(k));
  k = max_u64(k, dpp_u64<kDppXor2>(k));
  k = max_u64(k, dpp_u64<kDppHalfMirror>(k));
  k = max_u64(k, dpp_u64<kDppMirror>(k));
  k = max_u64(max_u64(readlane_u64(k, 0), readlane_u64(k, 16)), max_u64(readlane_u64(k, 32), readlane_u64(k, 48)));
  uint32_t c = key == k ? idx : kNoIndex;
  c = min_u32(c, (uint32_t)dpp_i32<kDppXor1>((int)c));
  c = min_u32(c, (uint32_t)dpp_i32<kDppXor2>((int)c));
  c = min_u32(c, (uint32_t)dpp_i32<kDppHalfMirror>((int)c));
  c = min_u32(c, (uint32_t)dpp_i32<kDppMirror>((int)c));
  c = min_u32(min_u32((uint32_t)__builtin_amdgcn_readlane((int)c, 0), (uint32_t)__builtin_amdgcn_readlane((int)c, 16)),
              min_u32((uint32_t)__builtin_amdgcn_readlane((int)c, 32), (uint32_t)__builtin_amdgcn_readlane((int)c, 48)));
  key = k;
  idx = c;
}

// Arg-reduce over the 64 lanes of a wavefront ("better value, lower index on ties", index < 0 = no entry); EVERY lane
// gets the winner (index -1 when no lane had an entry).  All lanes must be active.
template <bool kMax>
__device__ __forceinline__ void wave_arg(double& v, int64_t& i) {
  uint64_t key = arg_key<kMax>(v, i);
  uint32_t idx = i >= 0 ? (uint32_t)i : kNoIndex;
  wave_best(key, idx);
  v = key ? arg_value<kMax>(key) : (kMax ? -DBL_MAX : DBL_MAX);
  i = key ? (int64_t)idx : -1;
}

// Wavefront-level arg-reduce, then every thread folds the wavefronts' winners from LDS: every thread of the
// workgroup gets the result.  s_v / s_i hold one entry per wavefront (key bits and index) and may be reused after
// the call.
template <bool kMax, int kThreads>
__device__ __forceinline__ void block_arg(double& v, int64_t& i, double* s_v, int64_t* s_i) {
  constexpr int kWaves = kThreads / kWave;
  uint64_t key = arg_key<kMax>(v, i);
  uint32_t idx = i >= 0 ? (uint32_t)i : kNoIndex;
  wave_best(key, idx);
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  if (lane == 0) { s_v[wave] = __longlong_as_double((long long)key); s_i[wave] = (int64_t)idx; }
  __syncthreads();
  uint64_t ks[kWaves];
  uint32_t cs[kWaves];
#pragma unroll
  for (int w = 0; w < kWaves; w++) { ks[w] = (uint64_t)__double_as_longlong(s_v[w]); cs[w] = (uint32_t)s_i[w]; }
  uint64_t k = ks[0];
#pragma unroll
  for (int w = 1; w < kWaves; w++) k = max_u64(k, ks[w]);
  uint32_t c = kNoIndex;
#pragma unroll
  for (int w = 0; w < kWaves; w++) c = min_u32(c, ks[w] == k ? cs[w] : kNoIndex);
  __syncthreads();
  v = k ? arg_value<kMax>(k) : (kMax ? -DBL_MAX : DBL_MAX);
  i = k ? (int64_t)c : -1;
}

// ------------------------------------------------------------------ pricing
// This thread's share of up to kMaxPartials (value, index) partial results, folded into (v, i).  The loads of four
// rounds are issued before the first compare: a round trip to data the launch before has just written costs half a
// microsecond and more, and 256 < n <= 1024 is the usual case on HBM-sized shards.
template <bool kMax, int kThreads>
__device__ __forceinline__ void fold_share(const double* __restrict__ pv, const int64_t* __restrict__ pi, int n, double& v, int64_t& i) {
  for (int k0 = threadIdx.x; k0 < n; k0 += 4 * kThreads) {
    double vv[4];
    int64_t ii[4];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int k = k0 + q * kThreads;
      vv[q] = k < n ? pv[k] : 0.0;
      ii[q] = k < n ? pi[k] : -1;
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      if (kMax) take_max(v, i, vv[q], ii[q]); else take_min(v, i, vv[q], ii[q]);
    }
  }
}

// Folds up to kMaxPartials (value, index) partial results; every thread of the
// workgroup gets the winner.  Used redundantly by all workgroups of the kernel
// that consumes the partials, which avoids an extra launch or a grid barrier.
template <bool kMax, int kThreads>
__device__ __forceinline__ void fold_partials(const double* __restrict__ pv, const int64_t* __restrict__ pi, int n,
                                              double& v, int64_t& i, double* s_v, int64_t* s_i) {
  v = kMax ? -DBL_MAX : DBL_MAX;
  i = -1;
  fold_share<kMax, kThreads>(pv, pi, n, v, i);
  block_arg<kMax, kThreads>(v, i, s_v, s_i);
}

// MAX_COST pricing (reference lp_simplex_tableau.cc:188-198): largest positive
// reduced cost over non-basic columns, lowest column index on ties.  Stage 1:
// every workgroup reduces a strided share of the objective row with a
// wavefront-level arg-max and leaves one partial (value, local column).
__global__ __launch_bounds__(kPartialThreads) void k_price_partial(
    const double* __restrict__ obj, const uint8_t* __restrict__ is_basic, Shape sh,
    PivotState* __restrict__ st, double* __restrict__ pv, int64_t* __restrict__ pi) {
  __shared__ double s_v[kPartialThreads / kWave];
  __shared__ int64_t s_i[kPartialThreads / kWave];
  if (st->status != ST_RUNNING) return;
  double bv = -DBL_MAX;
  int64_t bi = -1;
  for (int l = blockIdx.x * kPartialThreads + threadIdx.x; l < sh.nloc; l += gridDim.x * kPartialThreads) {
    if (is_basic[l]) continue;
    const double c = obj[l];
    if (is_pos(c) && c > bv) { bv = c; bi = l; }
  }
  block_arg<true, kPartialThreads>(bv, bi, s_v, s_i);
  if (threadIdx.x == 0) {
    pv[blockIdx.x] = bv;
    pi[blockIdx.x] = bi;
    if (blockIdx.x == 0) st->n_active = 0;   // consumed by the update of the previous pivot, refilled by this one
  }
}

// Stage 2 when the candidate has to leave the GPU (multi-GPU exchange, single
// step API): one workgroup folds the partials into a candidate record.
__global__ __launch_bounds__(kPartialThreads) void k_price_fold(
    const double* __restrict__ pv, const int64_t* __restrict__ pi, int n_part, Shape sh,
    const PivotState* __restrict__ st, Candidate* __restrict__ cand) {
  __shared__ double s_v[kPartialThreads / kWave];
  __shared__ int64_t s_i[kPartialThreads / kWave];
  if (st->status != ST_RUNNING) return;
  double bv;
  int64_t bi;
  fold_partials<true, kPartialThreads>(pv, pi, n_part, bv, bi, s_v, s_i);
  if (threadIdx.x == 0) {
    Candidate c;
    c.key = bv;
    c.ce = bv;
    c.row = -1;
    c.gcol = bi < 0 ? -1 : gcol_of(bi * sh.world + sh.rank, sh.N);
    *cand = c;
  }
}

// MAX_REDUCTION pricing, per-column part (reference lp_simplex_tableau.cc:291-302
// and 341-352): one workgroup per candidate column computes the minimum ratio
// b_i / (-T_ij) over rows with (float)T_ij <= -1e-6f, lowest row on ties.
// crow[l] = -2: not a candidate, -1: candidate without eligible row.
// The pass is a read-only stream over the candidate columns (8*m bytes each, 16 B
// per lane, four loads in flight); the RHS replica comes from L2.  The correctly
// rounded division is only executed for entries that can still beat the lane's
// running minimum: for best > 0, b > best * (-a) * (1 + 2^-40) proves b / (-a) > best
// for any rounding of the products and the quotient, so skipping those entries is exact.
__device__ __forceinline__ void pr_consider(double a, double b, int64_t i, double bound, double& bv, int64_t& bi) {
  if ((float)a > -1e-6f) return;                 // _IsNonNegative (padding rows hold 0 and drop out here too)
  const double na = -a;
  if (bound > 0.0 && b > __dmul_rn(__dmul_rn(bound, na), 1.0000000000009095)) return;
  const double ratio = b / na;
  if (bv > ratio) { bv = ratio; bi = i; }
}

// Smallest running minimum of the wavefront: an upper bound of the column's minimum
// that every lane may use in the filter above (an entry whose ratio exceeds it is
// strictly larger than the minimum, so it can be neither the arg-min nor a tie).
__device__ __forceinline__ double wave_min_all(double v) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off, kWave));
  return v;
}

__global__ __launch_bounds__(256) void k_pr_columns(
    Tab T, const double* __restrict__ obj, const uint8_t* __restrict__ is_basic,
    Shape sh, const PivotState* __restrict__ st, double* __restrict__ cmin, int64_t* __restrict__ crow,
    const int64_t* __restrict__ pr_stamp) {
  __shared__ double s_v[256 / kWave];
  __shared__ int64_t s_i[256 / kWave];
  if (st->status != ST_RUNNING) return;
  const int l = blockIdx.x;
  if (is_basic[l] || !is_pos(obj[l])) {
    if (threadIdx.x == 0) crow[l] = -2;
    return;
  }
  if (pr_stamp[l] == st->iter) return;   // the last update already left this column's minimum ratio
  const double2* __restrict__ col = reinterpret_cast<const double2*>(T.col(l));
  const double2* __restrict__ rhs = reinterpret_cast<const double2*>(T.col(sh.nloc));
  const int n_vec = (int)(sh.lda / 2);           // multiple of 256
  double bv = DBL_MAX, bound = DBL_MAX;
  int64_t bi = -1;
  int v = threadIdx.x;
  for (; v + 3 * 256 < n_vec; v += 4 * 256) {
    double2 a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { a[u] = col[v + u * 256]; b[u] = rhs[v + u * 256]; }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int64_t i = 2 * (int64_t)(v + u * 256);
      pr_consider(a[u].x, b[u].x, i, bound, bv, bi);
      pr_consider(a[u].y, b[u].y, i + 1, bound, bv, bi);
    }
    bound = wave_min_all(bv);
  }
  for (; v < n_vec; v += 256) {
    const double2 a = col[v], b = rhs[v];
    pr_consider(a.x, b.x, 2 * (int64_t)v, bound, bv, bi);
    pr_consider(a.y, b.y, 2 * (int64_t)v + 1, bound, bv, bi);
  }
  block_arg<false, 256>(bv, bi, s_v, s_i);
  if (threadIdx.x == 0) { cmin[l] = bv; crow[l] = bi; }
}

// MAX_REDUCTION pricing, fold over columns (reference lp_simplex_tableau.cc:310-325
// and 353-362): largest c_j * min_j over candidates that have a row, lowest
// column on ties; if no candidate has a row the first candidate is reported
// with row = -1 (=> unbounded).  Stage 1: per-workgroup partials (best score and
// first candidate); stage 2 (k_pr_fold): one workgroup folds them into the record.
__global__ __launch_bounds__(kPartialThreads) void k_pr_partial(
    const double* __restrict__ obj, const double* __restrict__ cmin, const int64_t* __restrict__ crow,
    Shape sh, const PivotState* __restrict__ st, double* __restrict__ pv, int64_t* __restrict__ pi,
    int64_t* __restrict__ pfirst) {
  __shared__ double s_v[kPartialThreads / kWave];
  __shared__ int64_t s_i[kPartialThreads / kWave];
  if (st->status != ST_RUNNING) return;
  double bv = -DBL_MAX, fv = 0.0;
  int64_t bi = -1, fi = -1;
  for (int l = blockIdx.x * kPartialThreads + threadIdx.x; l < sh.nloc; l += gridDim.x * kPartialThreads) {
    const int64_t cr = crow[l];
    if (cr == -2) continue;
    if (fi < 0) fi = l;
    if (cr < 0) continue;
    const double score = __dmul_rn(obj[l], cmin[l]);
    if (score > bv) { bv = score; bi = l; }
  }
  block_arg<true, kPartialThreads>(bv, bi, s_v, s_i);
  fv = fi < 0 ? 0.0 : -(double)fi;   // first candidate = smallest local index: arg-max of -index
  block_arg<true, kPartialThreads>(fv, fi, s_v, s_i);
  if (threadIdx.x == 0) { pv[blockIdx.x] = bv; pi[blockIdx.x] = bi; pfirst[blockIdx.x] = fi; }
}

__global__ __launch_bounds__(kPartialThreads) void k_pr_fold(
    const double* __restrict__ obj, const int64_t* __restrict__ crow,
    const double* __restrict__ pv, const int64_t* __restrict__ pi, const int64_t* __restrict__ pfirst, int n_part,
    Shape sh, PivotState* __restrict__ st, Candidate* __restrict__ cand) {
  __shared__ double s_v[kPartialThreads / kWave];
  __shared__ int64_t s_i[kPartialThreads / kWave];
  if (st->status != ST_RUNNING) return;
  if (st->dirty == 1) { if (threadIdx.x == 0) st->status = ST_DIRTY; return; }
  if (st->iter >= st->max_iter) { if (threadIdx.x == 0) st->status = ST_LIMIT; return; }
  double bv;
  int64_t bi;
  fold_partials<true, kPartialThreads>(pv, pi, n_part, bv, bi, s_v, s_i);
  double fv = 0.0;
  int64_t fi = -1;
  for (int k = threadIdx.x; k < n_part; k += kPartialThreads) {
    const int64_t f = pfirst[k];
    if (f >= 0) take_max(fv, fi, -(double)f, f);
  }
  block_arg<true, kPartialThreads>(fv, fi, s_v, s_i);
  if (threadIdx.x == 0) {
    st->n_active = 0;
    Candidate c;
    if (bi >= 0) {
      c.key = bv; c.gcol = gcol_of(bi * sh.world + sh.rank, sh.N); c.row = crow[bi]; c.ce = obj[bi];
    } else if (fi >= 0) {
      c.key = -DBL_MAX; c.gcol = gcol_of(fi * sh.world + sh.rank, sh.N); c.row = -1; c.ce = obj[fi];
    } else {
      c.key = -DBL_MAX; c.gcol = -1; c.row = -1; c.ce = 0.0;
    }
    *cand = c;
  }
}

// ------------------------------------------------------- ratio test
// MAX_COST ratio test (reference lp_simplex_tableau.cc:207-227), stage 1: every
// workgroup stages its rows' ratios b_i / (-T_ie) through an LDS arg-min (lowest
// row on ties) and leaves one partial.  The pivot column is the local column of
// the entering variable, or `xcol` when its owner broadcast it.
// kFoldPrice: the entering column is first folded from the pricing partials
// (single-GPU loop); otherwise it comes with the candidate record.
template <bool kFoldPrice>
__global__ __launch_bounds__(kPartialThreads) void k_ratio_partial(
    Tab T, Shape sh, PivotState* __restrict__ st,
    const double* __restrict__ pv, const int64_t* __restrict__ pi, int n_price_part,
    Candidate* __restrict__ cand, const double* __restrict__ xcol,
    double* __restrict__ rv, int64_t* __restrict__ ri, const unsigned long long* __restrict__ first_basic = nullptr) {
  // first_basic (the loop that scans like the reference): per row the lowest basic column with a non-zero entry there, ~0
  // if none; a row without one does not take part (reference lp_simplex_tableau.cc:215-226)
  __shared__ double s_v[kPartialThreads / kWave];
  __shared__ int64_t s_i[kPartialThreads / kWave];
  if (st->status != ST_RUNNING) return;
  const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
  if (st->dirty == 1) { if (lead) st->status = ST_DIRTY; return; }
  if (st->iter >= st->max_iter) { if (lead) st->status = ST_LIMIT; return; }
  int64_t e;
  if (kFoldPrice) {
    double ce;
    int64_t l;
    fold_partials<true, kPartialThreads>(pv, pi, n_price_part, ce, l, s_v, s_i);
    if (l < 0) { if (lead) st->status = ST_SOLVED; return; }
    e = gcol_of(l * sh.world + sh.rank, sh.N);
    if (lead) {
      Candidate c;
      c.key = ce; c.gcol = e; c.row = -1; c.ce = ce;
      *cand = c;
    }
  } else {
    e = cand->gcol;
    if (e < 0) { if (lead) st->status = ST_SOLVED; return; }
  }
  const double* __restrict__ col = xcol ? xcol : T.col(ord_local(ordinal_of(e, sh.N), sh.world));
  const double* __restrict__ rhs = T.col(sh.nloc);
  double bv = DBL_MAX;
  int64_t bi = -1;
  for (int64_t i = blockIdx.x * kPartialThreads + threadIdx.x; i < sh.m; i += (int64_t)gridDim.x * kPartialThreads) {
    const double a = col[i];
    if ((float)a > -1e-6f) continue;   // skips _IsZero and _IsNonNegative entries alike
    if (first_basic && first_basic[i] == ~0ull) continue;
    const double ratio = rhs[i] / (-a);
    if (bv > ratio) { bv = ratio; bi = i; }
  }
  block_arg<false, kPartialThreads>(bv, bi, s_v, s_i);
  if (threadIdx.x == 0) {
    rv[blockIdx.x] = bv;
    ri[blockIdx.x] = bi;
    if (blockIdx.x == 0) st->n_active = 0;
  }
}

// The reference's scans, for tableaux whose basic columns are no longer unit columns (PivotState::dirty).
// MAX_COST (reference lp_simplex_tableau.cc:215-226): a row takes part in the ratio test if some basic column has a
// non-zero entry in it, and the first such column (ascending) is the leaving variable: first_basic[i] = lowest global id
// of a basic column with !_IsZero(T[i][col]), ~0 if none (preset by the caller).  One workgroup per local column.
__global__ __launch_bounds__(256) void k_basic_scan_rows(Tab T, Shape sh, const uint8_t* __restrict__ is_basic,
                                                          unsigned long long* __restrict__ first_basic) {
  const int l = blockIdx.x;
  if (!is_basic[l]) return;
  const unsigned long long g = (unsigned long long)gcol_of((int64_t)l * sh.world + sh.rank, sh.N);
  const double* __restrict__ col = T.col(l);
  for (int64_t i = threadIdx.x; i < sh.m; i += 256)
    if (!is_zero(col[i])) atomicMin(&first_basic[i], g);
}
// MAX_REDUCTION (reference lp_simplex_tableau.cc:240-251): for every basic column in ascending order the FIRST row with a
// non-zero entry gets that column as its basic variable, later columns overwriting earlier ones: disc[row] = highest
// global id among the basic columns whose first non-zero row it is, -1 if none (preset by the caller).
__global__ __launch_bounds__(256) void k_basis_discovery(Tab T, Shape sh, const uint8_t* __restrict__ is_basic, long long* __restrict__ disc) {
  __shared__ double s_v[256 / kWave];
  __shared__ int64_t s_i[256 / kWave];
  const int l = blockIdx.x;
  if (!is_basic[l]) return;
  const double* __restrict__ col = T.col(l);
  double bv = DBL_MAX;
  int64_t bi = -1;
  for (int64_t i = threadIdx.x; i < sh.m; i += 256)
    if (bi < 0 && !is_zero(col[i])) { bv = (double)i; bi = i; }
  block_arg<false, 256>(bv, bi, s_v, s_i);
  if (threadIdx.x == 0 && bi >= 0) atomicMax(&disc[bi], (long long)gcol_of((int64_t)l * sh.world + sh.rank, sh.N));
}

// ------------------------------------------------ pivot set-up + row gather
// One launch prepares everything the rank-1 update needs.  Every workgroup first
// folds the ratio partials (or takes the row from the candidate) and derives the
// scalars of TableauPivot (reference lp_simplex_tableau.cc:33-44) redundantly:
//   a = T[r][e], f = -1/a, g = c_e * f.
// Workgroups [0, n_row_blocks) then write their slice of col' (the pivot column
// with col'[r] = 1 + a) into colp; the others handle 256 local columns each (RHS
// replica included): row'[j] = T[r][j] * f, obj[j] = flush(obj[j] + T[r][j] * g),
// the list of columns the update has to touch (row' != 0), and the pricing partial
// of the NEXT iteration on the updated objective row (so the main loop needs no
// separate pricing launch).  The first column workgroup records the pivot.
// Everything of the set-up once the candidate c and the row r are known (see k_pivot_finalize).
// n_active_ctr: the counter the active list is appended under; bump_iter: the pivot count is
// advanced here (otherwise the update kernel does it, see k_pivot_setup_small).
__device__ __forceinline__ void pivot_setup_body(
    const Tab& T, const Shape& sh, const Candidate& c, int64_t r, const double* __restrict__ xcol, int n_row_blocks,
    int64_t d_override, PivotState* __restrict__ st, int32_t* __restrict__ n_active_ctr, bool bump_iter,
    double* __restrict__ colp, double* __restrict__ obj, const uint8_t* __restrict__ is_basic,
    const int64_t* __restrict__ basis, double* __restrict__ rowp, int32_t* __restrict__ active,
    int64_t* __restrict__ trace, int64_t trace_cap, double* __restrict__ next_pv, int64_t* __restrict__ next_pi,
    double* __restrict__ rhs_next, uint8_t* __restrict__ cand_next, PassInfo* __restrict__ pass, double* s_v, int64_t* s_i) {
  const int64_t e = c.gcol, eo = ordinal_of(e, sh.N);
  const bool dense = st->dense_next != 0;   // every column takes this pivot, see PivotState
  const double* __restrict__ col = xcol ? xcol : T.col(ord_local(eo, sh.world));
  LPAMD_CHECK(0, r >= 0 && r < sh.m && eo >= 0 && eo <= sh.N);
  const double a = col[r];
  const double f = -1.0 / a;                         // lp_simplex_tableau.cc:40
  const double g = __dmul_rn(c.ce, f);               // lp_simplex_tableau.cc:34-35
  if ((int)blockIdx.x < n_row_blocks) {
    const double* __restrict__ rhs = T.col(sh.nloc);
    const double rp_rhs = __dmul_rn(rhs[r], f);
    bool big = false;
    for (int64_t i = (int64_t)blockIdx.x * kPartialThreads + threadIdx.x; i < sh.m; i += (int64_t)n_row_blocks * kPartialThreads) {
      const double cpi = (i == r) ? __dadd_rn(1.0, a) : col[i];   // lp_simplex_tableau.cc:41-44
      big = big || !(fabs(cpi) < kDirtyBound);
      colp[i] = cpi;
      // right-hand side after this pivot (the update applies the same formula to the RHS replica, and
      // leaves the column alone when its row' entry is exactly 0)
      rhs_next[i] = (rp_rhs != 0.0 || dense) ? flushz(__dadd_rn(rhs[i], __dmul_rn(cpi, rp_rhs))) : rhs[i];
    }
    if (big) st->dirty_step = 1;   // pivot 0 of the pass (PivotState::dirty)
    return;
  }
  // column workgroups: strided over the local columns so that any width fits kMaxPartials workgroups
  __shared__ int s_cnt[kPartialThreads / kWave];
  __shared__ int s_base;
  const int cb = (int)blockIdx.x - n_row_blocks, n_cb = (int)gridDim.x - n_row_blocks;
  const int ncl = sh.nloc + 1;
  const int64_t d_now = d_override >= 0 ? d_override : basis[r];   // read before the bookkeeping thread overwrites basis[r]
  const int64_t dord = (d_now >= 0 && d_now != sh.N) ? ordinal_of(d_now, sh.N) : -1;
  const int l_e = (ord_rank(eo, sh.world) == sh.rank) ? ord_local(eo, sh.world) : -1;
  const int l_d = (dord >= 0 && ord_rank(dord, sh.world) == sh.rank) ? ord_local(dord, sh.world) : -1;
  double nbv = -DBL_MAX;   // pricing partial of the NEXT iteration, on the updated objective row
  int64_t nbi = -1;
  for (int l0 = cb * kPartialThreads; l0 < ncl; l0 += n_cb * kPartialThreads) {
    const int l = l0 + threadIdx.x;
    bool act = false;
    if (l < ncl) {
      const double t = T.col(l)[r];
      const double rp = __dmul_rn(t, f);
      double o = flushz(__dadd_rn(obj[l], __dmul_rn(t, g)));   // :36
      if (l == l_e) o = 0.0;                                     // :37
      obj[l] = o;
      rowp[l] = rp;
      act = rp != 0.0 || dense;
      if (l < sh.nloc) {
        const bool basic_next = l == l_e ? true : (l == l_d ? false : is_basic[l] != 0);
        const bool candidate = !basic_next && is_pos(o);
        cand_next[l] = candidate ? 1 : 0;
        if (candidate && o > nbv) { nbv = o; nbi = l; }
      }
    }
    // append to the active list: one reservation per workgroup (the counter is a single word every workgroup adds to)
    const unsigned long long mask = __ballot(act);
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    if (lane == 0) s_cnt[wave] = __popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
      int n = 0;
      for (int w = 0; w < kPartialThreads / kWave; w++) n += s_cnt[w];
      s_base = n ? atomicAdd(n_active_ctr, n) : 0;
    }
    __syncthreads();
    int base = s_base;
    for (int w = 0; w < wave; w++) base += s_cnt[w];
    LPAMD_CHECK(1, !act || (base >= 0 && base + __popcll(mask & ((1ull << lane) - 1ull)) < ncl));
    if (act) active[base + __popcll(mask & ((1ull << lane) - 1ull))] = l;
    __syncthreads();   // s_cnt / s_base are written again in the next round
  }
  block_arg<true, kPartialThreads>(nbv, nbi, s_v, s_i);
  if (threadIdx.x == 0) { next_pv[cb] = nbv; next_pi[cb] = nbi; }
  const int l = cb * kPartialThreads + threadIdx.x;
  if (l == 0) {
    // The basis flags and basis[r] are read by every workgroup of this launch, so
    // they are flipped by the update kernel that follows (k_update, workgroup 0).
    const int64_t d = d_now;
    st->e = e; st->r = r; st->d = d;
    st->a = a; st->ce = c.ce; st->f = f; st->g = g;
    st->l_e = l_e; st->l_d = l_d;
    pass->n_steps = 0; pass->n_full = 0; pass->n_part = 0; pass->touched = 0;   // filled by k_chain_list when the pass is a chain
    const int64_t it = st->iter;
    LPAMD_CHECK(2, it >= 0 && cb == 0);
    if (it < trace_cap) { trace[2 * it] = e; trace[2 * it + 1] = r; }
    if (bump_iter) st->iter = it + 1;
  }
}

template <bool kFindRow>
__global__ __launch_bounds__(kPartialThreads) void k_pivot_finalize(
    Tab T, Shape sh, const Candidate* __restrict__ cand, const double* __restrict__ xcol,
    const double* __restrict__ rv, const int64_t* __restrict__ ri, int n_ratio_part, int n_row_blocks,
    int64_t d_override, PivotState* __restrict__ st, double* __restrict__ colp,
    double* __restrict__ obj, const uint8_t* __restrict__ is_basic, const int64_t* __restrict__ basis,
    double* __restrict__ rowp, int32_t* __restrict__ active, int64_t* __restrict__ trace, int64_t trace_cap,
    double* __restrict__ next_pv, int64_t* __restrict__ next_pi,
    double* __restrict__ rhs_next, uint8_t* __restrict__ cand_next, PassInfo* __restrict__ pass) {
  __shared__ double s_v[kPartialThreads / kWave];
  __shared__ int64_t s_i[kPartialThreads / kWave];
  if (st->status != ST_RUNNING) return;
  const Candidate c = *cand;
  if (c.gcol < 0) { if (blockIdx.x == 0 && threadIdx.x == 0) st->status = ST_SOLVED; return; }   // no entering column
  int64_t r;
  if (kFindRow) {
    double best;
    fold_partials<false, kPartialThreads>(rv, ri, n_ratio_part, best, r, s_v, s_i);
  } else {
    r = c.row;
  }
  if (r < 0) { if (blockIdx.x == 0 && threadIdx.x == 0) st->status = ST_UNBOUNDED; return; }
  pivot_setup_body(T, sh, c, r, xcol, n_row_blocks, d_override, st, &st->n_active, true, colp, obj, is_basic, basis, rowp, active,
                   trace, trace_cap, next_pv, next_pi, rhs_next, cand_next, pass, s_v, s_i);
}

// Two-launch MAX_COST iteration for short columns on one GPU (where a pivot is bound by launches,
// not by HBM): pricing fold, ratio test and set-up in ONE kernel, then k_update.  Every workgroup
// folds the pricing partials the previous set-up left and runs the whole ratio test itself (2 * 8 * m
// bytes from L2 and m / 256 divisions per lane, cheaper than a kernel boundary while m is a few
// thousand), so all workgroups know (e, r) without a grid-wide exchange.  What the three-launch loop
// settles at kernel boundaries is kept apart differently here:
//  - the pricing partials are read from one buffer and those of the next iteration written to the
//    other (the host swaps them per iteration);
//  - the active-column counters alternate (slot), k_update clears the one it does not read;
//  - the pivot count is advanced by k_update, so that every workgroup of this launch sees the
//    same count in the budget test.
__global__ __launch_bounds__(kPartialThreads) void k_pivot_setup_small(
    Tab T, Shape sh, const double* __restrict__ pv, const int64_t* __restrict__ pi, int n_price_part, int n_row_blocks,
    PivotState* __restrict__ st, int slot, double* __restrict__ colp,
    double* __restrict__ obj, const uint8_t* __restrict__ is_basic, const int64_t* __restrict__ basis,
    double* __restrict__ rowp, int32_t* __restrict__ active, int64_t* __restrict__ trace, int64_t trace_cap,
    double* __restrict__ next_pv, int64_t* __restrict__ next_pi,
    double* __restrict__ rhs_next, uint8_t* __restrict__ cand_next, PassInfo* __restrict__ pass) {
  __shared__ double s_v[kPartialThreads / kWave];
  __shared__ int64_t s_i[kPartialThreads / kWave];
  if (st->status != ST_RUNNING) return;
  const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
  if (st->dirty == 1) { if (lead) st->status = ST_DIRTY; return; }
  if (st->iter >= st->max_iter) { if (lead) st->status = ST_LIMIT; return; }
  double ce;
  int64_t l;
  fold_partials<true, kPartialThreads>(pv, pi, n_price_part, ce, l, s_v, s_i);   // reference lp_simplex_tableau.cc:188-198
  if (l < 0) { if (lead) st->status = ST_SOLVED; return; }
  Candidate c;
  c.key = ce; c.gcol = gcol_of(l * sh.world + sh.rank, sh.N); c.row = -1; c.ce = ce;
  // ratio test (reference lp_simplex_tableau.cc:207-227), the whole column in every workgroup
  const double* __restrict__ col = T.col(l);
  const double* __restrict__ rhs = T.col(sh.nloc);
  double bv = DBL_MAX;
  int64_t bi = -1;
  for (int64_t i = threadIdx.x; i < sh.m; i += kPartialThreads) {
    const double a = col[i];
    if ((float)a > -1e-6f) continue;   // skips _IsZero and _IsNonNegative entries alike
    const double ratio = rhs[i] / (-a);
    if (bv > ratio) { bv = ratio; bi = i; }
  }
  block_arg<false, kPartialThreads>(bv, bi, s_v, s_i);   // every thread gets the row
  const int64_t r = bi;
  if (r < 0) { if (lead) st->status = ST_UNBOUNDED; return; }
  pivot_setup_body(T, sh, c, r, nullptr, n_row_blocks, -1, st, slot ? &st->n_active2 : &st->n_active, false, colp, obj, is_basic, basis,
                   rowp, active, trace, trace_cap, next_pv, next_pi, rhs_next, cand_next, pass, s_v, s_i);
}

// ------------------------------------------------------------ rank-1 update
// T[:, j] = flush(T[:, j] + col' * row'[j]) for every active column j
// (reference lp_simplex_tableau.cc:45-46).  HBM-bound: every entry of an
// active column is read once and written once (16 B/entry); nothing else
// touches HBM (col' is m doubles and stays in L2, row'[j] is one scalar per
// column).  A workgroup streams whole columns (contiguous lda*8 bytes each),
// 16 B per lane per access, kUnroll accesses in flight per lane, columns dealt
// round-robin over the grid so the whole chip sweeps the shard front to back.
// kNonTemporal marks the loads/stores `nt`: measured +4..5 % on shards that
// exceed the 256 MiB Infinity Cache, and a loss on shards that fit in it.
template <bool kNT> __device__ __forceinline__ double2 ld16(const double2* p) {
  if (kNT) {
    double2 v;
    v.x = __builtin_nontemporal_load(&p->x);
    v.y = __builtin_nontemporal_load(&p->y);
    return v;
  }
  return *p;
}
template <bool kNT> __device__ __forceinline__ void st16(double2* p, double2 v) {
  if (kNT) {
    __builtin_nontemporal_store(v.x, &p->x);
    __builtin_nontemporal_store(v.y, &p->y);
  } else {
    *p = v;
  }
}
// The same accesses through an address kept as an integer (a table of column addresses read with scalar loads):
// the pointer is rebuilt in the global address space, so that the access is a global one (a flat access would count
// against the LDS / scalar-load counter as well and every wait for an LDS read would wait for it) with the uniform
// base in scalar registers and the lane's index as the offset.
typedef double d2v __attribute__((ext_vector_type(2)));
template <bool kNT> __device__ __forceinline__ double2 ld16g(uint64_t base, uint32_t idx) {
  const __attribute__((address_space(1))) d2v* p = (const __attribute__((address_space(1))) d2v*)base;
  const d2v v = kNT ? __builtin_nontemporal_load(p + idx) : p[idx];
  double2 o;
  o.x = v.x;
  o.y = v.y;
  return o;
}
template <bool kNT> __device__ __forceinline__ void st16g(uint64_t base, uint32_t idx, double2 v) {
  __attribute__((address_space(1))) d2v* p = (__attribute__((address_space(1))) d2v*)base;
  d2v w;
  w.x = v.x;
  w.y = v.y;
  if (kNT) __builtin_nontemporal_store(w, p + idx); else p[idx] = w;
}
__device__ __forceinline__ double2 rank1(double2 t, double2 c, double rj) {
  double2 o;
  o.x = flushz(__dadd_rn(t.x, __dmul_rn(c.x, rj)));
  o.y = flushz(__dadd_rn(t.y, __dmul_rn(c.y, rj)));
  return o;
}

// kFusePR: while a candidate column of the NEXT max-reduction pricing streams through
// the registers, its minimum ratio b'_i / (-T'_ij) is taken on the updated values
// (b' = rhs_next), which saves the separate 8*m*n read pass of k_pr_columns.
template <int kUnroll, bool kNT, bool kPreC, bool kFusePR>
__global__ __launch_bounds__(kUpdateThreads, 2) void k_update(   // register budget for 2*kUnroll 16-byte loads in flight
    Tab T, int64_t lda, const double* __restrict__ colp,
    const double* __restrict__ rowp, const int32_t* __restrict__ active,
    const PivotState* __restrict__ st, Shape sh, uint8_t* __restrict__ is_basic, int64_t* __restrict__ basis,
    const double* __restrict__ rhs_next, const uint8_t* __restrict__ cand_next,
    double* __restrict__ cmin, int64_t* __restrict__ crow, int64_t* __restrict__ pr_stamp, int slot) {
  // slot < 0: three-launch loop (one counter, the set-up kernel advances the pivot count);
  // slot 0 / 1: two-launch loop (k_pivot_setup_small): read counter `slot`, clear the other one and
  // advance the pivot count here
  __shared__ double s_v[kUpdateThreads / kWave];
  __shared__ int64_t s_i[kUpdateThreads / kWave];
  if (st->status != ST_RUNNING) return;
  const int n_active = slot == 1 ? st->n_active2 : st->n_active;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    PivotState* wst = const_cast<PivotState*>(st);   // nothing in this launch reads these fields again
    if (slot >= 0) {
      if (slot == 1) wst->n_active = 0; else wst->n_active2 = 0;
      wst->iter = st->iter + 1;
    }
    wst->dense_next = 0;   // this update reaches every column the pivot had to reach
    if (st->dirty_step) { if (st->dirty == 0) wst->dirty = 1; wst->dirty_step = 0; }
    // bookkeeping of reference lp_simplex_tableau.cc:16-28 for the pivot set up by k_pivot_finalize
    const int64_t e = st->e, d = st->d, eo = ordinal_of(e, sh.N);
    if (ord_rank(eo, sh.world) == sh.rank) is_basic[ord_local(eo, sh.world)] = 1;
    if (d >= 0 && d != sh.N) {
      const int64_t dord = ordinal_of(d, sh.N);
      if (ord_rank(dord, sh.world) == sh.rank) is_basic[ord_local(dord, sh.world)] = 0;
    }
    basis[st->r] = e;
  }
  const int n_vec = (int)(lda / 2);                  // 16-byte vectors per column, a multiple of 256
  const double2* __restrict__ cp = reinterpret_cast<const double2*>(colp);
  const double2* __restrict__ bn = reinterpret_cast<const double2*>(rhs_next);
  const int64_t stamp = st->iter;
  for (int k = blockIdx.x; k < n_active; k += gridDim.x) {
    const int l = active[k];
    LPAMD_CHECK(3, n_active <= sh.nloc + 1 && l >= 0 && l <= sh.nloc);
    const double rj = rowp[l];
    double2* __restrict__ p = reinterpret_cast<double2*>(T.col(l));
    const bool price = kFusePR && l < sh.nloc && cand_next[l] != 0;   // uniform over the workgroup
    double bv = DBL_MAX, bound = DBL_MAX;
    int64_t bi = -1;
    int v = threadIdx.x;
    for (; v + (kUnroll - 1) * kUpdateThreads < n_vec; v += kUnroll * kUpdateThreads) {
      double2 t[kUnroll], c[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; u++) {
        t[u] = ld16<kNT>(p + v + u * kUpdateThreads);
        if (kPreC) c[u] = cp[v + u * kUpdateThreads];
      }
      // Every HBM load of the batch is issued before the first store: long read
      // bursts followed by long write bursts are what the memory system rewards
      // (interleaved load/store streams measured 10 % slower).  col' comes from L2.
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < kUnroll; u++) {
        if (!kPreC) c[u] = cp[v + u * kUpdateThreads];
        t[u] = rank1(t[u], c[u], rj);
        st16<kNT>(p + v + u * kUpdateThreads, t[u]);
      }
      if (kFusePR && price) {
        // third phase, after the whole store burst has been issued: the updated values are
        // still in registers, the new right-hand side comes from L2
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < kUnroll; u++) c[u] = bn[v + u * kUpdateThreads];
#pragma unroll
        for (int u = 0; u < kUnroll; u++) {
          const int64_t i = 2 * (int64_t)(v + u * kUpdateThreads);
          pr_consider(t[u].x, c[u].x, i, bound, bv, bi);
          pr_consider(t[u].y, c[u].y, i + 1, bound, bv, bi);
        }
        bound = wave_min_all(bv);
      }
    }
    for (; v < n_vec; v += kUpdateThreads) {
      const double2 o = rank1(ld16<kNT>(p + v), cp[v], rj);
      st16<kNT>(p + v, o);
      if (kFusePR && price) {
        const double2 b = bn[v];
        pr_consider(o.x, b.x, 2 * (int64_t)v, bound, bv, bi);
        pr_consider(o.y, b.y, 2 * (int64_t)v + 1, bound, bv, bi);
      }
    }
    if (kFusePR && price) {
      block_arg<false, kUpdateThreads>(bv, bi, s_v, s_i);
      if (threadIdx.x == 0) { cmin[l] = bv; crow[l] = bi; pr_stamp[l] = stamp; }
    }
  }
}

// ------------------------------------------ several pivots per pass (MAX_COST)
// Once k_pivot_finalize has set pivot 0 of a pass up, everything the next pivot needs is
// known without running the rank-1 update: the objective row after pivot 0 (pricing), the
// next entering column and the right-hand side as the update will leave them (ratio
// test), and the next pivot row.  Each of those is ONE column or ONE row, rebuilt with the
// update's own arithmetic, flush(T (+) col'_0 * row'_0); the same holds again after the
// second pivot has been set up, and so on.  A pass sets up a chain of up to kMaxChain pivots
// this way and then streams the tableau ONCE; every entry takes the rank-1 steps of the
// chain in their order,
//   T[i][j] = flush(... flush(flush(T[i][j] + col'_0[i]*row'_0[j]) + col'_1[i]*row'_1[j]) ...),
// which is the same sequence of separately rounded operations as one pass per pivot:
// 16*m*n bytes of HBM traffic for the whole chain.  The chain ends early when the budget
// is used up, at the optimum or when a column has no eligible row; the pass then applies
// what has been set up and the next iteration of the loop reports the outcome exactly as
// the one-pivot loop does.  Step s keeps col'_s at colp + s*lda, row'_s at rowp + s*rstride and
// the right-hand side after steps 0..s at rhsv + s*lda.
// Record 0 is free (pivot 0 of a pass lives in PivotState) and keeps running totals over the passes
// since profiling was switched on: e = columns touched summed over the pivots, r = passes,
// d = columns streamed, a = pivots.  Written by the bookkeeping thread of the pass kernel only.
__device__ __forceinline__ void chain_account(ChainStep* steps, const PassInfo& pass) {
  steps[0].e += pass.touched;
  steps[0].r += 1;
  steps[0].d += pass.n_full + pass.n_part;
  steps[0].a += (double)pass.n_steps;
}

// v as steps 0..j-1 leave it.  rs points at row'_0 of the entry's column (stride rstride between steps),
// cs at col'_0 of its row (stride lda).  A step whose row' entry is exactly 0 leaves the column alone.
// The operands of several steps are fetched before the first of them is applied: the steps depend on each
// other, their loads do not, and one round trip to L2 per step is what these short kernels would otherwise cost.
template <int kBatch>
__device__ __forceinline__ double chain_apply_batch(double v, int s, const double* __restrict__ cs, int64_t lda,
                                                    const double* __restrict__ rs, int64_t rstride) {
  double r[kBatch], c[kBatch];
#pragma unroll
  for (int q = 0; q < kBatch; q++) { r[q] = rs[(s + q) * rstride]; c[q] = cs[(s + q) * lda]; }
#pragma unroll
  for (int q = 0; q < kBatch; q++)
    if (r[q] != 0.0) v = flushz(__dadd_rn(v, __dmul_rn(c[q], r[q])));
  return v;
}
__device__ __forceinline__ double chain_apply(double v, int j, const double* __restrict__ cs, int64_t lda,
                                              const double* __restrict__ rs, int64_t rstride) {
  int s = 0;
  for (; s + 8 <= j; s += 8) v = chain_apply_batch<8>(v, s, cs, lda, rs, rstride);
  if (s + 4 <= j) { v = chain_apply_batch<4>(v, s, cs, lda, rs, rstride); s += 4; }
  if (s + 2 <= j) { v = chain_apply_batch<2>(v, s, cs, lda, rs, rstride); s += 2; }
  if (s < j) v = chain_apply_batch<1>(v, s, cs, lda, rs, rstride);
  return v;
}

// Step j >= 1: pricing fold on the objective row after step j-1, the entering column as steps
// 0..j-1 will leave it (written to colv, the tableau is not touched) and the ratio-test partials of
// reference lp_simplex_tableau.cc:207-227 on it and on the right-hand side after step j-1.
// kExternal (column shards): the entering column has been chosen among all ranks and arrives,
// already rebuilt by its owner, in `colv` together with the candidate record `xcand`.
// What decides whether the step exists is requested together with the pricing partials, before the first wait.
template <bool kExternal>
__global__ __launch_bounds__(kPartialThreads) void k_chain_ratio(
    Tab T, Shape sh, PivotState* __restrict__ st, ChainStep* __restrict__ steps, int j, int depth,
    const double* __restrict__ pv, const int64_t* __restrict__ pi, int n_price_part,
    const double* __restrict__ colp, const double* __restrict__ rowp, int64_t rstride, const double* __restrict__ rhsv,
    double* __restrict__ colv, double* __restrict__ rv, int64_t* __restrict__ ri, const Candidate* __restrict__ xcand) {
  __shared__ double s_v[kPartialThreads / kWave];
  __shared__ int64_t s_i[kPartialThreads / kWave];
  LPAMD_CLKR_DECL;
  LPAMD_CLKR(0, blockIdx.x == 0);
  const int64_t stride = (int64_t)gridDim.x * kPartialThreads;
  const int64_t i0 = (int64_t)blockIdx.x * kPartialThreads + threadIdx.x;
  const bool have0 = i0 < sh.m;
  const double* __restrict__ rhs = rhsv + (int64_t)(j - 1) * sh.lda;
  const int32_t status = st->status;
  LPAMD_CLKR(7, blockIdx.x == 0 && status != 99);
  const int32_t prev_done = j > 1 ? steps[j - 1].done : 1;
  const int64_t iter = st->iter, max_iter = st->max_iter;
  const double rhs0 = have0 ? rhs[i0] : 0.0;
  double ce = -DBL_MAX;
  int64_t l = -1;
  int64_t xe = -1;
  double xce = 0.0;
  if (kExternal) {
    xe = xcand->gcol;
    xce = xcand->ce;
  } else {
    fold_share<true, kPartialThreads>(pv, pi, n_price_part, ce, l);
  }
  if (status != ST_RUNNING) return;
  LPAMD_CLKR(4, blockIdx.x == 0);
  const bool lead = blockIdx.x == 0 && threadIdx.x == 0;
  if (j == 1) {
    if (lead) for (int s = 1; s < depth; s++) { steps[s].done = 0; steps[s].e = -1; }
  } else if (!prev_done) {
    return;                               // the chain ended before this step
  }
  if (iter >= max_iter) return;           // the budget ends with step j-1
  if (kExternal) {
    if (xe < 0) return;                   // optimal after step j-1
    if (lead) { steps[j].e = xe; steps[j].ce = xce; }
    l = 0;
  } else {
    block_arg<true, kPartialThreads>(ce, l, s_v, s_i);
    if (l < 0) return;                    // optimal after step j-1
    l = __builtin_amdgcn_readfirstlane((int)l);   // the same in every lane: row' of this column comes through scalar loads
    if (lead) { steps[j].e = gcol_of(l * sh.world + sh.rank, sh.N); steps[j].ce = ce; }
  }
  LPAMD_CLKR(1, blockIdx.x == 0);
  LPAMD_CHECK(4, j >= 1 && j < depth && depth <= kMaxChain && l >= 0 && l < sh.nloc);
  const double* __restrict__ col = T.col(l);
  double bv = DBL_MAX;
  int64_t bi = -1;
  for (int64_t i = i0; i < sh.m; i += stride) {
    double a;
    if (kExternal) {
      a = colv[i];
    } else {
      a = chain_apply(col[i], j, colp + i, sh.lda, rowp + l, rstride);
      colv[i] = a;
    }
    if ((float)a > -1e-6f) continue;
    const double ratio = (i == i0 ? rhs0 : rhs[i]) / (-a);
    if (bv > ratio) { bv = ratio; bi = i; }
  }
  LPAMD_CLKR(2, blockIdx.x == 0);
  block_arg<false, kPartialThreads>(bv, bi, s_v, s_i);
  if (threadIdx.x == 0) { rv[blockIdx.x] = bv; ri[blockIdx.x] = bi; }
  LPAMD_CLKR(3, blockIdx.x == 0);
  LPAMD_CLKR_FLUSH(j, 0, blockIdx.x == 0);
}

// Set-up of step j (reference lp_simplex_tableau.cc:10-44 on the tableau after step j-1): row
// fold, scalars, col'_j, the right-hand side after the step, the pivot row gathered with steps
// 0..j-1 applied on the fly, row'_j, the objective row, the pricing partials of the iteration after
// this step.  PivotState keeps describing pivot 0 until the pass kernel has used it.
// kFindRow: the row comes from the ratio-test partials (MAX_COST); otherwise with the candidate
// record `xcand` (MAX_REDUCTION prices column and row together).  kRebuild: the entering column is
// a local column and is rebuilt here through the steps so far; otherwise it is given in `colv`.
// kSlots (column shards): the step is set up straight from the gathered exchange slots.  Every workgroup
// folds the ranks' candidate records itself (largest reduced cost, lowest global column on ties; reference
// lp_simplex_tableau.cc:188-198), takes the winner's column from the winner's slot and folds the ratio-test partials
// its owner left there (k_export_lookahead<true>), which replaces the fold kernel and k_chain_ratio of that step
// (MAX_REDUCTION, kFindRow == false: the records bring their rows, and a record without a row does not compete).
// The budget has been settled by the exporting kernel ("no candidate" once it is used up): this launch advances
// the pivot count, so its workgroups cannot test it themselves.
template <bool kFindRow, bool kRebuild, bool kSlots = false>
__global__ __launch_bounds__(kPartialThreads) void k_chain_finalize(
    Tab T, Shape sh, PivotState* __restrict__ st, ChainStep* __restrict__ steps, int j,
    const double* __restrict__ rv, const int64_t* __restrict__ ri, int n_ratio_part, int n_row_blocks,
    double* __restrict__ colp, double* __restrict__ rowp, int64_t rstride, double* __restrict__ rhsv, const double* __restrict__ colv,
    const Candidate* __restrict__ xcand,
    double* __restrict__ obj, const uint8_t* __restrict__ is_basic, const int64_t* __restrict__ basis,
    int64_t* __restrict__ trace, int64_t trace_cap,
    double* __restrict__ next_pv, int64_t* __restrict__ next_pi,
    const uint64_t* __restrict__ slots = nullptr, int slot_world = 0, int64_t slot_words = 0, int slot_header = 0, int depth = 0) {
  __shared__ double s_v[kPartialThreads / kWave];
  __shared__ int64_t s_i[kPartialThreads / kWave];
  __shared__ int s_w;
  __shared__ int64_t s_e2, s_r2;
  __shared__ double s_ce2;
  LPAMD_CLK(j, 1, 0, blockIdx.x == 0);
  LPAMD_CLK(j, 2, 0, (int)blockIdx.x == n_row_blocks);
  LPAMD_CLK(j, 3, 0, blockIdx.x == gridDim.x - 1);
  // A lone wavefront issues an instruction every four to five cycles and waits 0.3 - 0.5 us for a round trip to data
  // the launches before have just written, so this kernel is as long as its longest wavefront's instruction stream and
  // chain of waits.  Requested before the first wait: what decides whether the step exists, the exchange records (one
  // per lane of the first wavefront), the ratio-test partials, the entries of the row or column this thread owns, and
  // the records of the steps so far, one per LANE of every wavefront (lane 0 stands for pivot 0, which lives in
  // PivotState): "which step pivoted on this row last" is then one ballot, and the basis flags of a column after the
  // steps so far take two lane reads per step instead of two scalar loads and a wait.
  const int lane = threadIdx.x & (kWave - 1);
  const bool row_block = (int)blockIdx.x < n_row_blocks;
  const int cb = (int)blockIdx.x - n_row_blocks, n_cb = (int)gridDim.x - n_row_blocks;
  const int ncl = sh.nloc + 1;
  const double* __restrict__ rhs_prev = rhsv + (int64_t)(j - 1) * sh.lda;
  const int32_t status = st->status;
  const int32_t prev_done = j > 1 ? steps[j - 1].done : 1;
  const int32_t dirty_step = st->dirty_step;   // 0, or the step that wrote a col' entry of kDirtyBound or more, plus one
  Candidate rec;
  rec.key = -DBL_MAX; rec.gcol = -1; rec.row = -1; rec.ce = 0.0;
  if (kSlots && slot_world <= kWave && (int)threadIdx.x < slot_world)
    rec = *reinterpret_cast<const Candidate*>(slots + (size_t)threadIdx.x * slot_words);
  int32_t q_in = -1, q_out = -1;   // step `lane`: local columns that turn basic / non-basic, entering variable, pivot row
  int64_t q_e = -1, q_r = -1;
  if (lane == 0) { q_in = st->l_e; q_out = st->l_d; q_e = st->e; q_r = st->r; }
  else if (lane < j) { q_in = steps[lane].l_in; q_out = steps[lane].l_out; q_e = steps[lane].e; q_r = steps[lane].r; }
  int64_t e2 = kSlots ? -1 : steps[j].e;
  double ce2 = kSlots ? 0.0 : steps[j].ce;
  const int64_t x_gcol = (!kFindRow && !kSlots) ? xcand->gcol : -1, x_row = (!kFindRow && !kSlots) ? xcand->row : -1;
  double best = DBL_MAX;
  int64_t r2 = -1;
  if (kFindRow && !kSlots) fold_share<false, kPartialThreads>(rv, ri, n_ratio_part, best, r2);
  // the row (row blocks) or column (column blocks) whose operands wait in registers
  const int64_t i0 = (int64_t)blockIdx.x * kPartialThreads + threadIdx.x;
  const bool have_row = row_block && i0 < sh.m;
  const int l0 = cb * kPartialThreads + (int)threadIdx.x;
  const bool have_col = !row_block && l0 < ncl;
  double rhs0 = 0.0, colv0 = 0.0, obj0 = 0.0;
  uint8_t basic0 = 0;
  if (have_row) {
    rhs0 = rhs_prev[i0];
    if (!kRebuild && !kSlots) colv0 = colv[i0];
  }
  if (have_col) {
    obj0 = obj[l0];
    if (l0 < sh.nloc) basic0 = is_basic[l0];
  }
  if (status != ST_RUNNING) return;
  const bool keeper = (int)blockIdx.x == n_row_blocks && threadIdx.x == 0;   // the thread that records the step
  if (kSlots && j == 1 && keeper)   // the step records of the pass before (k_chain_ratio does this in the other loops)
    for (int s = 1; s < depth; s++) { steps[s].done = 0; steps[s].e = -1; }
  if (!prev_done) return;
  if (dirty_step != 0 && dirty_step - 1 < j) return;   // a step before this one ends the chain (this launch's own mark is j + 1)
  if (kSlots) {
    if (slot_world <= kWave) {
      if (threadIdx.x < kWave) {
        const bool valid = rec.gcol >= 0 && (kFindRow || rec.row >= 0);   // a record that brings its row (MAX_REDUCTION) needs one
        const double key = valid ? rec.key : -DBL_MAX;
        const int64_t gc = valid ? rec.gcol : -1;
        double bk = key;
        int64_t bg = gc;
        wave_arg<true>(bk, bg);           // larger key, lower global column on ties
        const unsigned long long owner = __ballot(gc >= 0 && gc == bg);
        const int w = bg < 0 ? -1 : __ffsll((long long)owner) - 1;
        if (lane == 0) s_w = w;
        if (lane == w) { s_e2 = rec.gcol; s_ce2 = rec.ce; s_r2 = rec.row; }
      }
    } else if (threadIdx.x == 0) {
      int w = -1;
      Candidate bestc = rec;
      for (int k = 0; k < slot_world; k++) {
        const Candidate c = *reinterpret_cast<const Candidate*>(slots + (size_t)k * slot_words);
        if (c.gcol < 0 || (!kFindRow && c.row < 0)) continue;
        if (w < 0 || c.key > bestc.key || (c.key == bestc.key && c.gcol < bestc.gcol)) { w = k; bestc = c; }
      }
      s_w = w;
      s_e2 = bestc.gcol;
      s_ce2 = bestc.ce;
      s_r2 = bestc.row;
    }
    __syncthreads();
    const int w = s_w;
    if (w < 0) return;                    // optimal after step j-1, or the budget ends with it
    const uint64_t* __restrict__ mine = slots + (size_t)w * slot_words;
    e2 = s_e2;
    ce2 = s_ce2;
    colv = reinterpret_cast<const double*>(mine + slot_header);
    rv = colv + sh.lda;                   // the owner's ratio-test partials: values, then rows
    ri = reinterpret_cast<const int64_t*>(rv + n_ratio_part);
    if (have_row) colv0 = colv[i0];
    if (kFindRow) {
      fold_share<false, kPartialThreads>(rv, ri, n_ratio_part, best, r2);
    } else {
      r2 = s_r2;
    }
  } else {
    if (e2 < 0) return;
    if (!kFindRow) r2 = x_gcol == e2 ? x_row : -1;
  }
  if (kFindRow) block_arg<false, kPartialThreads>(best, r2, s_v, s_i);
  if (r2 < 0) return;   // unbounded after step j-1: found again, and reported, by the next iteration
  // the same in every lane: what hangs on the pivot row alone (col'_s at that row) comes through scalar loads
  r2 = (int64_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(r2 >> 32)) << 32) |
                 (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)r2));
  LPAMD_CLK(j, 1, 1, blockIdx.x == 0);
  LPAMD_CLK(j, 2, 1, (int)blockIdx.x == n_row_blocks);
  auto local_id = [&](int64_t g) -> int {
    if (g < 0 || g == sh.N) return -1;
    const int64_t o = ordinal_of(g, sh.N);
    return (ord_rank(o, sh.world) == sh.rank) ? ord_local(o, sh.world) : -1;
  };
  const int l_e2 = local_id(e2);
  LPAMD_CHECK(5, j >= 1 && j < kMaxChain && r2 < sh.m && e2 >= 0 && e2 <= sh.N + 1 && e2 != sh.N && l_e2 < sh.nloc);
  if (kRebuild && l_e2 < 0) return;   // not reachable: a rebuilt column is a local one
  const double* __restrict__ ecol = kRebuild ? T.col(l_e2) : colv;
  // ---- second round of requests: everything that hangs on the pivot row
  const double* __restrict__ c_r = colp + r2;       // col'_s[r2] at c_r[s * lda]
  double tv0 = 0.0;
  int64_t b_r2 = -1;
  if (!row_block) {
    if (have_col) tv0 = T.col(l0)[r2];
    b_r2 = basis[r2];
  }
  const double rhs_r2 = row_block ? rhs_prev[r2] : 0.0;
  const double a2 = kRebuild ? chain_apply(ecol[r2], j, colp + r2, sh.lda, rowp + l_e2, rstride) : colv[r2];
  const double f2 = -1.0 / a2;
  const double g2 = __dmul_rn(ce2, f2);
  LPAMD_CLK(j, 1, 2, blockIdx.x == 0 && g2 == g2);
  LPAMD_CLK(j, 2, 2, (int)blockIdx.x == n_row_blocks && g2 == g2);
  if (row_block) {
    double* __restrict__ cj = colp + (int64_t)j * sh.lda;
    double* __restrict__ rhs_now = rhsv + (int64_t)j * sh.lda;
    const double rp_rhs = __dmul_rn(rhs_r2, f2);   // row'_j of the right-hand-side column
    const int64_t stride = (int64_t)n_row_blocks * kPartialThreads;
    bool big = false;
    for (int64_t i = i0; i < sh.m; i += stride) {
      const bool first = i == i0;
      const double ci = kRebuild ? chain_apply(ecol[i], j, colp + i, sh.lda, rowp + l_e2, rstride) : (first ? colv0 : colv[i]);
      const double rhs_i = first ? rhs0 : rhs_prev[i];
      const double cpi = (i == r2) ? __dadd_rn(1.0, a2) : ci;
      big = big || !(fabs(cpi) < kDirtyBound);
      cj[i] = cpi;
      rhs_now[i] = rp_rhs != 0.0 ? flushz(__dadd_rn(rhs_i, __dmul_rn(cpi, rp_rhs))) : rhs_i;
    }
    if (big) st->dirty_step = j + 1;   // PivotState::dirty
    LPAMD_CLK(j, 1, 3, blockIdx.x == 0);
    return;
  }
  // the basic variable of row r2 after steps 0..j-1 (basis[] is flipped by the pass kernel): the last step so far
  // that pivoted on this row, else what the pass before left
  int64_t d2 = b_r2;
  {
    const unsigned long long hit = __ballot(lane < j && q_r == r2);
    if (hit) {
      const int last = 63 - __clzll((long long)hit);
      d2 = readlane_i64(q_e, last);
    }
  }
  const int l_d2 = local_id(d2);
  LPAMD_CHECK(6, d2 >= -1 && d2 <= sh.N + 1 && l_d2 < sh.nloc);
  LPAMD_CLK(j, 2, 3, cb == 0 && l_d2 != -7);
  double nbv = -DBL_MAX;
  int64_t nbi = -1;
  for (int l = l0; l < ncl; l += n_cb * kPartialThreads) {
    const bool first = l == l0;
    const double t = chain_apply(first ? tv0 : T.col(l)[r2], j, c_r, sh.lda, rowp + l, rstride);
    const double rp2 = __dmul_rn(t, f2);
    double o = flushz(__dadd_rn(first ? obj0 : obj[l], __dmul_rn(t, g2)));
    if (l == l_e2) o = 0.0;
    obj[l] = o;
    rowp[(int64_t)j * rstride + l] = rp2;
    if (l < sh.nloc) {
      bool basic = (first ? basic0 : is_basic[l]) != 0;   // flags as before the pass, then the steps set up so far
#pragma unroll 1
      for (int s = 0; s < j; s++) {
        const int in_s = __builtin_amdgcn_readlane(q_in, s), out_s = __builtin_amdgcn_readlane(q_out, s);
        if (l == in_s) basic = true; else if (l == out_s) basic = false;
      }
      if (l == l_e2) basic = true; else if (l == l_d2) basic = false;
      if (!basic && is_pos(o) && o > nbv) { nbv = o; nbi = l; }
    }
  }
  LPAMD_CLK(j, 2, 4, cb == 0);
  LPAMD_CLK(j, 3, 1, blockIdx.x == gridDim.x - 1);
  block_arg<true, kPartialThreads>(nbv, nbi, s_v, s_i);
  if (threadIdx.x == 0) { next_pv[cb] = nbv; next_pi[cb] = nbi; }
  LPAMD_CLK(j, 2, 5, cb == 0);
  if (cb == 0 && threadIdx.x == 0) {
    if (kSlots) { steps[j].e = e2; steps[j].ce = ce2; }
    steps[j].r = r2; steps[j].d = d2; steps[j].a = a2; steps[j].f = f2; steps[j].g = g2;
    steps[j].l_in = l_e2; steps[j].l_out = l_d2;
    steps[j].done = 1;
    const int64_t it = st->iter;
    LPAMD_CHECK(7, it >= 0);
    if (it < trace_cap) { trace[2 * it] = e2; trace[2 * it + 1] = r2; }
    st->iter = it + 1;
  }
}

// Steps 1..limit-1 of the pass that have been set up, counted from the front (step 0 lives in PivotState and always
// exists): one lane per step record, so that the flags arrive in one round trip instead of one per step.
__device__ __forceinline__ int chain_steps_done(const ChainStep* __restrict__ steps, int limit) {
  const int lane = threadIdx.x & (kWave - 1);
  const bool done = lane == 0 || (lane < limit && steps[lane].done != 0);
  const unsigned long long mask = __ballot(done);
  const int n = __ffsll((long long)~mask) - 1;      // lanes beyond kMaxChain <= 32 are never done: a zero bit exists
  return n < limit ? n : limit;
}

// Where row'_s of the k-th column of list_full lives when the runs are laid out for k_update_deep: the columns
// in batches of c, inside a batch the c values of one step next to each other.
__device__ __forceinline__ int64_t deep_run_index(int k, int s, int depth, int c) {
  return (int64_t)(k / c) * depth * c + (int64_t)s * c + (k % c);
}

// Once per pass, after the last set-up: the columns the pass has to stream, split into those that
// every step touches (list_full; the pass applies the chain to them without a single test) and those
// that some step leaves alone (list_part: entering and leaving columns of the chain, mostly), and
// the row' values of every listed column side by side (rsl_*[k * stride + s] = row'_s of the k-th
// listed column, zero beyond the chain), which the pass reads as one short run per column.  batch > 0: list_full's
// runs in the layout of deep_run_index instead, and the address of every column of list_full in ptr_full.
// bump_iter: the pivot count still lacks steps 1.. of this pass (k_chain_fused) and is advanced here.
// kListThreads: one reservation in the lists per workgroup (the counters are single words every workgroup adds to), so a
// shard of many columns wants large workgroups (98 k columns: 1024 threads, 96 reservations); a shard of a few thousand
// columns is a dozen such workgroups, too few to hide the loads, and wants small ones (m=4096 n=8192, 12 k columns: 17.4 us
// with 1024 threads, 59.5 k pivots/s; 61.0 k with 256).
template <int kListThreads>
__global__ __launch_bounds__(kListThreads) void k_chain_list(
    const double* __restrict__ rowp, int64_t rstride, int depth, const ChainStep* __restrict__ steps,
    const PivotState* __restrict__ st, int ncl, PassInfo* __restrict__ pass,
    int32_t* __restrict__ list_full, int32_t* __restrict__ list_part, double* __restrict__ rsl_full, double* __restrict__ rsl_part, int stride,
    int batch, Tab T, uint64_t* __restrict__ ptr_full, int bump_iter) {
  constexpr int kWaves = kListThreads / kWave;
  __shared__ int s_nf[kWaves], s_np[kWaves], s_tot[kWaves];
  __shared__ int s_bf, s_bp;
  if (st->status != ST_RUNNING) return;
  const int n_steps = chain_steps_done(steps, depth);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    pass->n_steps = n_steps;
    // steps 1.. were set up by k_chain_fused, which leaves the pivot count alone (its workgroups read it on entry)
    if (bump_iter) const_cast<PivotState*>(st)->iter = st->iter + (n_steps - 1);
  }
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  int nz = 0;
  if (l < ncl) {
#pragma unroll 8
    for (int s = 0; s < n_steps; s++) nz += rowp[s * rstride + l] != 0.0 ? 1 : 0;   // eight loads per wait
  }
  const bool full = nz == n_steps, part = nz > 0 && !full;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const unsigned long long mf = __ballot(full), mp = __ballot(part);
  int total = nz;
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) total += __shfl_down(total, off, kWave);
  if (lane == 0) { s_nf[wave] = __popcll(mf); s_np[wave] = __popcll(mp); s_tot[wave] = total; }
  __syncthreads();
  if (threadIdx.x == 0) {      // the workgroup's share of both lists and of the count, reserved with one atomic each
    int nf = 0, np = 0;
    long long tot = 0;
    for (int w = 0; w < kWaves; w++) { nf += s_nf[w]; np += s_np[w]; tot += s_tot[w]; }
    s_bf = nf ? atomicAdd(&pass->n_full, nf) : 0;
    s_bp = np ? atomicAdd(&pass->n_part, np) : 0;
    if (tot) atomicAdd((unsigned long long*)&pass->touched, (unsigned long long)tot);
  }
  __syncthreads();
  int bf = s_bf, bp = s_bp;
  for (int w = 0; w < wave; w++) { bf += s_nf[w]; bp += s_np[w]; }
  if (full || part) {
    const unsigned long long below = (1ull << lane) - 1ull;
    const int k = full ? bf + __popcll(mf & below) : bp + __popcll(mp & below);
    LPAMD_CHECK(8, k >= 0 && k < ncl && n_steps >= 1 && n_steps <= stride && stride <= kMaxChain);
    (full ? list_full : list_part)[k] = l;
    if (full && batch > 0) {   // batches of `batch` columns, the values of one step side by side (k_update_deep)
#pragma unroll 8
      for (int s = 0; s < stride; s++) rsl_full[deep_run_index(k, s, stride, batch)] = s < n_steps ? rowp[s * rstride + l] : 0.0;
      ptr_full[k] = (uint64_t)T.col(l);   // the addresses of a batch's columns: one scalar load in the pass
    } else {
      double* __restrict__ out = (full ? rsl_full : rsl_part) + (int64_t)k * stride;
#pragma unroll 8
      for (int s = 0; s < stride; s++) out[s] = s < n_steps ? rowp[s * rstride + l] : 0.0;
    }
  }
}

// Bookkeeping of reference lp_simplex_tableau.cc:16-28 for every pivot of the pass (one thread).
__device__ __forceinline__ void chain_bookkeeping(PivotState* st, ChainStep* steps, const PassInfo& pass, const Shape& sh,
                                                  uint8_t* is_basic, int64_t* basis) {
  auto flip = [&](int64_t e, int64_t d, int64_t r) {
    LPAMD_CHECK(9, e >= 0 && e <= sh.N + 1 && e != sh.N && r >= 0 && r < sh.m && d >= -1 && d <= sh.N + 1);
    const int64_t eo = ordinal_of(e, sh.N);
    if (ord_rank(eo, sh.world) == sh.rank) is_basic[ord_local(eo, sh.world)] = 1;
    if (d >= 0 && d != sh.N) {
      const int64_t dord = ordinal_of(d, sh.N);
      if (ord_rank(dord, sh.world) == sh.rank) is_basic[ord_local(dord, sh.world)] = 0;
    }
    basis[r] = e;
  };
  flip(st->e, st->d, st->r);
  for (int s = 1; s < pass.n_steps; s++) flip(steps[s].e, steps[s].d, steps[s].r);
  if (st->dirty_step) { if (st->dirty == 0) st->dirty = 1; st->dirty_step = 0; }
  chain_account(steps, pass);
  if (pass.n_steps > 1) {   // no other workgroup of the launch reads these fields
    const ChainStep& z = steps[pass.n_steps - 1];
    st->e = z.e; st->r = z.r; st->d = z.d; st->a = z.a; st->ce = z.ce; st->f = z.f; st->g = z.g;
  }
}

// Smallest magnitude seen so far, tracked on the HIGH words of the doubles read as floats: for values of one
// sign the bit pattern of a double grows with its magnitude, so does the pattern of its upper half, and a float
// minimum with |.| on its inputs orders those patterns (one instruction for two entries; exponents that would read
// as float NaNs belong to magnitudes above 2^1017 and are skipped, which is what a minimum wants).  The test
// `pattern <= upper half of the threshold` errs to the safe side: it also fires for the few values whose upper half
// equals the threshold's and whose lower half lies above it.
__device__ __forceinline__ float min_magnitude_high(float acc, double a, double b) {
  float out;
  asm("v_min3_f32 %0, %1, |%2|, |%3|" : "=v"(out) : "v"(acc), "v"(__double2hiint(a)), "v"(__double2hiint(b)));
  return out;
}
constexpr uint32_t kDropHigh = (uint32_t)(__builtin_bit_cast(uint64_t, (double)1e-6f) >> 32);

// The pass with the col' vectors read from L2 per column (any chain length, any column length).
template <int kUnroll, bool kNT>
__global__ __launch_bounds__(kUpdateThreads, 2) void k_update_chain(
    Tab T, int64_t lda, const double* __restrict__ colp, const PassInfo* __restrict__ passp,
    const int32_t* __restrict__ list_full, const int32_t* __restrict__ list_part,
    const double* __restrict__ rsl_full, const double* __restrict__ rsl_part, int stride,
    PivotState* __restrict__ st, ChainStep* __restrict__ steps, Shape sh,
    uint8_t* __restrict__ is_basic, int64_t* __restrict__ basis) {
  if (st->status != ST_RUNNING) return;
  const PassInfo pass = *passp;
  if (pass.n_steps <= 0) return;
  if (blockIdx.x == 0 && threadIdx.x == 0) chain_bookkeeping(st, steps, pass, sh, is_basic, basis);
  const int n_vec = (int)(lda / 2);
  const double2* __restrict__ cp = reinterpret_cast<const double2*>(colp);
  const int n_list = pass.n_full + pass.n_part;
  for (int k = blockIdx.x; k < n_list; k += gridDim.x) {
    const bool in_full = k < pass.n_full;
    const int l = in_full ? list_full[k] : list_part[k - pass.n_full];
    LPAMD_CHECK(10, l >= 0 && l <= sh.nloc && n_list <= sh.nloc + 1 && pass.n_steps <= stride);
    const double* __restrict__ rs = in_full ? rsl_full + (int64_t)k * stride : rsl_part + (int64_t)(k - pass.n_full) * stride;
    double2* __restrict__ p = reinterpret_cast<double2*>(T.col(l));
    int v = threadIdx.x;
    for (; v + (kUnroll - 1) * kUpdateThreads < n_vec; v += kUnroll * kUpdateThreads) {
      double2 t[kUnroll];
#pragma unroll
      for (int u = 0; u < kUnroll; u++) t[u] = ld16<kNT>(p + v + u * kUpdateThreads);
      __builtin_amdgcn_sched_barrier(0);   // all loads of the batch before its first store
      for (int s = 0; s < pass.n_steps; s++) {
        const double r = rs[s];
        if (r != 0.0) {
#pragma unroll
          for (int u = 0; u < kUnroll; u++) t[u] = rank1(t[u], cp[s * n_vec + v + u * kUpdateThreads], r);
        }
      }
#pragma unroll
      for (int u = 0; u < kUnroll; u++) st16<kNT>(p + v + u * kUpdateThreads, t[u]);
    }
    for (; v < n_vec; v += kUpdateThreads) {
      double2 o = ld16<kNT>(p + v);
      for (int s = 0; s < pass.n_steps; s++) {
        const double r = rs[s];
        if (r != 0.0) o = rank1(o, cp[s * n_vec + v], r);
      }
      st16<kNT>(p + v, o);
    }
  }
}

// The same pass with the col' vectors in REGISTERS.  k_update_chain re-reads every col'_s per
// column from L2 (8 bytes per entry and step next to the 16 bytes of HBM traffic), which limits a
// long chain.  Here a workgroup owns a slab of 512*kR rows, keeps its kK slices of col' in
// registers and walks the column lists, kC columns (kR*kC 16-byte loads per lane, all issued before
// the first store) in flight; the row' values of a column are one run of scalars (rsl_*), fetched
// when its turn comes.
//
// Full chain over list_full: no test per step.  The drop of |v| < 1e-6 after every add (reference
// lp_simplex_tableau.cc:5-8) is taken out of the inner loop: the chain is first evaluated WITHOUT
// the drop while the smallest |v| seen on the way is tracked (multiply and add per entry, one minimum per two
// entries: 2.5 instructions per entry and step instead of five).  If no lane saw a value below the threshold,
// no drop would have happened and the result is exactly the reference's; otherwise the wavefront
// redoes that column from the loaded values with the drop in place.  A drop is rare (the entering
// column of every step takes it, hardly anything else in a dense tableau).
// Everything else (a shorter chain, list_part, the tail of list_full) takes the step-by-step path.
// Needs lda % (512*kR) == 0.
template <int kK, int kR, int kC, bool kNT>
__global__ __launch_bounds__(kUpdateThreads, 2) void k_update_slab(
    Tab T, int64_t lda, const double* __restrict__ colp, const PassInfo* __restrict__ passp,
    const int32_t* __restrict__ list_full, const int32_t* __restrict__ list_part,
    const double* __restrict__ rsl_full, const double* __restrict__ rsl_part, int stride,
    PivotState* __restrict__ st, ChainStep* __restrict__ steps, Shape sh,
    uint8_t* __restrict__ is_basic, int64_t* __restrict__ basis, int n_slabs) {
  if (st->status != ST_RUNNING) return;
  const PassInfo pass = *passp;
  const int n_steps = pass.n_steps;
  if (n_steps <= 0) return;
  if (blockIdx.x == 0 && threadIdx.x == 0) chain_bookkeeping(st, steps, pass, sh, is_basic, basis);
  const int n_vec = (int)(lda / 2);
  const int slab = blockIdx.x % n_slabs, grp = blockIdx.x / n_slabs, n_grp = gridDim.x / n_slabs;
  const int v0 = slab * (kUpdateThreads * kR) + threadIdx.x;
  const double2* __restrict__ cp = reinterpret_cast<const double2*>(colp);
  double2 c[kK][kR];
#pragma unroll
  for (int s = 0; s < kK; s++)
#pragma unroll
    for (int r = 0; r < kR; r++) {
      c[s][r].x = c[s][r].y = 0.0;
      if (s < n_steps) c[s][r] = cp[s * n_vec + v0 + r * kUpdateThreads];
    }
  int k0 = grp * kC;
  if (n_steps == kK) {
    for (; k0 + kC <= pass.n_full; k0 += n_grp * kC) {
      double2* p[kC];
      double2 t[kC][kR];
#pragma unroll
      for (int u = 0; u < kC; u++) {
        LPAMD_CHECK(11, list_full[k0 + u] >= 0 && list_full[k0 + u] <= sh.nloc && stride >= kK);
        p[u] = reinterpret_cast<double2*>(T.col(list_full[k0 + u])) + v0;
#pragma unroll
        for (int r = 0; r < kR; r++) t[u][r] = ld16<kNT>(p[u] + r * kUpdateThreads);
      }
      __builtin_amdgcn_sched_barrier(0);   // all loads of the batch before its first store
#pragma unroll
      for (int u = 0; u < kC; u++) {
        const double* __restrict__ rs = rsl_full + (int64_t)(k0 + u) * stride;
        float low = FLT_MAX;
#pragma unroll
        for (int s = 0; s < kK; s++) {
          const double rj = rs[s];
#pragma unroll
          for (int r = 0; r < kR; r++) {   // products first, then the minimum over what the step before left
            const double px = __dmul_rn(c[s][r].x, rj), py = __dmul_rn(c[s][r].y, rj);
            if (s > 0) low = min_magnitude_high(low, t[u][r].x, t[u][r].y);
            t[u][r].x = __dadd_rn(t[u][r].x, px);
            t[u][r].y = __dadd_rn(t[u][r].y, py);
          }
        }
#pragma unroll
        for (int r = 0; r < kR; r++) low = min_magnitude_high(low, t[u][r].x, t[u][r].y);
        if (__any(__float_as_uint(low) <= kDropHigh)) {   // a value below the threshold on the way: this column again, with the drop
#pragma unroll
          for (int r = 0; r < kR; r++) t[u][r] = ld16<kNT>(p[u] + r * kUpdateThreads);   // not stored yet
          // row' is fetched again through an offset the compiler cannot see through: otherwise it keeps all the
          // products of the loop above alive for this rare path, which doubles the register count
          int zero;
          asm volatile("s_mov_b32 %0, 0" : "=s"(zero));
          const double* __restrict__ rs2 = rs + zero;
#pragma unroll
          for (int s = 0; s < kK; s++) {   // fully unrolled: col' must stay in registers
            const double rj = rs2[s];
#pragma unroll
            for (int r = 0; r < kR; r++) t[u][r] = rank1(t[u][r], c[s][r], rj);
          }
        }
#pragma unroll
        for (int r = 0; r < kR; r++) st16<kNT>(p[u] + r * kUpdateThreads, t[u][r]);
      }
    }
  }
  // step by step: what is left of list_full (all of it when the chain is shorter than kK), then list_part
  const int n_list = pass.n_full + pass.n_part;
  for (; k0 < n_list; k0 += n_grp * kC) {
#pragma unroll 1
    for (int u = 0; u < kC; u++) {
      const int k = k0 + u;
      if (k >= n_list) break;
      const bool in_full = k < pass.n_full;
      const int l = in_full ? list_full[k] : list_part[k - pass.n_full];
      LPAMD_CHECK(12, l >= 0 && l <= sh.nloc && n_list <= sh.nloc + 1 && n_steps <= kK && n_steps <= stride && (int64_t)(v0 + (kR - 1) * kUpdateThreads) < lda / 2);
      const double* __restrict__ rs = in_full ? rsl_full + (int64_t)k * stride : rsl_part + (int64_t)(k - pass.n_full) * stride;
      double2* __restrict__ p = reinterpret_cast<double2*>(T.col(l)) + v0;
      double2 w[kR];
#pragma unroll
      for (int r = 0; r < kR; r++) w[r] = ld16<kNT>(p + r * kUpdateThreads);
#pragma unroll
      for (int s = 0; s < kK; s++) {
        const double rj = s < n_steps ? rs[s] : 0.0;
        if (rj != 0.0) {
#pragma unroll
          for (int r = 0; r < kR; r++) w[r] = rank1(w[r], c[s][r], rj);
        }
      }
#pragma unroll
      for (int r = 0; r < kR; r++) st16<kNT>(p + r * kUpdateThreads, w[r]);
    }
  }
}

// The pass for chains longer than the registers hold (depth > 16).  A wavefront can issue one vector instruction
// every four to five cycles while its SIMD executes an fp64 multiply or add every 2.4, so the arithmetic of a long
// chain only reaches the SIMD's rate with three or four wavefronts resident, and k_update_slab's col' slices
// (4*depth registers per lane) leave room for two.  Here the col' slices of a workgroup's 128 rows sit in LDS
// (depth KiB), shared by its eight wavefronts, which deal the batches of kC columns among themselves; a wavefront
// walks the STEPS in its outer loop: col'_s of its rows is one LDS read, the kC row' values of the step one scalar
// load (the runs are stored batch by batch for that, deep_run_index), and the kC columns give 2*kC independent
// add chains per lane.  The entries of the next batch are requested before the current one is worked on.
// The |v| < 1e-6 drop is handled as in k_update_slab: evaluated without it while the smallest magnitude per column
// is tracked, and a column that saw one is redone from memory with the drop in place.
constexpr int kDeepThreads = 512;
template <int kC, bool kNT>
__global__ __launch_bounds__(kDeepThreads, kC == 4 ? 4 : 2) void k_update_deep(
    Tab T, int64_t lda, const double* __restrict__ colp, const PassInfo* __restrict__ passp,
    const int32_t* __restrict__ list_full, const int32_t* __restrict__ list_part,
    const double* __restrict__ rsl_full, const double* __restrict__ rsl_part, int depth,
    PivotState* __restrict__ st, ChainStep* __restrict__ steps, Shape sh,
    uint8_t* __restrict__ is_basic, int64_t* __restrict__ basis, int n_slabs, const uint64_t* __restrict__ ptr_full) {
  extern __shared__ double2 s_c[];   // [n_steps][kWave]
  if (st->status != ST_RUNNING) return;
  const PassInfo pass = *passp;
  const int n_steps = pass.n_steps;
  if (n_steps <= 0) return;
  if (blockIdx.x == 0 && threadIdx.x == 0) chain_bookkeeping(st, steps, pass, sh, is_basic, basis);
  const int slab = blockIdx.x % n_slabs, grp = blockIdx.x / n_slabs, n_grp = gridDim.x / n_slabs;
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int n_vec = (int)(lda / 2);
  const uint32_t v0 = (uint32_t)(slab * kWave + lane);
  const double2* __restrict__ cp = reinterpret_cast<const double2*>(colp);
  for (int q = threadIdx.x; q < n_steps * kWave; q += kDeepThreads)
    s_c[q] = cp[(int64_t)(q / kWave) * n_vec + slab * kWave + (q % kWave)];
  __syncthreads();
  constexpr int kWaves = kDeepThreads / kWave;
  const int stream = grp * kWaves + wave, n_streams = n_grp * kWaves;   // this wavefront: batches stream, stream + n_streams, ...
  const int n_batches = n_steps == depth ? pass.n_full / kC : 0;
  // The addresses of a batch's columns are one scalar load (ptr_full, left by k_chain_list) and the lane's offset is
  // added by the memory instruction itself.
  auto load_batch = [&](int bb, double2 (&dst)[kC]) {
    const uint64_t* __restrict__ pp = ptr_full + (int64_t)bb * kC;
#pragma unroll
    for (int u = 0; u < kC; u++) dst[u] = ld16g<kNT>(pp[u], v0);
  };
  // The chain on one batch held in t, then its stores.
  auto work = [&](int bb, double2 (&t)[kC]) {
    const double* __restrict__ rb = rsl_full + (int64_t)bb * depth * kC;
    float low[kC];
#pragma unroll
    for (int u = 0; u < kC; u++) low[u] = FLT_MAX;
    // The operands of a step (col'_s of the rows: one LDS read; the kC row' values: one scalar load) are requested
    // one step ahead into a second set of registers, and only AFTER the current step has laid hands on its own
    // set: scalar loads return out of order, so every wait for one is a wait for all outstanding, and
    // a request issued before that wait would be waited for on the spot.
    auto step = [&](const double2& cs, const double (&rj)[kC], double2& cs_ahead, double (&rj_ahead)[kC], int s_ahead) {
      asm volatile("" : : "v"(cs.x), "s"(rj[kC - 1]));   // the current set has arrived (a use the compiler cannot move)
      __builtin_amdgcn_sched_barrier(0);
      cs_ahead = s_c[s_ahead * kWave + lane];
#pragma unroll
      for (int u = 0; u < kC; u++) rj_ahead[u] = rb[s_ahead * kC + u];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < kC; u++) {
        const double px = __dmul_rn(cs.x, rj[u]), py = __dmul_rn(cs.y, rj[u]);
        low[u] = min_magnitude_high(low[u], t[u].x, t[u].y);   // what the step before left (a stored entry is 0 or large)
        t[u].x = __dadd_rn(t[u].x, px);
        t[u].y = __dadd_rn(t[u].y, py);
      }
    };
    double2 cs_a = s_c[lane], cs_b;
    double rj_a[kC], rj_b[kC];
#pragma unroll
    for (int u = 0; u < kC; u++) rj_a[u] = rb[u];
    int s = 0;
    for (; s + 2 <= depth; s += 2) {
      step(cs_a, rj_a, cs_b, rj_b, s + 1);
      step(cs_b, rj_b, cs_a, rj_a, s + 2 < depth ? s + 2 : s + 1);
    }
    if (s < depth) step(cs_a, rj_a, cs_b, rj_b, s);
    // Which columns saw a magnitude below the threshold: a scalar mask.  In the usual case (none) the stores follow
    // without a single load between them; a load under a condition in front of a store would make the compiler
    // wait for every outstanding memory operation, the store before included, at each of the kC stores.
    uint32_t redo = 0;
#pragma unroll
    for (int u = 0; u < kC; u++) {
      low[u] = min_magnitude_high(low[u], t[u].x, t[u].y);
      if (__any(__float_as_uint(low[u]) <= kDropHigh)) redo |= 1u << u;
    }
    const uint64_t* __restrict__ pp = ptr_full + (int64_t)bb * kC;
    if (redo == 0) {
#pragma unroll
      for (int u = 0; u < kC; u++) st16g<kNT>(pp[u], v0, t[u]);
    } else {
#pragma unroll 1
      for (int u = 0; u < kC; u++) {
        if ((redo >> u) & 1u) {                               // this column again, with the drop
          double2 w = ld16g<kNT>(pp[u], v0);                  // not stored yet
          for (int q = 0; q < depth; q++) w = rank1(w, s_c[q * kWave + lane], rb[q * kC + u]);
          st16g<kNT>(pp[u], v0, w);
        }
      }
#pragma unroll
      for (int u = 0; u < kC; u++)
        if (!((redo >> u) & 1u)) st16g<kNT>(pp[u], v0, t[u]);
    }
  };
  // Two register sets take turns (the loop is unrolled twice): while one is worked on, the loads of the next batch
  // fill the other.  The same loads every round (the last one fetches its own batch again, unused): a load under a
  // condition would make the compiler wait for all outstanding loads before the first use of the current batch.
  int b = stream;
  if (b < n_batches) {
    double2 ta[kC], tb[kC];
    load_batch(b, ta);
    for (;;) {
      load_batch(b + n_streams < n_batches ? b + n_streams : b, tb);
      work(b, ta);
      b += n_streams;
      if (b >= n_batches) break;
      load_batch(b + n_streams < n_batches ? b + n_streams : b, ta);
      work(b, tb);
      b += n_streams;
      if (b >= n_batches) break;
    }
  }
  // step by step: what is left of list_full (all of it when the chain ended early), then list_part
  const int n_list = pass.n_full + pass.n_part;
  for (int k = n_batches * kC + stream; k < n_list; k += n_streams) {
    const bool in_full = k < pass.n_full;
    const int l = in_full ? list_full[k] : list_part[k - pass.n_full];
    LPAMD_CHECK(13, l >= 0 && l <= sh.nloc && n_list <= sh.nloc + 1 && n_steps <= depth && (int64_t)v0 < lda / 2);
    double2* __restrict__ p = reinterpret_cast<double2*>(T.col(l)) + v0;
    double2 w = ld16<kNT>(p);
    for (int s = 0; s < n_steps; s++) {
      const double rj = in_full ? rsl_full[deep_run_index(k, s, depth, kC)] : rsl_part[(int64_t)(k - pass.n_full) * depth + s];
      if (rj != 0.0) w = rank1(w, s_c[s * kWave + lane], rj);
    }
    st16<kNT>(p, w);
  }
}

// ------------------------------------------------- multi-GPU exchange helpers
// Every rank folds the all-gathered candidate records identically (same rule as
// lpamd_fold_candidates on the host) and keeps the winner in its own `cand`, so
// no rank has to wait for the host to learn the entering column.
__global__ __launch_bounds__(kWave) void k_fold_candidates(
    const Candidate* __restrict__ recs, int world, int strategy_max_reduction, PivotState* __restrict__ st,
    Candidate* __restrict__ cand) {
  if (st->status != ST_RUNNING) return;
  if (threadIdx.x != 0) return;
  if (st->dirty == 1) { st->status = ST_DIRTY; return; }
  if (st->iter >= st->max_iter) { st->status = ST_LIMIT; return; }
  Candidate best;
  best.key = -DBL_MAX; best.gcol = -1; best.row = -1; best.ce = 0.0;
  bool any = false;
  for (int k = 0; k < world; k++) {
    const Candidate c = recs[k];
    if (c.gcol < 0) continue;
    any = true;
    if (strategy_max_reduction && c.row < 0) continue;
    if (best.gcol < 0 || c.key > best.key || (c.key == best.key && c.gcol < best.gcol)) best = c;
  }
  if (best.gcol < 0) { st->status = any ? ST_UNBOUNDED : ST_SOLVED; return; }
  *cand = best;
}

// The owner of the winning column writes it into the exchange buffer, every other
// rank writes zeros: an integer sum-all-reduce of the buffers (bit patterns added
// to zeros) then delivers the exact column to every rank without a data-dependent
// broadcast root.
__global__ __launch_bounds__(256) void k_export_or_zero(
    Tab T, Shape sh, const PivotState* __restrict__ st, const Candidate* __restrict__ cand,
    double* __restrict__ xbuf) {
  if (st->status != ST_RUNNING) return;
  const int64_t e = cand->gcol;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= sh.lda) return;
  double v = 0.0;
  if (e >= 0) {
    const int64_t eo = ordinal_of(e, sh.N);
    if (ord_rank(eo, sh.world) == sh.rank) v = T.col(ord_local(eo, sh.world))[i];
  }
  xbuf[i] = v;
}

// Look-ahead exchange (MAX_COST).  Once k_pivot_finalize has set a pivot up, the next
// entering candidate of this rank is known (pricing partials on the updated objective
// row) although the rank-1 update has not run yet.  This kernel writes that column AS
// IT WILL BE after the pending update, T[:, l] (+) col' * row'[l] with the update's own
// arithmetic, into `xlocal` without touching the tableau, so the exchange of the next
// pivot column can travel while the update streams the shard.  pending == 0: no pivot
// is waiting for its update (first iteration of a run), the column is copied as it is.
// pending > 1: a chain of pivots has been set up (k_chain_finalize); the steps that really exist
// (done flags) are applied in their order.
// The local candidate is folded here from the pricing partials the last set-up left (every
// workgroup folds them itself, workgroup 0 writes the 32-byte record), which saves the launch of
// k_price_fold in front of every exchange.
// kRatio: the workgroup also leaves the ratio-test partial of its rows (reference lp_simplex_tableau.cc:207-227: smallest
// b_i / (-a_i) over (float)a_i <= -1e-6f, lowest row on ties) on that column and the right-hand side after the same steps, in
// rpart_v / rpart_i behind the column, so that whoever receives the slot can fold the pivot row without a ratio-test
// launch of its own; and a used-up budget travels as "no candidate" (the kernel that consumes the slots advances the
// pivot count itself and cannot test it).
template <bool kRatio>
__global__ __launch_bounds__(kPartialThreads) void k_export_lookahead(
    Tab T, Shape sh, const PivotState* __restrict__ st, const double* __restrict__ pv, const int64_t* __restrict__ pi, int n_price_part,
    Candidate* __restrict__ cand_local, const double* __restrict__ colp, const double* __restrict__ rowp, int64_t rstride,
    const ChainStep* __restrict__ steps, int pending, double* __restrict__ xlocal,
    const double* __restrict__ rhsv, double* __restrict__ rpart_v, int64_t* __restrict__ rpart_i) {
  __shared__ double s_v[kPartialThreads / kWave];
  __shared__ int64_t s_i[kPartialThreads / kWave];
  // requested before the first wait: the gating state, the step records, this thread's share of the pricing partials
  // and its entry of the right-hand side (a round trip each if they were asked for one after the other)
  const int32_t status = st->status;
  const int64_t iter = st->iter, max_iter = st->max_iter;
  const int n = pending > 0 ? chain_steps_done(steps, pending) : 0;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const double* __restrict__ rhs = n == 0 ? T.col(sh.nloc) : rhsv + (int64_t)(n - 1) * sh.lda;
  const double rhs_i = (kRatio && i < sh.m) ? rhs[i] : 0.0;
  double ce = -DBL_MAX;
  int64_t l = -1;
  fold_share<true, kPartialThreads>(pv, pi, n_price_part, ce, l);
  if (status != ST_RUNNING) return;
  block_arg<true, kPartialThreads>(ce, l, s_v, s_i);
  if (kRatio && iter >= max_iter) l = -1;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    Candidate c;
    c.key = ce; c.ce = ce; c.row = -1;
    c.gcol = l < 0 ? -1 : gcol_of(l * sh.world + sh.rank, sh.N);
    *cand_local = c;
  }
  if (!kRatio && i >= sh.lda) return;
  double v = 0.0;
  if (l >= 0 && i < sh.m) v = chain_apply(T.col(l)[i], n, colp + i, sh.lda, rowp + l, rstride);   // the update leaves columns with row' == 0 alone
  if (i < sh.lda) xlocal[i] = v;
  if (kRatio) {
    double bv = DBL_MAX;
    int64_t bi = -1;
    if (l >= 0 && i < sh.m && (float)v <= -1e-6f) { bv = rhs_i / (-v); bi = i; }
    block_arg<false, kPartialThreads>(bv, bi, s_v, s_i);
    if (threadIdx.x == 0) { rpart_v[blockIdx.x] = bv; rpart_i[blockIdx.x] = bi; }
  }
}

// Runs on the exchange stream, next to the update: folds the gathered records (largest
// reduced cost, lowest global column on ties; reference lp_simplex_tableau.cc:188-198),
// keeps the winner in `cand_la` and fills the exchange buffer with this rank's
// look-ahead column if it owns the winner, with zeros otherwise (the integer
// sum-all-reduce that follows hands every rank the exact column).  It reads nothing the
// update writes and leaves the pivot state alone: "no candidate" travels as gcol = -1
// and turns into SOLVED in the ratio-test kernel of the next pivot.
__global__ __launch_bounds__(256) void k_lookahead_select(
    const Candidate* __restrict__ recs, int world, Shape sh, const double* __restrict__ xlocal,
    Candidate* __restrict__ cand_la, double* __restrict__ xbuf) {
  __shared__ Candidate s_best;
  if (threadIdx.x == 0) {
    Candidate best;
    best.key = -DBL_MAX; best.gcol = -1; best.row = -1; best.ce = 0.0;
    for (int k = 0; k < world; k++) {
      const Candidate c = recs[k];
      if (c.gcol < 0) continue;
      if (best.gcol < 0 || c.key > best.key || (c.key == best.key && c.gcol < best.gcol)) best = c;
    }
    s_best = best;
    if (blockIdx.x == 0) *cand_la = best;
  }
  __syncthreads();
  const int64_t e = s_best.gcol;
  const bool mine = e >= 0 && ord_rank(ordinal_of(e, sh.N), sh.world) == sh.rank;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < sh.lda) xbuf[i] = mine ? xlocal[i] : 0.0;
}

// --------------------------------------------------------------- utilities
__global__ void k_gather_row(Tab T, int32_t ncl, int64_t r, double* __restrict__ out) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x;
  if (l < ncl) out[l] = T.col(l)[r];
}

__global__ void k_fill(double* __restrict__ p, int64_t n, double v) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// GetTableauSimplexSolution, per-column part (reference lp_simplex_tableau.cc:388-399):
// for a basic column the LAST row with a non-zero entry defines x = -b_i / T_ij.
__global__ __launch_bounds__(256) void k_column_values(
    Tab T, const uint8_t* __restrict__ is_basic, Shape sh, double* __restrict__ x) {
  __shared__ double s_v[256 / kWave];
  __shared__ int64_t s_i[256 / kWave];
  const int l = blockIdx.x;
  if (!is_basic[l]) { if (threadIdx.x == 0) x[l] = 0.0; return; }
  const double* __restrict__ col = T.col(l);
  double bv = 0.0;
  int64_t bi = -1;
  for (int64_t i = threadIdx.x; i < sh.m; i += 256)
    if (!is_zero(col[i])) { bv = (double)i; bi = i; }
  block_arg<true, 256>(bv, bi, s_v, s_i);
  if (threadIdx.x == 0) {
    const double* rhs = T.col(sh.nloc);
    x[l] = bi < 0 ? 0.0 : -rhs[bi] / col[bi];
  }
}

// ---- synthetic dense LP: entry (i, j) is a pure function of (seed, i, j) ----
__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__host__ __device__ __forceinline__ double u01(uint64_t seed, uint64_t i, uint64_t j) {
  const uint64_t h = splitmix64(splitmix64(seed ^ (i * 0xD1B54A32D192ED03ull)) ^ (j * 0x8CB92BA72F3D8DD7ull));
  return (double)(h >> 11) * 0x1.0p-53;
}
// max c.x  s.t.  A x <= b, x >= 0 in the reference's slack-form convention
// (row i reads 0 = b_i + sum_j T_ij x_j, reference lp.cc:135-145): structural
// column j < n: -A_ij, A ~ U(0.1, 1.1); slack column n+i: -1; RHS: b_i ~ U(1,2)*n;
// objective c_j ~ U(0.5, 1.5).  Row index m addresses the objective row.
__host__ __device__ __forceinline__ double synth_entry(uint64_t seed, int64_t m, int64_t n, int64_t i, int64_t j) {
  const int64_t N = n + m;
  if (i == m) return j < n ? 0.5 + u01(seed, (uint64_t)m, (uint64_t)j) : 0.0;
  if (j < n) return -(0.1 + u01(seed, (uint64_t)i, (uint64_t)j));
  if (j < N) return (j - n == i) ? -1.0 : 0.0;
  if (j == N) return (1.0 + u01(seed, (uint64_t)i, (uint64_t)N)) * (double)n;
  return 0.0;
}

__global__ __launch_bounds__(256) void k_fill_synthetic(
    Tab T, double* __restrict__ obj, uint8_t* __restrict__ is_basic, int64_t* __restrict__ basis,
    Shape sh, int64_t n, uint64_t seed) {
  const int l = blockIdx.x;                       // local column, nloc = RHS replica
  const int64_t g = l < sh.nloc ? gcol_of((int64_t)l * sh.world + sh.rank, sh.N) : sh.N;
  const int64_t i = (int64_t)blockIdx.y * blockDim.x + threadIdx.x;
  if (i < sh.lda) T.col(l)[i] = i < sh.m ? synth_entry(seed, sh.m, n, i, g) : 0.0;
  if (i == 0) {
    obj[l] = synth_entry(seed, sh.m, n, sh.m, g);
    is_basic[l] = (l < sh.nloc && g >= n && g < n + sh.m) ? 1 : 0;
  }
  if (l == 0 && i < sh.m) basis[i] = n + i;
}

}  // namespace lpamd
