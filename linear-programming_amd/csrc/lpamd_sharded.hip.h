// Column shards over several GPUs behind the C ABI (lpamd_group_* in include/lpamd.h).
//
// A group is one LP whose tableau columns are dealt round-robin over `world` shards, one
// lpamd_engine per shard (see lpamd_engine.hip).  Two ways to run it:
//   - one process drives several GPUs: one host thread and one RCCL communicator per shard
//     (ncclCommInitAll); this is what `solver --gpus N` and LPModel use;
//   - one process per GPU (torch.distributed.run, MPI, ...): every process holds one shard
//     and joins the communicator with an id that the caller delivers (ncclCommInitRank).
// RCCL is loaded on first use (dlopen), so single-GPU users never pay for it.
//
// The pivot loop needs ONE collective per pivot: every rank packs its best local candidate,
// the 32-byte record and the candidate COLUMN itself (as the pivots set up so far will leave
// it), into its slot; an all-gather hands every rank all slots; every rank folds the records
// with the reference's tie-break (reference lp_simplex_tableau.cc:188-198 / 310-325: larger
// key, lowest global column on ties) and reads the winner's column from the winner's slot.
// Sending a column that may lose costs bandwidth (world * 8 * m bytes per pivot, 2 MiB at
// m = 32768 over 8 ranks) and saves the second, dependent collective; both are latency-bound.
// Everything else (row gather, rank-1 update) is rank-local and bit-identical across ranks.
// Inside a chain of MAX_COST pivots the owner of a candidate also takes the ratio-test partials
// on its column (the right-hand side is replicated) and sends them behind the column, and the
// set-up kernel of every rank folds records and partials itself: two launches and the
// collective per pivot (fused_step; MAX_REDUCTION records bring their rows instead).
//
// Where the ranks can map each other's memory the collective itself goes (link, ShardLink in
// lpamd_shardstep_kernels.hip.h): the launch that sets a step up stores this rank's slot for the next
// step straight into every rank's receive buffer as tagged words, which the next launch polls, and
// the other all-gathers of a slot's size or less are one launch over the same buffers
// (k_link_all_gather).  RCCL then forms the group, trades the buffers' handles, carries the long
// vectors of the feasibility phase, and takes everything back if the self-test of the mapping fails.
//
// LPAMD_GROUP_LOCAL_EXCHANGE replaces RCCL by device copies between the shards of one process
// (events + host barriers).  It exists so that several shards can share ONE GPU, which RCCL
// refuses, for tests on a one-GPU machine; it also works across GPUs with peer access.
#pragma once

#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only; the symbols are resolved with dlsym

#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>

namespace lpamd {

constexpr int kSlotHeaderWords = 8;   // 64 bytes: the candidate record (32 B) and padding

// In-order loop (MAX_REDUCTION): the local candidate record and its column as stored.
__global__ __launch_bounds__(256) void k_export_candidate(
    Tab T, Shape sh, const PivotState* __restrict__ st, const Candidate* __restrict__ cand, uint64_t* __restrict__ slot) {
  if (st->status != ST_RUNNING) return;
  const Candidate c = *cand;
  if (blockIdx.x == 0 && threadIdx.x == 0) *reinterpret_cast<Candidate*>(slot) = c;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= sh.lda) return;
  double v = 0.0;
  if (c.gcol >= 0) {
    const int64_t eo = ordinal_of(c.gcol, sh.N);
    if (ord_rank(eo, sh.world) == sh.rank) v = T.col(ord_local(eo, sh.world))[i];
  }
  reinterpret_cast<double*>(slot + kSlotHeaderWords)[i] = v;
}

// Fold of the gathered slots.  kInOrder: the fold of the in-order loop, which also settles the
// status (budget, optimum, unbounded; reference lp_simplex_tableau.cc:201-205, 230-233, 365-373)
// and keeps the winner in the engine's own candidate.  Otherwise (look-ahead loop): runs next to
// the rank-1 update, touches no pivot state; "no candidate" travels as gcol = -1.
// chain_steps / chain_j (look-ahead form only): a MAX_REDUCTION winner of step chain_j >= 1 of a pass is
// recorded for k_chain_finalize here, after the exchange, if that step may still be set up.
template <bool kInOrder>
__global__ __launch_bounds__(256) void k_pick_winner(
    const uint64_t* __restrict__ slots, int world, int64_t slot_words, Shape sh, int max_reduction,
    PivotState* __restrict__ st, Candidate* __restrict__ cand_out, double* __restrict__ xbuf,
    ChainStep* __restrict__ chain_steps, int chain_j) {
  __shared__ int s_w;
  if (kInOrder && st->status != ST_RUNNING) return;
  if (threadIdx.x == 0) {
    Candidate best;
    best.key = -DBL_MAX; best.gcol = -1; best.row = -1; best.ce = 0.0;
    int w = -1;
    bool any = false;
    for (int k = 0; k < world; k++) {
      const Candidate c = *reinterpret_cast<const Candidate*>(slots + (size_t)k * slot_words);
      if (c.gcol < 0) continue;
      any = true;
      if (max_reduction && c.row < 0) continue;
      if (best.gcol < 0 || c.key > best.key || (c.key == best.key && c.gcol < best.gcol)) { best = c; w = k; }
    }
    s_w = w;
    if (blockIdx.x == 0) {
      if (kInOrder) {
        if (st->dirty == 1) st->status = ST_DIRTY;
        else if (st->iter >= st->max_iter) st->status = ST_LIMIT;
        else if (w < 0) st->status = any ? ST_UNBOUNDED : ST_SOLVED;
        else *cand_out = best;
      } else {
        *cand_out = best;
        if (chain_steps && chain_j >= 1 && w >= 0 && best.row >= 0 && prb_step_alive(st, chain_steps, chain_j)) {
          chain_steps[chain_j].e = best.gcol;
          chain_steps[chain_j].ce = best.ce;
        }
      }
    }
  }
  __syncthreads();
  const int w = s_w;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < sh.lda) xbuf[i] = w >= 0 ? reinterpret_cast<const double*>(slots + (size_t)w * slot_words + kSlotHeaderWords)[i] : 0.0;
}

// out[i] = sum over ranks of stage[k][i] as 64-bit integers: with at most one non-zero
// contribution per word this hands every rank the exact bit pattern.
__global__ __launch_bounds__(256) void k_sum_words(const uint64_t* __restrict__ stage, int world, int64_t words, uint64_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= words) return;
  uint64_t s = 0;
  for (int k = 0; k < world; k++) s += stage[(size_t)k * words + i];
  out[i] = s;
}

// All-gather of up to `cap` 64-bit words per rank over the ranks' receive buffers, in ONE launch: thread i stores word i of
// this rank's block into every rank's gather area as a pair of tagged words, then polls word i of every rank's block in its
// own area (see ShardLink for the tags; `at` = start of the gather area, three sets of `world` blocks of 2 * cap tagged words
// that take turns).  What a latency-bound collective of a few hundred KiB costs over xGMI is then one store and one poll.
__global__ __launch_bounds__(256) void k_link_all_gather(
    const uint64_t* __restrict__ send, uint64_t* __restrict__ recv, int64_t words, int world, int my_rank, uint64_t* const* __restrict__ peers,
    const uint64_t* __restrict__ in, int64_t at, int64_t cap, uint32_t tag, long long timeout, PivotState* __restrict__ st) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= words) return;
  const int64_t set = at + (int64_t)(tag % kLinkBuffers) * world * 2 * cap;
  const uint64_t v = send[i];
  for (int p = 0; p < world; p++) ll_put(peers[p] + set + ((int64_t)my_rank * cap + i) * 2, v, tag);
  // a rank whose run has failed (a word that never came) gives the others its words and waits for nobody any more: the
  // launches its host has already enqueued drain at once instead of one time limit each
  if (st->status == ST_ERROR) {
    for (int k = 0; k < world; k++) recv[(int64_t)k * words + i] = k == my_rank ? v : 0;
    return;
  }
  bool late = false;
  for (int k0 = 0; k0 < world; k0 += 4) {      // four ranks' words per round trip
    uint64_t w[4] = {0, 0, 0, 0};
    const uint64_t* p[4];
    uint32_t want = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int k = k0 + q < world ? k0 + q : k0;
      p[q] = in + set + ((int64_t)k * cap + i) * 2;
      if (k0 + q < world) want |= 1u << q;
    }
    const uint64_t* const pc[4] = {p[0], p[1], p[2], p[3]};
    late |= !ll_get_many<4>(pc, tag, w, want, timeout ? timeout : kLinkTimeoutTicks);
#pragma unroll
    for (int q = 0; q < 4; q++)
      if (k0 + q < world) recv[(int64_t)(k0 + q) * words + i] = w[q];
  }
  if (late) st->status = ST_ERROR;
}

// Self-test of the in-launch exchange (ShardLink) when a group is formed: every rank stores a tagged token into every
// rank's hello words, then looks for every rank's token in its own.
__global__ __launch_bounds__(kWave) void k_link_hello_put(uint64_t* const* __restrict__ peers, int world, int64_t hello_at, int my_rank) {
  if ((int)threadIdx.x < world) ll_put(peers[threadIdx.x] + hello_at + 2 * my_rank, 0xC0FFEE00ull + (uint64_t)my_rank, 1u);
}
__global__ __launch_bounds__(kWave) void k_link_hello_get(const uint64_t* __restrict__ in, int world, int64_t hello_at, int32_t* __restrict__ ok) {
  bool good = true;
  if ((int)threadIdx.x < world) {
    uint64_t v = 0;
    good = ll_get(in + hello_at + 2 * threadIdx.x, 1u, v, 2LL * 100000000LL) && v == 0xC0FFEE00ull + (uint64_t)threadIdx.x;
  }
  const unsigned long long bad = __ballot(!good);
  if (threadIdx.x == 0) *ok = bad == 0 ? 1 : 0;
}

}  // namespace lpamd

namespace {

// ---- RCCL, resolved on first use -------------------------------------------------------
struct RcclApi {
  void* so = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommAbort) CommAbort = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};
std::mutex g_rccl_mutex;
RcclApi g_rccl;

int rccl_api(RcclApi** out) {
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  if (!g_rccl.so) {
    // a copy that the process already holds (e.g. the one PyTorch ships) is found by its soname
    const char* names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    void* so = nullptr;
    for (const char* n : names)
      if ((so = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!so) return set_err(LPAMD_E_UNSUPPORTED, "RCCL is not available: %s", dlerror());
    RcclApi a;
    a.so = so;
#define LPAMD_RCCL_SYM(field, name)                                                       \
  a.field = reinterpret_cast<decltype(a.field)>(dlsym(so, name));                         \
  if (!a.field) return set_err(LPAMD_E_UNSUPPORTED, "RCCL symbol %s is missing", name)
    LPAMD_RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
    LPAMD_RCCL_SYM(CommInitRank, "ncclCommInitRank");
    LPAMD_RCCL_SYM(CommInitAll, "ncclCommInitAll");
    LPAMD_RCCL_SYM(CommDestroy, "ncclCommDestroy");
    LPAMD_RCCL_SYM(CommAbort, "ncclCommAbort");
    LPAMD_RCCL_SYM(CommCount, "ncclCommCount");
    LPAMD_RCCL_SYM(AllGather, "ncclAllGather");
    LPAMD_RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef LPAMD_RCCL_SYM
    g_rccl = a;
  }
  *out = &g_rccl;
  return LPAMD_OK;
}

#define RCCL_TRY(api, expr)                                                                              \
  do {                                                                                                   \
    ncclResult_t _r = (expr);                                                                            \
    if (_r != ncclSuccess) return set_err(LPAMD_E_HIP, "%s failed: %s", #expr, (api)->GetErrorString(_r)); \
  } while (0)

// ---- host barrier of the in-process exchange -------------------------------------------
class HostBarrier {
 public:
  void reset(int n) { n_ = n; waiting_ = 0; phase_ = 0; broken_ = false; }
  // false: another shard failed and the barrier was broken
  bool wait() {
    std::unique_lock<std::mutex> lock(mu_);
    if (broken_) return false;
    const uint64_t phase = phase_;
    if (++waiting_ == n_) {
      waiting_ = 0;
      phase_++;
      cv_.notify_all();
      return true;
    }
    cv_.wait(lock, [&] { return phase_ != phase || broken_; });
    return !broken_;
  }
  void abort() {
    std::lock_guard<std::mutex> lock(mu_);
    broken_ = true;
    cv_.notify_all();
  }
 private:
  std::mutex mu_;
  std::condition_variable cv_;
  int n_ = 1, waiting_ = 0;
  uint64_t phase_ = 0;
  bool broken_ = false;
};

}  // namespace

struct lpamd_shard {   // one column shard of a group and its exchange state
  lpamd_engine* eng = nullptr;
  int device = 0;
  int rank = 0;
  hipStream_t xstream = nullptr;             // exchange stream: the collective of the next pass runs beside the update
  hipEvent_t ev_sent = nullptr, ev_arrived = nullptr;
  hipEvent_t ev_pub = nullptr, ev_read = nullptr;   // in-process exchange
  ncclComm_t comm = nullptr;
  bool comm_aborted = false;                 // a shard of this process failed: the communicator was torn down with ncclCommAbort
  uint64_t* slot = nullptr;                  // slot_words words (record, column, ratio-test partials): what this rank sends
  uint64_t* slots = nullptr;                 // world slots: what every rank receives
  double* xbuf = nullptr;                    // lda doubles: the entering column of the pivot being set up
  uint64_t* send = nullptr;                  // assembly of rows / columns / objective: max(N + 2, lda) words ...
  uint64_t* stage = nullptr;                 // ... world blocks of them ...
  uint64_t* summed = nullptr;                // ... and their exact sum
  // exchange inside the chained step's launch (ShardLink, lpamd_shardstep_kernels.hip.h)
  uint64_t* link_buf = nullptr;              // kLinkBuffers x world slots of tagged words, then the hello words: what the ranks store into
  uint64_t** link_peers = nullptr;           // device array [world]: every rank's link_buf as this process maps it
  int32_t* link_ok = nullptr;                // result of the self-test
  std::vector<void*> link_mapped;            // peers' buffers opened through IPC handles
  uint32_t link_tag = 0;                     // tag of the last slot exported that way
  int64_t link_exports = 0;                  // slots exported that way so far
  uint32_t gather_tag = 0;                   // tag of the last all-gather that went over the receive buffers
  int rc = LPAMD_OK;
  std::string err;
};

struct lpamd_group {
  int64_t m = 0, N = 0, lda = 0, slot_words = 0, wide = 0;
  int world = 1;                  // ranks of the exchange
  int engine_world = 1;           // shards the columns are dealt over (differs from world only in a rehearsal)
  bool local_exchange = false;
  bool inline_exchange = true;    // collectives inside a chain run on the engine's stream (nothing to overlap them with)
  bool fused_step = true;         // MAX_COST steps inside a chain: ratio-test partials travel with the slot, two launches per step
  lpamd_host_allgather_fn host_gather = nullptr;   // the collective, when the caller supplies it (lpamd_group_create_rank_hosted)
  void* host_ctx = nullptr;
  bool link = false;              // MAX_COST steps of a chain: the slots travel inside the launches, no collective
  int64_t ll_words = 0;           // tagged words per slot
  bool link_gather = false;       // all-gathers of up to gather_cap words go over the receive buffers too (k_link_all_gather)
  int64_t gather_cap = 0;
  long long link_timeout = 0;     // ticks a reader waits for a word (0: the kernels' default of ten seconds; LPAMD_LINK_TIMEOUT_MS)
  int64_t link_fault_at = 0;      // experiments only (LPAMD_LINK_FAULT=n): rank 0 mis-tags its n-th slot, which the other ranks then never see
  int chain = 1;                  // pivots per pass of the look-ahead loop
  int chain_mr = 1;               // ... and of the MAX_REDUCTION loop
  int poll_every = 256;
  std::vector<lpamd_shard> shards;           // the shards this process holds
  std::vector<const void*> pub;              // in-process exchange: what every rank offers
  HostBarrier barrier;
  std::mutex abort_mu;
};

namespace {

using Group = lpamd_group;
using Shard = lpamd_shard;

// A shard of this process has failed: the collectives the others have enqueued would wait for it for ever (and with
// them every hipStreamSynchronize behind them), so every communicator this process holds is aborted.  The group is
// unusable afterwards; lpamd_group_destroy skips the aborted communicators.
void abort_comms(lpamd_group* g);

// Runs fn on every local shard, one host thread per shard when there are several (the loops
// are collective: all shards must be inside them at the same time).  The first failure wins.
template <typename Fn>
int for_each_shard(Group* g, Fn fn) {
  if (g->shards.size() == 1) {
    Shard& s = g->shards[0];
    HIP_TRY(hipSetDevice(s.device));
    const int rc = fn(s);
    if (rc != LPAMD_OK) {   // one process per GPU: the peers' collectives would wait for this rank for ever
      char keep[sizeof(g_err)];
      std::memcpy(keep, g_err, sizeof(keep));
      abort_comms(g);
      std::memcpy(g_err, keep, sizeof(keep));
    }
    return rc;
  }
  g->barrier.reset((int)g->shards.size());
  std::vector<std::thread> threads;
  for (Shard& s : g->shards) {
    s.rc = LPAMD_OK;
    s.err.clear();
    threads.emplace_back([g, &s, &fn] {
      if (hipSetDevice(s.device) != hipSuccess) s.rc = set_err(LPAMD_E_HIP, "hipSetDevice(%d) failed", s.device);
      else s.rc = fn(s);
      if (s.rc != LPAMD_OK) {
        s.err = g_err;            // g_err is per thread
        g->barrier.abort();       // nobody waits for a shard that gave up
        abort_comms(g);           // ... nor inside a collective whose peer will never arrive
      }
    });
  }
  for (std::thread& t : threads) t.join();
  for (Shard& s : g->shards)
    if (s.rc != LPAMD_OK) return set_err(s.rc, "shard %d: %s", s.rank, s.err.c_str());
  return LPAMD_OK;
}

void abort_comms(Group* g) {
  if (g->local_exchange) return;
  RcclApi* api = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (!g_rccl.so) return;
    api = &g_rccl;
  }
  std::lock_guard<std::mutex> lock(g->abort_mu);
  for (Shard& q : g->shards)
    if (q.comm && !q.comm_aborted) { q.comm_aborted = true; (void)api->CommAbort(q.comm); }
}

// All-gather of `words` 64-bit words per rank on `stream` (send and recv are device memory).
int64_t link_gather_at(const Group* g);

int all_gather_words(Group* g, Shard& s, const uint64_t* send, uint64_t* recv, int64_t words, hipStream_t stream) {
  if (g->link_gather && words <= g->gather_cap) {   // over the ranks' receive buffers: one launch, no collective
    s.gather_tag = s.gather_tag + 1 == 0 ? 1 : s.gather_tag + 1;
    hipLaunchKernelGGL(k_link_all_gather, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, stream, send, recv, words, g->world, s.rank,
                       s.link_peers, s.link_buf, link_gather_at(g), g->gather_cap, s.gather_tag, g->link_timeout, s.eng->st);
    return check_launch("k_link_all_gather");
  }
  if (g->host_gather) {   // the caller's collective, staged through host memory; synchronous
    std::vector<uint64_t> mine((size_t)words), every((size_t)words * g->world);
    HIP_TRY(hipMemcpyAsync(mine.data(), send, sizeof(uint64_t) * words, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (g->host_gather(g->host_ctx, mine.data(), every.data(), (int64_t)sizeof(uint64_t) * words) != 0)
      return set_err(LPAMD_E_STATE, "the caller's all-gather failed");
    HIP_TRY(hipMemcpyAsync(recv, every.data(), sizeof(uint64_t) * words * g->world, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return LPAMD_OK;
  }
  if (!g->local_exchange) {
    RcclApi* api = nullptr;
    LP_TRY(rccl_api(&api));
    if (s.comm_aborted) return set_err(LPAMD_E_STATE, "another shard failed");
    RCCL_TRY(api, api->AllGather(send, recv, (size_t)words, ncclUint64, s.comm, stream));
    return LPAMD_OK;
  }
  // In-process exchange: publish, wait until everybody has published, copy every rank's block,
  // wait until everybody has copied (the caller may overwrite `send` afterwards).
  g->pub[s.rank] = send;
  HIP_TRY(hipEventRecord(s.ev_pub, stream));
  if (g->shards.size() > 1 && !g->barrier.wait()) return set_err(LPAMD_E_STATE, "another shard failed");
  for (Shard& q : g->shards) {
    HIP_TRY(hipStreamWaitEvent(stream, q.ev_pub, 0));
    HIP_TRY(hipMemcpyAsync(recv + (size_t)q.rank * words, g->pub[q.rank], sizeof(uint64_t) * words, hipMemcpyDeviceToDevice, stream));
  }
  HIP_TRY(hipEventRecord(s.ev_read, stream));
  if (g->shards.size() > 1 && !g->barrier.wait()) return set_err(LPAMD_E_STATE, "another shard failed");
  for (Shard& q : g->shards) HIP_TRY(hipStreamWaitEvent(stream, q.ev_read, 0));
  return LPAMD_OK;
}

// Shards of one process that share a GPU and trade inside their launches (ShardLink): a launch that waits for another
// shard's launch must not sit in front of it in a hardware queue the two streams happen to share (the runtime deals its
// streams over a few queues, and a queue starts its packets in order), so every shard's stream first waits for what the
// other shards have enqueued so far.  The slots still travel by the launches' own stores and are read by polling.
int link_fence_local(Group* g, Shard& s, hipStream_t stream) {
  if (!g->local_exchange || g->shards.size() < 2) return LPAMD_OK;
  HIP_TRY(hipEventRecord(s.ev_pub, stream));
  if (!g->barrier.wait()) return set_err(LPAMD_E_STATE, "another shard failed");
  for (Shard& q : g->shards)
    if (&q != &s) HIP_TRY(hipStreamWaitEvent(stream, q.ev_pub, 0));
  if (!g->barrier.wait()) return set_err(LPAMD_E_STATE, "another shard failed");
  return LPAMD_OK;
}

// Every rank gets the element-wise exact sum of the ranks' host vectors (n doubles; at most
// one rank contributes a non-zero bit pattern per entry).  Synchronises.
int sum_exact(Group* g, Shard& s, double* vec, int64_t n) {
  hipStream_t st = s.eng->stream;
  HIP_TRY(hipMemcpyAsync(s.send, vec, sizeof(double) * n, hipMemcpyHostToDevice, st));
  LP_TRY(all_gather_words(g, s, s.send, s.stage, n, st));
  hipLaunchKernelGGL(k_sum_words, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, s.stage, g->world, n, s.summed);
  LP_TRY(check_launch("k_sum_words"));
  HIP_TRY(hipMemcpyAsync(vec, s.summed, sizeof(double) * n, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return LPAMD_OK;
}

// Full N + 2 vector from per-rank vectors that hold the local columns only; the replicated
// right-hand-side entry is taken from rank 0.
int assemble(Group* g, Shard& s, std::vector<double>& v) {
  if (s.rank != 0) v[g->N] = 0.0;
  return sum_exact(g, s, v.data(), g->N + 2);
}
int assembled_row(Group* g, Shard& s, int64_t r, std::vector<double>& row) {
  std::fill(row.begin(), row.end(), 0.0);
  LP_TRY(lpamd_read_row(s.eng, r, row.data()));
  return assemble(g, s, row);
}
int assembled_objective(Group* g, Shard& s, std::vector<double>& obj) {
  std::fill(obj.begin(), obj.end(), 0.0);
  LP_TRY(lpamd_download(s.eng, nullptr, 0, obj.data(), nullptr));
  return assemble(g, s, obj);
}
int assembled_basic(Group* g, Shard& s, std::vector<uint8_t>& basic) {
  std::vector<uint8_t> b(g->N + 2, 0);
  LP_TRY(lpamd_download(s.eng, nullptr, 0, nullptr, b.data()));
  std::vector<double> v(g->N + 2, 0.0);
  for (int64_t j = 0; j < g->N + 2; j++) v[j] = (j != g->N && owns(s.eng, j) && b[j]) ? 1.0 : 0.0;
  LP_TRY(sum_exact(g, s, v.data(), g->N + 2));
  for (int64_t j = 0; j < g->N + 2; j++) basic[j] = v[j] != 0.0 ? 1 : 0;
  return LPAMD_OK;
}
// Column `col` on every rank (read by its owner).
int assembled_column(Group* g, Shard& s, int64_t col, std::vector<double>& out) {
  if (col == g->N) return lpamd_read_column(s.eng, col, out.data());
  std::fill(out.begin(), out.end(), 0.0);
  if (owns(s.eng, col)) LP_TRY(lpamd_read_column(s.eng, col, out.data()));
  return sum_exact(g, s, out.data(), g->m);
}

// LPModel::TableauPivot(d -> out, e -> in, row r) on every shard (reference lp_simplex_tableau.cc:10-52):
// the owner of column e offers it, everybody else zeros; the exact sum lands in xbuf.
int pivot_shards(Group* g, Shard& s, int64_t d, int64_t e, int64_t r, double ce) {
  hipStream_t st = s.eng->stream;
  if (owns(s.eng, e)) HIP_TRY(hipMemcpyAsync(s.send, s.eng->T.col(local_of(s.eng, e)), sizeof(double) * g->lda, hipMemcpyDeviceToDevice, st));
  else HIP_TRY(hipMemsetAsync(s.send, 0, sizeof(double) * g->lda, st));
  LP_TRY(all_gather_words(g, s, s.send, s.stage, g->lda, st));
  hipLaunchKernelGGL(k_sum_words, dim3((unsigned)((g->lda + 255) / 256)), dim3(256), 0, st, s.stage, g->world, g->lda,
                     reinterpret_cast<uint64_t*>(s.xbuf));
  LP_TRY(check_launch("k_sum_words"));
  return lpamd_pivot_shard(s.eng, d, e, r, ce, s.xbuf);
}

// ---- the pivot loops ----------------------------------------------------------------------
// Exchange for the NEXT pivot: gather the slots, keep the winner, deliver its column.  on_x: on
// the exchange stream, ordered after what the engine's stream holds so far (the export) and beside
// what it gets next (the update); otherwise in line on the engine's stream.
int exchange_lookahead(Group* g, Shard& s, bool on_x) {
  lpamd_engine* e = s.eng;
  hipStream_t st = on_x ? s.xstream : e->stream;
  if (on_x) {
    HIP_TRY(hipEventRecord(s.ev_sent, e->stream));
    HIP_TRY(hipStreamWaitEvent(s.xstream, s.ev_sent, 0));
  }
  LP_TRY(all_gather_words(g, s, s.slot, s.slots, g->slot_words, st));
  hipLaunchKernelGGL(k_pick_winner<false>, dim3((unsigned)((g->lda + 255) / 256)), dim3(256), 0, st, s.slots, g->world, g->slot_words,
                     e->sh, 0, e->st, e->cand_la, s.xbuf, nullptr, 0);
  LP_TRY(check_launch("k_pick_winner"));
  if (on_x) HIP_TRY(hipEventRecord(s.ev_arrived, s.xstream));
  return LPAMD_OK;
}

int export_lookahead(Shard& s) {
  return lpamd_lookahead_export(s.eng, s.slot, s.slot + kSlotHeaderWords);
}

// The tag the next exported slot carries (s.link_tag, which every rank expects), or a wrong one when a fault was asked for.
uint32_t link_tag_out(Group* g, Shard& s) {
  s.link_exports++;
  if (g->link_fault_at > 0 && s.rank == 0 && s.link_exports == g->link_fault_at) return s.link_tag ^ 0x40000000u;
  return s.link_tag;
}

// With the exchange inside the launches a rank can end up in ST_ERROR alone (a slot that did not arrive in time), and the
// ranks would then leave the loop at different reads of the state, one of them with collectives enqueued that the other
// never joins.  So the ranks of a communicator trade what they have read and all stop as soon as one has stopped.
int agree_on_result(Group* g, Shard& s, int32_t* res) {
  if ((!g->link && !g->link_gather) || g->local_exchange) return LPAMD_OK;   // (a world of one trades with itself: the rehearsal walks this code)
  hipStream_t st = s.eng->stream;
  const uint64_t mine = (uint64_t)(uint32_t)*res;
  std::vector<uint64_t> every(g->world, 0);
  HIP_TRY(hipStreamWaitEvent(st, s.ev_arrived, 0));   // one collective of a communicator at a time: the exchange for the next pass first
  HIP_TRY(hipMemcpyAsync(s.send, &mine, sizeof(mine), hipMemcpyHostToDevice, st));
  LP_TRY(all_gather_words(g, s, s.send, s.stage, 1, st));
  HIP_TRY(hipMemcpyAsync(every.data(), s.stage, sizeof(uint64_t) * g->world, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  for (int k = 0; k < g->world; k++)
    if ((int32_t)every[k] != *res && *res == LPAMD_LIMIT) *res = (int32_t)every[k] == LPAMD_LIMIT ? *res : LPAMD_ERROR;
  return LPAMD_OK;
}

// MAX_COST: look-ahead exchange and chains of pivots per pass (see lpamd.h, lpamd_lookahead_* and
// lpamd_chain_step).  Per pass: set-up of the first pivot on the column the previous exchange
// delivered; chain - 1 times {export with ratio-test partials, all-gather, set-up straight from the
// slots} (or, without fused_step, {export, exchange + fold kernel, lpamd_chain_step}); export +
// exchange of the next pass's first pivot on the exchange stream while the pass streams the shard.
int run_lookahead(Group* g, Shard& s, int64_t k, int32_t* result, int64_t* done) {
  lpamd_engine* e = s.eng;
  int64_t start = 0;
  LP_TRY(lpamd_pivot_count(e, &start));
  if (k == 0) { *result = LPAMD_LIMIT; *done = 0; return LPAMD_OK; }
  LP_TRY(lpamd_set_pivot_budget(e, k));
  LP_TRY(export_lookahead(s));                 // nothing pending: the column as stored
  LP_TRY(exchange_lookahead(g, s, true));
  int64_t enqueued = 0;
  int32_t res = LPAMD_LIMIT;
  for (;;) {
    const int64_t n = k < 0 ? g->poll_every : std::min<int64_t>(g->poll_every, k - enqueued);
    const int64_t passes = (n + g->chain - 1) / g->chain;
    for (int64_t p = 0; p < passes; p++) {
      HIP_TRY(hipStreamWaitEvent(e->stream, s.ev_arrived, 0));
      LP_TRY(lpamd_lookahead_setup(e, s.xbuf));
      if (g->fused_step && e->shard_step && g->chain > 1) {
        // one launch and the collective per step: the set-up of step j also leaves this rank's slot for step j+1.
        // With the link there is no collective either: every launch stores its slot into every rank's receive buffer and the
        // next launch reads it there (tags: see ShardLink); the slot for step 1 comes from k_export_lookahead_link.
        ShardLink lk{};
        if (g->link) {
          lk.peers = s.link_peers;
          lk.in = s.link_buf;
          lk.ll_words = g->ll_words;
          lk.my_rank = s.rank;
          lk.timeout = g->link_timeout;
          s.link_tag = s.link_tag + 1 == 0 ? 1 : s.link_tag + 1;
          lk.tag_out = link_tag_out(g, s);
          LP_TRY(export_lookahead_ratio_link(e, g->world, kSlotHeaderWords, lk));
        } else {
          LP_TRY(export_lookahead_ratio(e, s.slot, kSlotHeaderWords));
        }
        for (int j = 1; j < g->chain; j++) {
          const bool last = j + 1 >= g->chain;
          if (!g->link) LP_TRY(all_gather_words(g, s, s.slot, s.slots, g->slot_words, e->stream));
          else LP_TRY(link_fence_local(g, s, e->stream));
          if (g->link) {
            lk.tag_in = s.link_tag;
            lk.tag_out = 0;
            if (!last) { s.link_tag = s.link_tag + 1 == 0 ? 1 : s.link_tag + 1; lk.tag_out = link_tag_out(g, s); }
          }
          LP_TRY(chain_step_export_from_slots(e, s.slots, g->world, g->slot_words, kSlotHeaderWords, last ? nullptr : s.slot, g->link ? &lk : nullptr));
        }
      } else
      for (int j = 1; j < g->chain; j++) {
        if (g->fused_step) {   // export with ratio-test partials, all-gather, set-up straight from the slots
          LP_TRY(export_lookahead_ratio(e, s.slot, kSlotHeaderWords));
          LP_TRY(all_gather_words(g, s, s.slot, s.slots, g->slot_words, e->stream));
          LP_TRY(chain_step_from_slots(e, s.slots, g->world, g->slot_words, kSlotHeaderWords));
          continue;
        }
        LP_TRY(export_lookahead(s));
        LP_TRY(exchange_lookahead(g, s, !g->inline_exchange));
        if (!g->inline_exchange) HIP_TRY(hipStreamWaitEvent(e->stream, s.ev_arrived, 0));
        LP_TRY(lpamd_chain_step(e, s.xbuf));
      }
      LP_TRY(export_lookahead(s));             // next candidate, as the pending pass will leave it ...
      LP_TRY(exchange_lookahead(g, s, true));  // ... travels while ...
      LP_TRY(lpamd_lookahead_update(e));       // ... the pass streams the shard
    }
    enqueued += passes * g->chain;
    LP_TRY(lpamd_poll(e, &res));
    LP_TRY(agree_on_result(g, s, &res));
    if (res != LPAMD_LIMIT || (k >= 0 && enqueued >= k)) break;
  }
  HIP_TRY(hipStreamWaitEvent(e->stream, s.ev_arrived, 0));   // the last exchange is not consumed; keep the streams ordered
  HIP_TRY(hipStreamSynchronize(e->stream));
  int64_t now = 0;
  LP_TRY(lpamd_pivot_count(e, &now));
  *result = res;
  *done = now - start;
  return LPAMD_OK;
}

// MAX_REDUCTION (and any strategy without look-ahead): pricing, ONE all-gather of record + column,
// fold, pivot; no host round trip inside a batch.
int run_in_order(Group* g, Shard& s, int32_t strategy, int64_t k, int32_t* result, int64_t* done) {
  lpamd_engine* e = s.eng;
  int64_t start = 0;
  LP_TRY(lpamd_pivot_count(e, &start));
  if (k == 0) { *result = LPAMD_LIMIT; *done = 0; return LPAMD_OK; }
  LP_TRY(lpamd_set_pivot_budget(e, k));
  int64_t enqueued = 0;
  int32_t res = LPAMD_LIMIT;
  Candidate* local = reinterpret_cast<Candidate*>(s.send);   // this rank's record before it is packed into the slot
  for (;;) {
    const int64_t n = k < 0 ? g->poll_every : std::min<int64_t>(g->poll_every, k - enqueued);
    for (int64_t i = 0; i < n; i++) {
      LP_TRY(lpamd_local_candidate(e, strategy, local));
      hipLaunchKernelGGL(k_export_candidate, dim3((unsigned)((g->lda + 255) / 256)), dim3(256), 0, e->stream, e->T, e->sh, e->st, local, s.slot);
      LP_TRY(check_launch("k_export_candidate"));
      LP_TRY(all_gather_words(g, s, s.slot, s.slots, g->slot_words, e->stream));
      hipLaunchKernelGGL(k_pick_winner<true>, dim3((unsigned)((g->lda + 255) / 256)), dim3(256), 0, e->stream, s.slots, g->world,
                         g->slot_words, e->sh, strategy == LPAMD_MAX_REDUCTION ? 1 : 0, e->st, e->cand, s.xbuf, nullptr, 0);
      LP_TRY(check_launch("k_pick_winner"));
      LP_TRY(lpamd_apply_pivot(e, strategy, -1, -1, s.xbuf));
    }
    enqueued += n;
    LP_TRY(lpamd_poll(e, &res));
    LP_TRY(agree_on_result(g, s, &res));
    if (res != LPAMD_LIMIT || (k >= 0 && enqueued >= k)) break;
  }
  int64_t now = 0;
  LP_TRY(lpamd_pivot_count(e, &now));
  *result = res;
  *done = now - start;
  return LPAMD_OK;
}

// MAX_REDUCTION from bounds (lpamd_prbound_kernels.hip.h): every rank prices its own columns (a few rows per
// column, exact evaluation of what the bounds leave), the ranks trade record + column in one all-gather, and
// the winner is set up on entries rebuilt through the pivots before it: a chain of pivots per pass here too.
int run_reduction_chain(Group* g, Shard& s, int64_t k, int32_t* result, int64_t* done) {
  lpamd_engine* e = s.eng;
  // everything a pass enqueues reads the depth from the engine: a pass of fewer steps than the engine's depth would take
  // the step-by-step path of the pass kernel
  struct DepthGuard {
    lpamd_engine* e; int32_t keep;
    ~DepthGuard() { e->chain_depth = keep; }
  } guard{e, e->chain_depth};
  e->chain_depth = std::min<int32_t>(e->chain_depth, g->chain_mr);
  int64_t start = 0;
  LP_TRY(lpamd_pivot_count(e, &start));
  if (k == 0) { *result = LPAMD_LIMIT; *done = 0; return LPAMD_OK; }
  LP_TRY(lpamd_set_pivot_budget(e, k));
  const unsigned pick_grid = (unsigned)((g->lda + 255) / 256);
  int64_t enqueued = 0;
  int32_t res = LPAMD_LIMIT;
  for (;;) {
    const int64_t n = k < 0 ? g->poll_every : std::min<int64_t>(g->poll_every, k - enqueued);
    const int64_t passes = (n + g->chain_mr - 1) / g->chain_mr;
    for (int64_t p = 0; p < passes; p++) {
      for (int j = 0; j < g->chain_mr; j++) {
        LP_TRY(prb_shard_price(e, s.slot, kSlotHeaderWords));
        LP_TRY(all_gather_words(g, s, s.slot, s.slots, g->slot_words, e->stream));
        if (j == 0) {   // the fold settles budget / optimum / unbounded and keeps the winner in the engine's candidate
          hipLaunchKernelGGL(k_pick_winner<true>, dim3(pick_grid), dim3(256), 0, e->stream, s.slots, g->world, g->slot_words, e->sh, 1, e->st,
                             e->cand, s.xbuf, nullptr, 0);
          LP_TRY(check_launch("k_pick_winner"));
          LP_TRY(prb_shard_setup(e, e->cand, s.xbuf));
        } else if (g->fused_step) {   // every workgroup of the set-up folds the records itself and takes column and row from the winner's slot
          LP_TRY(chain_step_from_slots(e, s.slots, g->world, g->slot_words, kSlotHeaderWords, true));
        } else {
          hipLaunchKernelGGL(k_pick_winner<false>, dim3(pick_grid), dim3(256), 0, e->stream, s.slots, g->world, g->slot_words, e->sh, 1, e->st,
                             e->cand_la, s.xbuf, e->steps, j);
          LP_TRY(check_launch("k_pick_winner"));
          LP_TRY(prb_shard_setup(e, e->cand_la, s.xbuf));
        }
      }
      LP_TRY(lpamd_lookahead_update(e));
    }
    enqueued += passes * g->chain_mr;
    LP_TRY(lpamd_poll(e, &res));
    LP_TRY(agree_on_result(g, s, &res));
    if (res != LPAMD_LIMIT || (k >= 0 && enqueued >= k)) break;
  }
  int64_t now = 0;
  LP_TRY(lpamd_pivot_count(e, &now));
  *result = res;
  *done = now - start;
  return LPAMD_OK;
}

int run_pivots_shard_loops(Group* g, Shard& s, int32_t strategy, int64_t k, int32_t* result, int64_t* done);
int run_pivots_shard(Group* g, Shard& s, int32_t strategy, int64_t k, int32_t* result, int64_t* done) {
  LP_TRY(run_pivots_shard_loops(g, s, strategy, k, result, done));
  if (*result == ST_DIRTY - 1)   // every rank sees the same entering columns, so all of them end here together
    return set_err(LPAMD_E_UNSUPPORTED, "basic columns are no longer unit columns (an entering column with entries of 4e9 and more): "
                                        "the loop that takes the reference's pivots on such a tableau runs on one GPU only");
  return LPAMD_OK;
}
int run_pivots_shard_loops(Group* g, Shard& s, int32_t strategy, int64_t k, int32_t* result, int64_t* done) {
  if (strategy != LPAMD_MAX_COST && strategy != LPAMD_MAX_REDUCTION)
    return set_err(LPAMD_E_UNSUPPORTED, "pivoting strategy %d is not implemented for the tableau path", strategy);
  int32_t probe = LPAMD_LIMIT;
  LP_TRY(lpamd_poll(s.eng, &probe));           // refreshes the host view of the pivot state
  if (s.eng->dense_first && k != 0) {
    // The first pivot after an upload with entries below the drop threshold is applied alone and to every column
    // (PivotState::dense_next; every rank saw the same matrix at upload, so all take this branch together).
    int64_t d1 = 0;
    LP_TRY(run_in_order(g, s, strategy, 1, result, &d1));
    *done = d1;
    if (*result != LPAMD_LIMIT || k == 1 || d1 == 0) return LPAMD_OK;
    int64_t d2 = 0;
    LP_TRY(run_pivots_shard_loops(g, s, strategy, k < 0 ? k : k - 1, result, &d2));
    *done = d1 + d2;
    return LPAMD_OK;
  }
  if (strategy == LPAMD_MAX_COST) return run_lookahead(g, s, k, result, done);
  if (strategy == LPAMD_MAX_REDUCTION && s.eng->pr_bounds && g->chain > 1) return run_reduction_chain(g, s, k, result, done);
  if (strategy == LPAMD_MAX_REDUCTION) return run_in_order(g, s, strategy, k, result, done);
  return set_err(LPAMD_E_UNSUPPORTED, "pivoting strategy %d is not implemented for the tableau path", strategy);
}

// LPModel::TableauSimplexInitialize over the shards (reference lp_simplex_tableau.cc:64-171).  The
// scans of that phase look at single rows, columns and the objective row; every rank assembles
// them with exact sums, takes the same decisions on the same data and applies the same explicit
// pivots to its shard.  The inner solve is the device-resident loop above.
int phase1_shard(Group* g, Shard& s, int32_t strategy, int32_t* result) {
  lpamd_engine* e = s.eng;
  const int64_t m = g->m, N = g->N, A = N + 1, W = N + 2;
  int32_t need = 0;
  LP_TRY(lpamd_need_init(e, &need));
  if (!need) { *result = LPAMD_SOLVED; return LPAMD_OK; }
  *result = LPAMD_ERROR;
  std::vector<double> saved(W, 0.0), tmp(W, 0.0), col(m), row(W, 0.0), obj_now(W, 0.0);
  std::vector<uint8_t> basic(W, 0);
  LP_TRY(assembled_objective(g, s, saved));
  tmp[A] = -1.0;                                                       // :79
  LP_TRY(lpamd_write_objective(e, tmp.data()));
  LP_TRY(lpamd_set_basic_flag(e, A, 0));                               // :71
  LP_TRY(lpamd_fill_column(e, A, 1.0));                                // :82-86
  LP_TRY(lpamd_read_column(e, N, col.data()));
  int64_t r = -1;
  double mv = 0.0;
  for (int64_t i = 0; i < m; i++)                                      // MinReduce, ties to the later row (:88-91)
    if (!(mv < col[i])) { mv = col[i]; r = i; }
  if (r < 0) return LPAMD_OK;
  LP_TRY(assembled_row(g, s, r, row));
  LP_TRY(assembled_basic(g, s, basic));
  int64_t d = -1;
  for (int64_t j = 0; j < W; j++) {                                    // :94-104
    if (j == N || h_is_zero(row[j])) continue;
    if (basic[j]) { d = j; break; }
  }
  if (d < 0) return LPAMD_OK;
  LP_TRY(pivot_shards(g, s, d, A, r, tmp[A]));
  LP_TRY(lpamd_read_column(e, N, col.data()));
  for (int64_t i = 0; i < m; i++)                                      // :106-108 (live assert upstream)
    if (!h_is_nonneg(col[i])) return LPAMD_OK;
  int32_t inner = LPAMD_ERROR;
  int64_t inner_done = 0;
  LP_TRY(run_pivots_shard(g, s, strategy, -1, &inner, &inner_done));   // :110
  if (inner != LPAMD_SOLVED) return LPAMD_OK;                          // :111 (live assert upstream)
  double oc = 0.0;
  LP_TRY(lpamd_objective_constant(e, &oc));
  if (h_is_neg(oc)) { *result = LPAMD_NOSOLUTION; return LPAMD_OK; }   // :113-115
  LP_TRY(assembled_basic(g, s, basic));
  if (basic[A]) {                                                      // :117-135
    LP_TRY(assembled_column(g, s, A, col));
    bool pivoted = false;
    for (int64_t i = 0; i < m && !pivoted; i++) {
      if (h_is_zero(col[i])) continue;
      LP_TRY(assembled_row(g, s, i, row));
      for (int64_t j = 0; j < W; j++) {
        if (j != N && basic[j]) continue;
        if (h_is_zero(row[j])) continue;
        if (j == N) return LPAMD_OK;   // upstream would pivot on the constant column and abort on its assert
        LP_TRY(assembled_objective(g, s, obj_now));
        LP_TRY(pivot_shards(g, s, A, j, i, obj_now[j]));
        pivoted = true;
        break;
      }
    }
    if (!pivoted) return LPAMD_OK;
    LP_TRY(lpamd_set_basic_flag(e, A, 0));
    LP_TRY(assembled_basic(g, s, basic));
  }
  LP_TRY(lpamd_fill_column(e, A, 0.0));                                // :143
  saved[A] = 0.0;
  std::vector<double> acc(W, 0.0);                                     // objective re-expression (:149-166)
  for (int64_t j = 0; j < W; j++) {
    if (j == N || !basic[j] || h_is_zero(saved[j])) continue;
    LP_TRY(assembled_column(g, s, j, col));
    int64_t rho = -1;
    double bv = (double)1e-6f;                                         // MaxAbsReduce init {-1, kEpsilonF}
    for (int64_t i = 0; i < m; i++)
      if (!(std::fabs(bv) > std::fabs(col[i]))) { bv = col[i]; rho = i; }
    if (rho < 0) return LPAMD_OK;
    LP_TRY(assembled_row(g, s, rho, row));
    const double sc = -saved[j] / row[j];                              // :162
    for (int64_t q = 0; q < W; q++) {
      const double prod = row[q] * sc;
      acc[q] = h_flushz(acc[q] + prod);
    }
  }
  for (int64_t q = 0; q < W; q++) saved[q] = h_flushz(saved[q] + acc[q]);   // :166
  LP_TRY(lpamd_write_objective(e, saved.data()));
  *result = LPAMD_SOLVED;
  return LPAMD_OK;
}

// LPModel::TableauSimplexSolve over the shards (reference lp_simplex_tableau.cc:173-379)
int solve_shard(Group* g, Shard& s, int32_t strategy, int64_t max_pivots, int32_t* result, int64_t* n_pivots) {
  lpamd_engine* e = s.eng;
  int32_t need = 0;
  LP_TRY(lpamd_need_init(e, &need));
  if (need) {                                                          // :174-178
    int32_t r1 = LPAMD_ERROR;
    LP_TRY(phase1_shard(g, s, strategy, &r1));
    if (r1 != LPAMD_SOLVED) {
      *result = r1 == LPAMD_NOSOLUTION ? LPAMD_NOSOLUTION : LPAMD_ERROR;
      return lpamd_pivot_count(e, n_pivots);
    }
    LP_TRY(lpamd_need_init(e, &need));
    if (need) { *result = LPAMD_ERROR; return lpamd_pivot_count(e, n_pivots); }   // :179
  }
  int64_t done = 0;
  LP_TRY(run_pivots_shard(g, s, strategy, max_pivots, result, &done));
  return lpamd_pivot_count(e, n_pivots);
}

// Receive buffers of the exchange inside a launch: memory that no cache of this or another GPU keeps a copy of (the peers
// store into it, this rank polls it).  Such memory is kept for the life of the PROCESS and handed from group to group
// (cleared each time), never given back to the allocator: pages that were mapped uncached, freed and handed out again by
// hipMalloc as ordinary memory were seen to serve OLD contents to the kernels of the next owner (the second of two groups
// created one after the other walked other pivots from its third pivot on; with the buffers from hipMalloc, or kept, it
// does not).  A process holds as many of them as it had shards alive at once, a few MB each.
struct LinkBlock { int device; size_t bytes; uint64_t* ptr; bool in_use; };
std::mutex g_link_pool_mutex;
std::vector<LinkBlock> g_link_pool;

int link_pool_get(int device, size_t bytes, uint64_t** out) {
  std::lock_guard<std::mutex> lock(g_link_pool_mutex);
  LinkBlock* hit = nullptr;
  for (LinkBlock& b : g_link_pool)
    if (!b.in_use && b.device == device && b.bytes == bytes) { hit = &b; break; }
  if (!hit) {
    uint64_t* p = nullptr;
    const char* dbg_mem = getenv("LPAMD_LINK_MEM");   // experiments only: "plain" takes hipMalloc memory
    if (dbg_mem && !strcmp(dbg_mem, "plain")) HIP_TRY(hipMalloc(&p, bytes));
    else if (hipExtMallocWithFlags((void**)&p, bytes, hipDeviceMallocUncached) != hipSuccess) {
      (void)hipGetLastError();
      p = nullptr;
      HIP_TRY(hipExtMallocWithFlags((void**)&p, bytes, hipDeviceMallocFinegrained));
    }
    g_link_pool.push_back(LinkBlock{device, bytes, p, false});
    hit = &g_link_pool.back();
  }
  HIP_TRY(hipMemset(hit->ptr, 0, bytes));    // tags of an earlier group must not read as this one's
  HIP_TRY(hipDeviceSynchronize());
  hit->in_use = true;
  *out = hit->ptr;
  return LPAMD_OK;
}
void link_pool_put(uint64_t* p) {
  if (!p) return;
  std::lock_guard<std::mutex> lock(g_link_pool_mutex);
  for (LinkBlock& b : g_link_pool)
    if (b.ptr == p) b.in_use = false;
}

void destroy_shard(Group* g, Shard& s) {
  (void)hipSetDevice(s.device);
  if (s.eng && s.eng->stream) (void)hipStreamSynchronize(s.eng->stream);   // (returns once an aborted collective has been flushed)
  if (s.xstream) (void)hipStreamSynchronize(s.xstream);
  if (s.comm && !s.comm_aborted) {   // ncclCommAbort has already released an aborted communicator
    RcclApi* api = nullptr;
    if (rccl_api(&api) == LPAMD_OK) (void)api->CommDestroy(s.comm);
  }
  for (hipEvent_t ev : {s.ev_sent, s.ev_arrived, s.ev_pub, s.ev_read})
    if (ev) (void)hipEventDestroy(ev);
  for (void* p : s.link_mapped) (void)hipIpcCloseMemHandle(p);
  link_pool_put(s.link_buf);
  for (void* p : {(void*)s.link_peers, (void*)s.link_ok})
    if (p) (void)hipFree(p);
  if (s.xstream) (void)hipStreamDestroy(s.xstream);
  for (void* p : {(void*)s.slot, (void*)s.slots, (void*)s.xbuf, (void*)s.send, (void*)s.stage, (void*)s.summed})
    if (p) (void)hipFree(p);
  if (s.eng) lpamd_destroy(s.eng);
  s = Shard();
  (void)g;
}

int64_t link_gather_at(const Group* g) { return (int64_t)kLinkBuffers * g->world * g->ll_words; }
int64_t link_hello_at(const Group* g) { return link_gather_at(g) + (int64_t)kLinkBuffers * g->world * 2 * g->gather_cap; }
int64_t link_words(const Group* g) { return link_hello_at(g) + 2 * (int64_t)g->world + 64; }

int init_shard(Group* g, Shard& s, int rank, int device) {
  s.rank = rank;
  s.device = device;
  HIP_TRY(hipSetDevice(device));
  LP_TRY(lpamd_create(&s.eng, g->m, g->N, rank, g->engine_world, nullptr));
  HIP_TRY(hipStreamCreateWithFlags(&s.xstream, hipStreamNonBlocking));
  for (hipEvent_t* ev : {&s.ev_sent, &s.ev_arrived, &s.ev_pub, &s.ev_read}) HIP_TRY(hipEventCreateWithFlags(ev, hipEventDisableTiming));
  LP_TRY(dev_alloc(s.eng, (void**)&s.slot, sizeof(uint64_t) * g->slot_words, "slot"));
  LP_TRY(dev_alloc(s.eng, (void**)&s.slots, sizeof(uint64_t) * g->slot_words * g->world, "slots"));
  LP_TRY(dev_alloc(s.eng, (void**)&s.xbuf, sizeof(double) * g->lda, "xbuf"));
  LP_TRY(dev_alloc(s.eng, (void**)&s.send, sizeof(uint64_t) * g->wide, "send"));
  LP_TRY(dev_alloc(s.eng, (void**)&s.stage, sizeof(uint64_t) * g->wide * g->world, "stage"));
  LP_TRY(dev_alloc(s.eng, (void**)&s.summed, sizeof(uint64_t) * g->wide, "summed"));
  HIP_TRY(hipStreamSynchronize(s.eng->stream));
  if (g->link) {
    LP_TRY(link_pool_get(device, sizeof(uint64_t) * (size_t)link_words(g), &s.link_buf));
    HIP_TRY(hipMalloc(&s.link_peers, sizeof(uint64_t*) * g->world));
    HIP_TRY(hipMalloc(&s.link_ok, sizeof(int32_t)));
    HIP_TRY(hipMemset(s.link_ok, 0, sizeof(int32_t)));
  }
  return LPAMD_OK;
}

// ---- the exchange inside the launch: who maps whose receive buffers ---------------------------------
// Shards of ONE process: plain pointers (peer access between the devices switched on), then the self-test.
int link_connect_local(Group* g) {
  if (!g->link) return LPAMD_OK;
  for (Shard& s : g->shards) g->link = g->link && s.eng->shard_step;
  if (g->chain < 2) g->link = false;       // no chained step, nothing to carry
  if (!g->link) return LPAMD_OK;
  std::vector<uint64_t*> peers(g->world, nullptr);
  for (Shard& s : g->shards) peers[s.rank] = s.link_buf;
  for (Shard& s : g->shards) {
    HIP_TRY(hipSetDevice(s.device));
    for (Shard& q : g->shards)
      if (q.device != s.device) {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, s.device, q.device) != hipSuccess || !can) { (void)hipGetLastError(); g->link = false; return LPAMD_OK; }
        const hipError_t err = hipDeviceEnablePeerAccess(q.device, 0);
        if (err != hipSuccess && err != hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); g->link = false; return LPAMD_OK; }
        (void)hipGetLastError();
      }
    HIP_TRY(hipMemcpy(s.link_peers, peers.data(), sizeof(uint64_t*) * g->world, hipMemcpyHostToDevice));
  }
  if (getenv("LPAMD_LINK_NOHELLO")) return LPAMD_OK;   // experiments only
  for (Shard& s : g->shards) {
    HIP_TRY(hipSetDevice(s.device));
    hipLaunchKernelGGL(k_link_hello_put, dim3(1), dim3(kWave), 0, s.eng->stream, s.link_peers, g->world, link_hello_at(g), s.rank);
    LP_TRY(check_launch("k_link_hello_put"));
  }
  for (Shard& s : g->shards) { HIP_TRY(hipSetDevice(s.device)); HIP_TRY(hipStreamSynchronize(s.eng->stream)); }
  bool all = true;
  for (Shard& s : g->shards) {
    HIP_TRY(hipSetDevice(s.device));
    hipLaunchKernelGGL(k_link_hello_get, dim3(1), dim3(kWave), 0, s.eng->stream, s.link_buf, g->world, link_hello_at(g), s.link_ok);
    LP_TRY(check_launch("k_link_hello_get"));
    int32_t ok = 0;
    HIP_TRY(hipMemcpyAsync(&ok, s.link_ok, sizeof(ok), hipMemcpyDeviceToHost, s.eng->stream));
    HIP_TRY(hipStreamSynchronize(s.eng->stream));
    all = all && ok == 1;
  }
  g->link = all;
  g->link_gather = all && !g->local_exchange;   // (shards that share a GPU keep their event-ordered copies)
  if (const char* env = getenv("LPAMD_LINK_GATHER")) g->link_gather = g->link_gather && atoi(env) != 0;
  return LPAMD_OK;
}

// One process per GPU: the ranks trade IPC handles of their receive buffers through the communicator, map each other's,
// run the self-test and agree on its outcome; anything short of "every rank reached every rank" leaves the collective in
// charge on ALL ranks.  Collective: every rank makes the same three all-gathers whatever happens to it locally.
int link_connect_ranks(Group* g, Shard& s) {
  if (!g->link) return LPAMD_OK;
  hipStream_t st = s.eng->stream;
  constexpr int kWords = 1 + (int)(sizeof(hipIpcMemHandle_t) / sizeof(uint64_t));
  static_assert(sizeof(hipIpcMemHandle_t) % sizeof(uint64_t) == 0, "handles travel as 64-bit words");
  std::vector<uint64_t> mine(kWords, 0), every((size_t)kWords * g->world, 0);
  hipIpcMemHandle_t h;
  // (g->link is what the environment asked for, the same on every rank; whether THIS rank's engine runs the one-launch
  // step is not, and travels with the handle)
  if (!s.eng->shard_step || g->chain < 2) mine[0] = 0;
  else if (hipIpcGetMemHandle(&h, s.link_buf) == hipSuccess) { mine[0] = 1; std::memcpy(&mine[1], &h, sizeof(h)); }
  else (void)hipGetLastError();
  auto gather = [&](int words) -> int {
    HIP_TRY(hipMemcpyAsync(s.send, mine.data(), sizeof(uint64_t) * words, hipMemcpyHostToDevice, st));
    LP_TRY(all_gather_words(g, s, s.send, s.stage, words, st));
    HIP_TRY(hipMemcpyAsync(every.data(), s.stage, sizeof(uint64_t) * words * g->world, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return LPAMD_OK;
  };
  LP_TRY(gather(kWords));
  bool good = true;
  for (int k = 0; k < g->world; k++) good = good && every[(size_t)k * kWords] == 1;
  std::vector<uint64_t*> peers(g->world, nullptr);
  if (good) {
    for (int k = 0; k < g->world && good; k++) {
      if (k == s.rank) { peers[k] = s.link_buf; continue; }
      hipIpcMemHandle_t hk;
      std::memcpy(&hk, &every[(size_t)k * kWords + 1], sizeof(hk));
      void* p = nullptr;
      if (hipIpcOpenMemHandle(&p, hk, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); good = false; break; }
      s.link_mapped.push_back(p);
      peers[k] = static_cast<uint64_t*>(p);
    }
  }
  if (good) {
    HIP_TRY(hipMemcpyAsync(s.link_peers, peers.data(), sizeof(uint64_t*) * g->world, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_link_hello_put, dim3(1), dim3(kWave), 0, st, s.link_peers, g->world, link_hello_at(g), s.rank);
    LP_TRY(check_launch("k_link_hello_put"));
  }
  mine[0] = good ? 1 : 0;
  LP_TRY(gather(1));                       // also a barrier: every rank's tokens are on their way before anybody looks
  for (int k = 0; k < g->world; k++) good = good && every[k] == 1;
  int32_t ok = 0;
  if (good) {
    hipLaunchKernelGGL(k_link_hello_get, dim3(1), dim3(kWave), 0, st, s.link_buf, g->world, link_hello_at(g), s.link_ok);
    LP_TRY(check_launch("k_link_hello_get"));
    HIP_TRY(hipMemcpyAsync(&ok, s.link_ok, sizeof(ok), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
  }
  mine[0] = ok == 1 ? 1 : 0;
  LP_TRY(gather(1));
  for (int k = 0; k < g->world; k++) good = good && every[k] == 1;
  g->link = good;
  // every later all-gather of a slot's size or less goes the same way (pass boundary, MAX_REDUCTION steps, in-order loop,
  // the ranks' agreement); longer ones (rows and objective of the feasibility phase) stay with the collective
  g->link_gather = good;
  if (const char* env = getenv("LPAMD_LINK_GATHER")) g->link_gather = g->link_gather && atoi(env) != 0;
  return LPAMD_OK;
}

Group* new_group(int64_t m, int64_t N, int world, int32_t flags) {
  Group* g = new Group();
  g->m = m;
  g->N = N;
  g->world = world;
  g->engine_world = world;
  g->lda = lpamd_exchange_bytes(m) / (int64_t)sizeof(double);
  if (const char* env = getenv("LPAMD_LDA_PAD")) g->lda += round_up(std::max(0, atoi(env)), kSlabRows);
  g->slot_words = kSlotHeaderWords + g->lda + 2 * ((g->lda + kPartialThreads - 1) / kPartialThreads);   // record, column, ratio-test partials
  g->wide = std::max<int64_t>(N + 2, g->lda);
  g->local_exchange = (flags & LPAMD_GROUP_LOCAL_EXCHANGE) != 0;
  g->pub.assign(world, nullptr);
  if (const char* env = getenv("LPAMD_SHARD_INLINE")) g->inline_exchange = atoi(env) != 0;
  // The in-process exchange enqueues 2 event records and `world` device copies per shard and pivot.  With 8 shards on
  // one GPU and 64 pivots between two reads of the pivot state, that backlog was seen to exhaust something in the
  // driver that every process on the machine needs (other processes' HSA start-up crashed until the backlog had
  // drained); 16 pivots between reads keep it short.  RCCL exchanges enqueue one collective per pivot at most and read the
  // state every 256 pivots (a read drains the stream: rank 0 of 8 of m=32768 n=65536 walks 20.8 k pivots/s at 64, 21.0 k at 256).
  // Every cross-stream event wait is a barrier packet with a completion signal of its own in the queue, so the likely
  // resource is the runtime's pool of signals / the queue's packet ring.  The bound holds whatever the environment asks for.
  if (g->local_exchange) g->poll_every = 16;
  if (const char* env = getenv("LPAMD_SHARD_POLL")) g->poll_every = std::max(1, atoi(env));
  g->poll_every = std::min(g->poll_every, g->local_exchange ? 16 : 256);
  if (const char* env = getenv("LPAMD_SHARD_FUSED")) g->fused_step = atoi(env) != 0;
  // The exchange inside the step's launch: wherever the one-launch step runs (the connect functions look at the engines).
  // Shards that share a GPU (in-process exchange) order their launches with events on top, see link_fence_local.
  g->ll_words = 2 * (g->slot_words + (g->lda + kPartialThreads - 1) / kPartialThreads);   // + the column's entry at the row of every ratio-test partial
  g->link = g->fused_step && world <= kWave;
  g->gather_cap = g->slot_words;
  if (const char* env = getenv("LPAMD_SHARD_LINK")) g->link = g->link && atoi(env) != 0;
  if (const char* env = getenv("LPAMD_LINK_TIMEOUT_MS")) g->link_timeout = std::max(1, atoi(env)) * 100000LL;
  if (const char* env = getenv("LPAMD_LINK_FAULT")) g->link_fault_at = atoll(env);
  return g;
}

int finish_group(Group* g) {
  int32_t depth = 1;
  LP_TRY(lpamd_chain_depth(g->shards[0].eng, &depth));
  g->chain = depth;
  if (const char* env = getenv("LPAMD_SHARD_CHAIN")) g->chain = std::max(1, std::min(depth, atoi(env)));
  int32_t depth_mr = depth;
  LP_TRY(lpamd_loop_depth(g->shards[0].eng, LPAMD_MAX_REDUCTION, &depth_mr));
  g->chain_mr = std::max(1, std::min<int>(g->chain, depth_mr));
  return LPAMD_OK;
}

#define CHECK_GROUP(g)                                                   \
  do {                                                                   \
    if (!(g) || (g)->shards.empty()) return set_err(LPAMD_E_ARG, "null group"); \
  } while (0)

}  // namespace

extern "C" {

int lpamd_comm_unique_id(void* id_out) {
  if (!id_out) return set_err(LPAMD_E_ARG, "comm_unique_id: null");
  static_assert(sizeof(ncclUniqueId) == LPAMD_COMM_ID_BYTES, "LPAMD_COMM_ID_BYTES must match ncclUniqueId");
  RcclApi* api = nullptr;
  LP_TRY(rccl_api(&api));
  ncclUniqueId id;
  RCCL_TRY(api, api->GetUniqueId(&id));
  std::memcpy(id_out, &id, sizeof(id));
  return LPAMD_OK;
}

int lpamd_group_create(lpamd_group** out, int64_t m, int64_t N, int32_t n_shards, const int32_t* devices, int32_t flags) {
  if (!out) return set_err(LPAMD_E_ARG, "group_create: out is null");
  *out = nullptr;
  if (m <= 0 || N <= 0 || n_shards <= 0 || n_shards > N + 1) return set_err(LPAMD_E_ARG, "group_create: bad shape m=%lld N=%lld shards=%d", (long long)m, (long long)N, n_shards);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return set_err(LPAMD_E_NODEVICE, "no HIP device visible");
  const bool local = (flags & LPAMD_GROUP_LOCAL_EXCHANGE) != 0;
  std::vector<int> devs(n_shards);
  for (int k = 0; k < ((flags & LPAMD_GROUP_REHEARSE_RANK0) ? 1 : n_shards); k++) {
    devs[k] = devices ? devices[k] : k;
    if (devs[k] < 0 || devs[k] >= ndev) return set_err(LPAMD_E_NODEVICE, "group_create: shard %d wants device %d, %d visible", k, devs[k], ndev);
    for (int q = 0; q < k && !local; q++)
      if (devs[q] == devs[k]) return set_err(LPAMD_E_ARG, "group_create: device %d used twice (RCCL needs one GPU per shard)", devs[k]);
  }
  int prev = 0;
  (void)hipGetDevice(&prev);
  // Rehearsal: only rank 0 of the deal exists and trades with itself; the work per pivot is that of one rank
  // of an n_shards-GPU run (the pivots are not the LP's: the other ranks' candidates are missing).
  const bool rehearse = (flags & LPAMD_GROUP_REHEARSE_RANK0) != 0;
  const int engine_world = n_shards;
  if (rehearse) { n_shards = 1; devs.resize(1); }
  Group* g = new_group(m, N, n_shards, flags);
  g->engine_world = engine_world;
  g->shards.resize(n_shards);
  int rc = LPAMD_OK;
  for (int k = 0; k < n_shards && rc == LPAMD_OK; k++) rc = init_shard(g, g->shards[k], k, devs[k]);
  if (rc == LPAMD_OK && !local) {
    RcclApi* api = nullptr;
    rc = rccl_api(&api);
    if (rc == LPAMD_OK) {
      std::vector<ncclComm_t> comms(n_shards, nullptr);
      ncclResult_t r = api->CommInitAll(comms.data(), n_shards, devs.data());
      if (r != ncclSuccess) rc = set_err(LPAMD_E_HIP, "ncclCommInitAll failed: %s", api->GetErrorString(r));
      else for (int k = 0; k < n_shards; k++) g->shards[k].comm = comms[k];
    }
  }
  if (rc == LPAMD_OK) rc = finish_group(g);
  // (a rehearsal over RCCL takes the road of one process per GPU: handles through the communicator, the three all-gathers)
  if (rc == LPAMD_OK) rc = (rehearse && !local) ? link_connect_ranks(g, g->shards[0]) : link_connect_local(g);
  (void)hipSetDevice(prev);
  if (rc != LPAMD_OK) {
    char keep[sizeof(g_err)];
    std::memcpy(keep, g_err, sizeof(keep));
    lpamd_group_destroy(g);
    std::memcpy(g_err, keep, sizeof(keep));
    return rc;
  }
  *out = g;
  return LPAMD_OK;
}

static int create_rank_impl(lpamd_group** out, int64_t m, int64_t N, int32_t rank, int32_t world, const void* comm_id,
                            lpamd_host_allgather_fn gather, void* ctx) {
  if (!out) return set_err(LPAMD_E_ARG, "group_create_rank: out is null");
  *out = nullptr;
  // (world > N + 1 would leave a rank without a column: every rank must see that BEFORE any of them enters ncclCommInitRank)
  if (m <= 0 || N <= 0 || world <= 0 || world > N + 1 || rank < 0 || rank >= world || (!comm_id && !gather))
    return set_err(LPAMD_E_ARG, "group_create_rank: bad arguments m=%lld N=%lld rank=%d world=%d", (long long)m, (long long)N, rank, world);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return set_err(LPAMD_E_NODEVICE, "no HIP device visible");
  Group* g = new_group(m, N, world, 0);
  g->host_gather = gather;
  g->host_ctx = ctx;
  g->shards.resize(1);
  int rc = init_shard(g, g->shards[0], rank, dev);
  if (rc == LPAMD_OK && !gather) {
    RcclApi* api = nullptr;
    rc = rccl_api(&api);
    if (rc == LPAMD_OK) {
      ncclUniqueId id;
      std::memcpy(&id, comm_id, sizeof(id));
      ncclResult_t r = api->CommInitRank(&g->shards[0].comm, world, id, rank);
      if (r != ncclSuccess) rc = set_err(LPAMD_E_HIP, "ncclCommInitRank failed: %s", api->GetErrorString(r));
      int count = 0;
      if (rc == LPAMD_OK && (api->CommCount(g->shards[0].comm, &count) != ncclSuccess || count != world))
        rc = set_err(LPAMD_E_STATE, "the communicator holds %d ranks, %d expected", count, world);
    }
  }
  if (rc == LPAMD_OK) rc = finish_group(g);
  if (rc == LPAMD_OK) rc = link_connect_ranks(g, g->shards[0]);
  if (rc != LPAMD_OK) {
    char keep[sizeof(g_err)];
    std::memcpy(keep, g_err, sizeof(keep));
    lpamd_group_destroy(g);
    std::memcpy(g_err, keep, sizeof(keep));
    return rc;
  }
  *out = g;
  return LPAMD_OK;
}

int lpamd_group_create_rank(lpamd_group** out, int64_t m, int64_t N, int32_t rank, int32_t world, const void* comm_id) {
  if (!comm_id) return set_err(LPAMD_E_ARG, "group_create_rank: no communicator id");
  return create_rank_impl(out, m, N, rank, world, comm_id, nullptr, nullptr);
}

int lpamd_group_create_rank_hosted(lpamd_group** out, int64_t m, int64_t N, int32_t rank, int32_t world,
                                   lpamd_host_allgather_fn gather, void* ctx) {
  if (!gather) return set_err(LPAMD_E_ARG, "group_create_rank_hosted: no all-gather function");
  return create_rank_impl(out, m, N, rank, world, nullptr, gather, ctx);
}

int lpamd_group_destroy(lpamd_group* g) {
  if (!g) return LPAMD_OK;
  int prev = 0;
  (void)hipGetDevice(&prev);
  for (Shard& s : g->shards) destroy_shard(g, s);
  (void)hipSetDevice(prev);
  delete g;
  return LPAMD_OK;
}

int lpamd_group_shards(const lpamd_group* g, int32_t* n_local, int32_t* world) {
  CHECK_GROUP(g);
  if (n_local) *n_local = (int32_t)g->shards.size();
  if (world) *world = g->world;
  return LPAMD_OK;
}

int lpamd_group_comm_ranks(lpamd_group* g, int32_t* ranks) {
  CHECK_GROUP(g);
  if (!ranks) return set_err(LPAMD_E_ARG, "group_comm_ranks: null");
  *ranks = 0;
  if (g->local_exchange || !g->shards[0].comm || g->shards[0].comm_aborted) return LPAMD_OK;
  RcclApi* api = nullptr;
  LP_TRY(rccl_api(&api));
  int count = 0;
  RCCL_TRY(api, api->CommCount(g->shards[0].comm, &count));
  *ranks = count;
  return LPAMD_OK;
}

int lpamd_group_link_active(lpamd_group* g, int32_t* active) {
  CHECK_GROUP(g);
  if (!active) return set_err(LPAMD_E_ARG, "group_link_active: null");
  *active = g->link ? 1 : 0;
  return LPAMD_OK;
}

int lpamd_group_engine(lpamd_group* g, int32_t local_index, lpamd_engine** out) {
  CHECK_GROUP(g);
  if (!out || local_index < 0 || local_index >= (int32_t)g->shards.size()) return set_err(LPAMD_E_ARG, "group_engine: bad index %d", local_index);
  *out = g->shards[local_index].eng;
  return LPAMD_OK;
}

int lpamd_group_chain(lpamd_group* g, int32_t* pivots_per_pass) {
  CHECK_GROUP(g);
  if (!pivots_per_pass) return set_err(LPAMD_E_ARG, "group_chain: null");
  *pivots_per_pass = g->chain;
  return LPAMD_OK;
}

int lpamd_group_upload(lpamd_group* g, const double* T, int64_t ld, const double* obj, const uint8_t* is_basic) {
  CHECK_GROUP(g);
  return for_each_shard(g, [&](Shard& s) { return lpamd_upload(s.eng, T, ld, obj, is_basic); });
}

int lpamd_group_fill_synthetic(lpamd_group* g, int64_t n_struct, uint64_t seed) {
  CHECK_GROUP(g);
  return for_each_shard(g, [&](Shard& s) { return lpamd_fill_synthetic(s.eng, n_struct, seed); });
}

int lpamd_group_download(lpamd_group* g, double* T, int64_t ld, double* obj, uint8_t* is_basic) {
  CHECK_GROUP(g);
  for (Shard& s : g->shards) {   // one after the other: the shards write disjoint columns of the same host matrix
    HIP_TRY(hipSetDevice(s.device));
    LP_TRY(lpamd_download(s.eng, T, ld, obj, is_basic));
  }
  return LPAMD_OK;
}

int lpamd_group_run_pivots(lpamd_group* g, int32_t strategy, int64_t k, int32_t* result, int64_t* done) {
  CHECK_GROUP(g);
  if (!result) return set_err(LPAMD_E_ARG, "group_run_pivots: null result");
  std::vector<int32_t> res(g->shards.size(), LPAMD_ERROR);
  std::vector<int64_t> dn(g->shards.size(), 0);
  LP_TRY(for_each_shard(g, [&](Shard& s) {
    const size_t k_local = &s - g->shards.data();
    return run_pivots_shard(g, s, strategy, k, &res[k_local], &dn[k_local]);
  }));
  for (size_t q = 1; q < res.size(); q++)
    if (res[q] != res[0] || dn[q] != dn[0]) return set_err(LPAMD_E_STATE, "shards disagree: result %d/%d, pivots %lld/%lld", res[0], res[q], (long long)dn[0], (long long)dn[q]);
  *result = res[0];
  if (done) *done = dn[0];
  return LPAMD_OK;
}

int lpamd_group_solve(lpamd_group* g, int32_t strategy, int64_t max_pivots, int32_t* result, int64_t* n_pivots) {
  CHECK_GROUP(g);
  if (!result) return set_err(LPAMD_E_ARG, "group_solve: null result");
  if (strategy != LPAMD_MAX_COST && strategy != LPAMD_MAX_REDUCTION)
    return set_err(LPAMD_E_UNSUPPORTED, "pivoting strategy %d is not implemented for the tableau path", strategy);
  std::vector<int32_t> res(g->shards.size(), LPAMD_ERROR);
  std::vector<int64_t> np(g->shards.size(), 0);
  LP_TRY(for_each_shard(g, [&](Shard& s) {
    const size_t k_local = &s - g->shards.data();
    return solve_shard(g, s, strategy, max_pivots, &res[k_local], &np[k_local]);
  }));
  for (size_t q = 1; q < res.size(); q++)
    if (res[q] != res[0] || np[q] != np[0]) return set_err(LPAMD_E_STATE, "shards disagree: result %d/%d, pivots %lld/%lld", res[0], res[q], (long long)np[0], (long long)np[q]);
  *result = res[0];
  if (n_pivots) *n_pivots = np[0];
  return LPAMD_OK;
}

int lpamd_group_pivot(lpamd_group* g, int64_t d, int64_t e, int64_t r) {
  CHECK_GROUP(g);
  if (e < 0 || e > g->N + 1 || e == g->N || r < 0 || r >= g->m || d > g->N + 1) return set_err(LPAMD_E_ARG, "group_pivot: bad arguments d=%lld e=%lld r=%lld", (long long)d, (long long)e, (long long)r);
  return for_each_shard(g, [&](Shard& s) {
    std::vector<double> obj(g->N + 2, 0.0);
    LP_TRY(assembled_objective(g, s, obj));
    return pivot_shards(g, s, d, e, r, obj[e]);
  });
}

int lpamd_group_need_init(lpamd_group* g, int32_t* out) {
  CHECK_GROUP(g);
  HIP_TRY(hipSetDevice(g->shards[0].device));
  return lpamd_need_init(g->shards[0].eng, out);   // the right-hand side is replicated
}

int lpamd_group_phase1(lpamd_group* g, int32_t strategy, int32_t* result) {
  CHECK_GROUP(g);
  if (!result) return set_err(LPAMD_E_ARG, "group_phase1: null result");
  std::vector<int32_t> res(g->shards.size(), LPAMD_ERROR);
  LP_TRY(for_each_shard(g, [&](Shard& s) { return phase1_shard(g, s, strategy, &res[&s - g->shards.data()]); }));
  *result = res[0];
  return LPAMD_OK;
}

int lpamd_group_column_values(lpamd_group* g, double* x) {
  CHECK_GROUP(g);
  if (!x) return set_err(LPAMD_E_ARG, "group_column_values: null");
  return for_each_shard(g, [&](Shard& s) {
    std::vector<double> v(g->N + 2, 0.0);
    LP_TRY(lpamd_column_values(s.eng, v.data()));
    LP_TRY(assemble(g, s, v));
    if (&s == g->shards.data()) std::memcpy(x, v.data(), sizeof(double) * (g->N + 2));
    return (int)LPAMD_OK;
  });
}

int lpamd_group_objective_constant(lpamd_group* g, double* out) {
  CHECK_GROUP(g);
  HIP_TRY(hipSetDevice(g->shards[0].device));
  return lpamd_objective_constant(g->shards[0].eng, out);   // replicated with the right-hand side
}

int lpamd_group_pivot_count(lpamd_group* g, int64_t* n) {
  CHECK_GROUP(g);
  HIP_TRY(hipSetDevice(g->shards[0].device));
  return lpamd_pivot_count(g->shards[0].eng, n);
}

int lpamd_group_trace(lpamd_group* g, int64_t* e, int64_t* r, int64_t cap, int64_t* n) {
  CHECK_GROUP(g);
  HIP_TRY(hipSetDevice(g->shards[0].device));
  return lpamd_trace(g->shards[0].eng, e, r, cap, n);       // every rank records the full trace
}

}  // extern "C"
