// Probe for the exchange inside a launch (ShardLink): two PROCESSES on one GPU map each other's receive buffer through
// IPC handles (uncached device memory) and play ping-pong with tagged 64-bit words while both kernels are running,
// i.e. what two ranks of a multi-GPU run do, minus the xGMI hop.  Prints the round trip and whether every word arrived.
//   hipcc -O2 --offload-arch=gfx950 -o link_pingpong scripts/link_pingpong.hip && ./link_pingpong [rounds]
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <sys/wait.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "[%d] %s: %s\n", role, #x, hipGetErrorString(e_)); _exit(2); } } while (0)

static int role = 0;

__device__ __forceinline__ void put(uint64_t* p, uint32_t data, uint32_t tag) {
  __hip_atomic_store(p, ((uint64_t)data << 32) | tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ bool get(const uint64_t* p, uint32_t tag, uint32_t& data) {
  const long long t0 = wall_clock64();
  for (uint32_t r = 0; r < (1u << 26); r++) {
    const uint64_t w = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((uint32_t)w == tag) { data = (uint32_t)(w >> 32); return true; }
    if ((r & 63) == 63 && wall_clock64() - t0 > 500000000LL) return false;   // five seconds
    __builtin_amdgcn_s_sleep(1);
  }
  return false;
}

// `words` tagged words per message, one thread each.  Role 0 serves (sends tag t, waits for the echo), role 1 echoes.
__global__ void k_pingpong(uint64_t* mine, uint64_t* theirs, int words, int rounds, int role, int* bad, long long* ticks) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= words) return;
  const long long t0 = wall_clock64();
  int errors = 0;
  for (int r = 1; r <= rounds; r++) {
    uint32_t d = 0;
    if (role == 0) {
      put(theirs + t, (uint32_t)(r * 7 + t), (uint32_t)r);
      if (!get(mine + t, (uint32_t)r, d)) { errors = 1 << 20; break; }
      if (d != (uint32_t)(r * 7 + t + 1)) errors++;
    } else {
      if (!get(mine + t, (uint32_t)r, d)) { errors = 1 << 20; break; }
      if (d != (uint32_t)(r * 7 + t)) errors++;
      put(theirs + t, d + 1, (uint32_t)r);
    }
  }
  if (errors) atomicAdd(bad, errors);
  if (t == 0) *ticks = wall_clock64() - t0;
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 2000;
  const int words = argc > 2 ? atoi(argv[2]) : 4096;
  int to_child[2], to_parent[2];
  if (pipe(to_child) || pipe(to_parent)) return 2;
  const pid_t pid = fork();          // before anything touches the GPU
  role = pid == 0 ? 1 : 0;
  const int rd = role == 0 ? to_parent[0] : to_child[0], wr = role == 0 ? to_child[1] : to_parent[1];
  CK(hipSetDevice(0));
  uint64_t* mine = nullptr;
  const size_t bytes = sizeof(uint64_t) * words;
  const char* kind = "uncached";
  if (hipExtMallocWithFlags((void**)&mine, bytes, hipDeviceMallocUncached) != hipSuccess) {
    (void)hipGetLastError();
    kind = "fine-grained";
    CK(hipExtMallocWithFlags((void**)&mine, bytes, hipDeviceMallocFinegrained));
  }
  CK(hipMemset(mine, 0, bytes));
  CK(hipDeviceSynchronize());
  hipIpcMemHandle_t h, other;
  CK(hipIpcGetMemHandle(&h, mine));
  if (write(wr, &h, sizeof(h)) != (ssize_t)sizeof(h) || read(rd, &other, sizeof(other)) != (ssize_t)sizeof(other)) { fprintf(stderr, "[%d] pipe\n", role); _exit(2); }
  void* theirs = nullptr;
  CK(hipIpcOpenMemHandle(&theirs, other, hipIpcMemLazyEnablePeerAccess));
  int* bad = nullptr;
  long long* ticks = nullptr;
  CK(hipMalloc(&bad, sizeof(int)));
  CK(hipMalloc(&ticks, sizeof(long long)));
  CK(hipMemset(bad, 0, sizeof(int)));
  char go = 'g';   // both sides mapped: start
  if (write(wr, &go, 1) != 1 || read(rd, &go, 1) != 1) _exit(2);
  hipLaunchKernelGGL(k_pingpong, dim3((words + 255) / 256), dim3(256), 0, 0, mine, (uint64_t*)theirs, words, rounds, role, bad, ticks);
  CK(hipGetLastError());
  CK(hipDeviceSynchronize());
  int hbad = 0;
  long long hticks = 0;
  CK(hipMemcpy(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost));
  CK(hipMemcpy(&hticks, ticks, sizeof(long long), hipMemcpyDeviceToHost));
  printf("role %d (%s memory, %d words x %d rounds): %s, %.2f us per round trip\n", role, kind, words, rounds,
         hbad == 0 ? "every word arrived with its value" : (hbad >= (1 << 20) ? "TIMED OUT" : "WRONG VALUES"), hticks / 100.0 / rounds);
  fflush(stdout);
  if (write(wr, &go, 1) != 1 || read(rd, &go, 1) != 1) _exit(2);   // nobody unmaps while the other may still run
  CK(hipIpcCloseMemHandle(theirs));
  CK(hipFree(mine));
  if (role == 1) _exit(hbad ? 1 : 0);
  int status = 0;
  waitpid(pid, &status, 0);
  return (hbad || !WIFEXITED(status) || WEXITSTATUS(status)) ? 1 : 0;
}
