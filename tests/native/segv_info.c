/* TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Loaded by tests/conftest.py for GPU runs.  When the test process receives SIGSEGV / SIGBUS / SIGABRT this prints
 * what the signal itself says before handing over to the handler that was installed before (Python's faulthandler):
 * si_code tells a real fault of this thread (SEGV_MAPERR / SEGV_ACCERR with the address) from a signal SENT to the
 * process (SI_USER / SI_TKILL with the sender's pid; SI_KERNEL: the kernel, e.g. the GPU driver after a device-side
 * memory fault), and the native backtrace names the library the thread was in, which a Python traceback cannot.
 */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

static struct sigaction g_prev[65];

static void put(const char* s) { ssize_t r = write(2, s, strlen(s)); (void)r; }

static void handler(int sig, siginfo_t* si, void* uctx) {
  char line[256];
  const char* kind = si->si_code == SI_USER ? "SI_USER" : si->si_code == SI_TKILL ? "SI_TKILL" : si->si_code == SI_KERNEL ? "SI_KERNEL"
                   : (sig == SIGSEGV && si->si_code == SEGV_MAPERR) ? "SEGV_MAPERR" : (sig == SIGSEGV && si->si_code == SEGV_ACCERR) ? "SEGV_ACCERR" : "other";
  snprintf(line, sizeof line, "\n[segv_info] signal %d si_code %d (%s) si_addr %p si_pid %d own pid %d\n", sig, si->si_code, kind,
           si->si_addr, (si->si_code == SI_USER || si->si_code == SI_TKILL) ? (int)si->si_pid : -1, (int)getpid());
  put(line);
  void* frames[48];
  const int n = backtrace(frames, 48);
  backtrace_symbols_fd(frames, n, 2);
  put("[segv_info] end of native backtrace\n");
  /* hand over: the previous handler (faulthandler) prints the Python frames and re-raises */
  struct sigaction* p = &g_prev[sig];
  if (p->sa_flags & SA_SIGINFO) {
    if (p->sa_sigaction) { p->sa_sigaction(sig, si, uctx); return; }
  } else if (p->sa_handler != SIG_DFL && p->sa_handler != SIG_IGN) {
    p->sa_handler(sig);
    return;
  }
  signal(sig, SIG_DFL);
  raise(sig);
}

int segv_info_install(void) {
  void* warm[2];
  backtrace(warm, 2); /* loads libgcc now, not inside the handler */
  const int sigs[3] = {SIGSEGV, SIGBUS, SIGABRT};
  for (int k = 0; k < 3; k++) {
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = handler;
    sa.sa_flags = SA_SIGINFO | SA_ONSTACK | SA_NODEFER;
    sigemptyset(&sa.sa_mask);
    if (sigaction(sigs[k], &sa, &g_prev[sigs[k]]) != 0) return -1;
  }
  return 0;
}
