// LD_PRELOAD helper for debugging: prints the native backtrace of the thread that raises SIGABRT/SIGSEGV/SIGBUS.
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>
#include <sys/syscall.h>
static void handler(int sig, siginfo_t* si, void* ctx) {
  (void)ctx;
  char msg[128];
  int n = snprintf(msg, sizeof msg, "\n[abrt_bt] signal %d (si_code %d, addr %p) in tid %ld\n", sig, si ? si->si_code : 0,
                   si ? si->si_addr : 0, (long)syscall(SYS_gettid));
  write(2, msg, n);
  void* bt[96];
  int k = backtrace(bt, 96);
  backtrace_symbols_fd(bt, k, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}
__attribute__((constructor)) static void init(void) {
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = handler;
  sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
  sigaction(SIGABRT, &sa, 0);
  sigaction(SIGSEGV, &sa, 0);
  sigaction(SIGBUS, &sa, 0);
  void* bt[4];
  backtrace(bt, 4);  // loads libgcc now, not inside the handler
}
