/* LD_PRELOAD helper for crash hunting on the GPU box (no gdb there): prints a native backtrace of the faulting thread on
 * SIGSEGV / SIGBUS / SIGABRT and the fault address, then re-raises. Addresses resolve offline with
 * llvm-addr2line against the same in-tree .so files.   build: gcc -O1 -g -shared -fPIC segv_bt.c -o libsegv_bt.so */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

static void handler(int sig, siginfo_t* si, void* uc) {
  (void)uc;
  char line[128];
  int n = snprintf(line, sizeof line, "\n=== segv_bt: signal %d, fault address %p ===\n", sig, si ? si->si_addr : 0);
  (void)!write(2, line, (size_t)n);
  void* frames[64];
  int depth = backtrace(frames, 64);
  backtrace_symbols_fd(frames, depth, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}

__attribute__((constructor)) static void install(void) {
  static char stack[1 << 16];
  stack_t ss = {.ss_sp = stack, .ss_size = sizeof stack, .ss_flags = 0};
  sigaltstack(&ss, 0);
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = handler;
  sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
  sigaction(SIGSEGV, &sa, 0);
  sigaction(SIGBUS, &sa, 0);
  sigaction(SIGABRT, &sa, 0);
}
