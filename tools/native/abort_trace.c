/* Diagnostic: print a native backtrace when the process aborts (SIGABRT / SIGSEGV).
 * Built by tools/stress_threads.py and loaded with ctypes; not part of the library. */
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static void handler(int sig) {
  void *frames[64];
  const char msg[] = "\n== native backtrace of the failing thread ==\n";
  ssize_t ignored = write(2, msg, sizeof(msg) - 1);
  (void)ignored;
  int n = backtrace(frames, 64);
  backtrace_symbols_fd(frames, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}

void abort_trace_install(void) {
  signal(SIGABRT, handler);
  signal(SIGSEGV, handler);
}
