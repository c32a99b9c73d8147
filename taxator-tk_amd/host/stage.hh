// taxator-tk_amd/host/stage.hh -- "where is this process?" for whoever watches it from outside.
// The reference's taxator has nothing to wait for but its own threads; this one waits for a GPU runtime.  A driver that
// seems stuck is asked with SIGUSR1 (report and carry on) or SIGTERM (report and leave with 143): it writes the stage the
// host code reached last and the blocking runtime call the engine entered last (rpa_debug_where) to stderr, using only
// async-signal-safe calls (plus glibc's backtrace, which is safe once libgcc is loaded: install_stage_reporter calls it once
// up front).  TAXATOR_WATCHDOG_S=n arms the same report from inside after n seconds (alarm), exit code 124.
#pragma once
#include <atomic>
#include <csignal>
#include <cstdlib>
#include <cstring>
#include <execinfo.h>
#include <unistd.h>

#include "../../include/taxator_rpa.h"

namespace taxator {

inline std::atomic<const char*>& stage_slot() {
  static std::atomic<const char*> s{"start"};
  return s;
}
inline void stage(const char* text) { stage_slot().store(text, std::memory_order_relaxed); }

namespace detail {
inline void put(const char* s) { if (s) { ssize_t r = write(2, s, strlen(s)); (void)r; } }
inline void report(int sig) {
  put("taxator: stage: "); put(stage_slot().load(std::memory_order_relaxed));
  put("; engine: "); put(rpa_debug_where());
  put(sig == SIGALRM ? " (watchdog expired)\n" : "\n");
  // the stack of the thread that took the signal (a signal sent to the process goes to the main thread when that one does
  // not block it): module + exported symbol + offset per frame, straight to stderr
  void* frames[48];
  const int n = backtrace(frames, 48);
  backtrace_symbols_fd(frames, n, 2);
  if (sig == SIGTERM) _exit(143);
  if (sig == SIGALRM) _exit(124);
}
}  // namespace detail

inline void install_stage_reporter() {
  { void* warm[2]; (void)backtrace(warm, 2); }  // loads libgcc's unwinder now, not inside a signal handler
  struct sigaction sa;
  memset(&sa, 0, sizeof(sa));
  sa.sa_handler = detail::report;
  sigemptyset(&sa.sa_mask);
  sigaction(SIGUSR1, &sa, nullptr);
  sigaction(SIGTERM, &sa, nullptr);
  if (const char* v = getenv("TAXATOR_WATCHDOG_S")) {
    const int n = atoi(v);
    if (n > 0) { sigaction(SIGALRM, &sa, nullptr); alarm((unsigned)n); }
  }
}

}  // namespace taxator
