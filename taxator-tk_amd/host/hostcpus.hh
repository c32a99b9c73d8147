// taxator-tk_amd/host/hostcpus.hh -- how many host threads this process can really run: the smaller of the scheduler
// affinity mask and the cgroup CPU quota.  std::thread::hardware_concurrency() reports the machine (256 on an MI355X
// host of which a container may own 16): sizing thread teams by it means hundreds of threads per parse phase.
#pragma once
#include <sched.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <thread>

namespace taxator {

inline unsigned usable_cpus() {
  unsigned n = std::max(1u, std::thread::hardware_concurrency());
  cpu_set_t set;
  CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof(set), &set) == 0) {
    const int c = CPU_COUNT(&set);
    if (c > 0) n = std::min(n, (unsigned)c);
  }
  // cgroup v2: "<quota> <period>" or "max <period>"; cgroup v1: cpu.cfs_quota_us / cpu.cfs_period_us
  long long quota = -1, period = 0;
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
    char q[32] = {0};
    if (fscanf(f, "%31s %lld", q, &period) == 2 && q[0] != 'm') quota = atoll(q);
    fclose(f);
  } else {
    if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &quota) != 1) quota = -1; fclose(g); }
    if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &period) != 1) period = 0; fclose(g); }
  }
  if (quota > 0 && period > 0) n = std::min(n, (unsigned)std::max(1ll, (quota + period - 1) / period));
  return std::max(1u, n);
}

}  // namespace taxator
