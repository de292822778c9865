// host_pool.hpp -- part of libbpmi (host only; no GPU call).  A small pool of worker threads for the library's data-parallel HOST
// loops (the range-proof algebra of rp_algebra_host.hpp, the seeded blinding vectors of transcript_host.hpp): those loops are 0.2-0.5 ms
// of arithmetic per call on eight threads, and creating and joining eight std::threads costs as much again (round 4: a 128 x 64-bit
// proof calls them three times).  run(n, f) calls f(0) .. f(n - 1), f(0) on the calling thread, and returns when all are done.
//   * one pool per process, created at first use, never destroyed (its threads sleep on a condition variable; the process's exit
//     ends them); a child of fork() finds the pool's owner is another process and starts its own;
//   * one run at a time: a second caller (another host thread of the application) does not wait for the pool, it creates
//     threads of its own as the code did before -- no lock is held while user work runs, nothing can deadlock on it.
#pragma once
#include <unistd.h>

#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace hostpool {

struct Pool {
  std::vector<std::thread> th;
  std::mutex mu;
  std::condition_variable cv_work, cv_done;
  const std::function<void(int)> *job = nullptr;
  int njobs = 0, next = 0, running = 0;
  bool busy = false;
  void worker() {
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      cv_work.wait(lk, [this] { return job != nullptr && next < njobs; });
      const int t = next++;
      const std::function<void(int)> *j = job;
      lk.unlock();
      (*j)(t);
      lk.lock();
      if (--running == 0) cv_done.notify_all();
    }
  }
};
static Pool *g_pool = nullptr;
static pid_t g_pool_pid = 0;
static std::mutex g_pool_mu;

static inline void run_on_new_threads(int n, const std::function<void(int)> &f) {
  std::vector<std::thread> th;
  for (int t = 1; t < n; t++) th.emplace_back(f, t);
  f(0);
  for (auto &x : th) x.join();
}
#define BPMI_POOL_MAX 32
static inline void run(int n, const std::function<void(int)> &f) {
  if (n <= 1) { f(0); return; }
  if (n > BPMI_POOL_MAX) { run_on_new_threads(n, f); return; }
  Pool *p;
  {
    std::lock_guard<std::mutex> g(g_pool_mu);
    if (!g_pool || g_pool_pid != getpid()) { g_pool = new Pool(); g_pool_pid = getpid(); }      // (a forked child leaves the parent's object alone)
    p = g_pool;
  }
  {
    std::unique_lock<std::mutex> lk(p->mu);
    if (p->busy) { lk.unlock(); run_on_new_threads(n, f); return; }
    p->busy = true;
    while ((int)p->th.size() < n - 1) p->th.emplace_back([p] { p->worker(); });
    p->job = &f; p->njobs = n; p->next = 1; p->running = n - 1;
  }
  p->cv_work.notify_all();
  f(0);
  std::unique_lock<std::mutex> lk(p->mu);
  p->cv_done.wait(lk, [p] { return p->running == 0; });
  p->job = nullptr; p->njobs = 0; p->busy = false;
}

}  // namespace hostpool
