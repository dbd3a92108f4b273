// group.hip - kpdi_group: several MI355X driven from ONE thread of ONE process (include/kpdi.h, "a group of contexts").
//
// The reference's dictionary_indexing() is one call in one interpreter (signals/ebsd.py:1827-1984; the chunk loop of
// indexing/_dictionary_indexing.py:100-128 never leaves the process).  A group keeps that call shape on a node with
// several GPUs: one kpdi_ctx per device, one host thread per member (member 0 runs on the caller's thread), every entry
// point the per-context one fanned out, every dictionary chunk handed to the member(s) group_assign.h names (whole chunks
// wherever the call is chunked), and ONE merged result.
//
// Host threads.  Every member has a worker thread with a FIFO of jobs.  Calls that configure the group or collect a
// result are fanned out and JOINED (run_all: member 0 on the caller's thread, behind whatever its worker still holds).
// Dictionary chunks are different: a push only QUEUES the pieces on the workers of the members that take them and
// returns - the upload / generation / sweep of member i's chunk runs while the caller fetches (`.compute()`s,
// simulates, reads) the next one, which goes to member i + 1.  A worker's failure is kept and reported by the next
// joining call.
//
// Host code only: everything that touches a device goes through the per-context C ABI (api.hip) or the few hooks of
// group_hooks.h (in-process RCCL communicator, peer-copy gather).
#include "../../include/kpdi.h"
#include "group_hooks.h"
#include "group_assign.h"

#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <algorithm>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

// one host thread per member: kernel launches, uploads and RCCL calls of the members run concurrently (a push is ~10
// launches = ~0.1 ms of host time; one thread driving 8 devices in turn would serialise 0.8 ms per chunk)
struct Worker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<std::function<int()>> q;
  bool running = false, stop = false;
  int rc = 0;       // first failure since the last collect()
  std::string err;

  void loop() {
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      cv.wait(lk, [&] { return !q.empty() || stop; });
      if (q.empty()) return;  // (stop: after the queue has drained)
      std::function<int()> j = std::move(q.front());
      q.pop_front();
      running = true;
      lk.unlock();
      cv.notify_all();  // (a poster waiting for room)
      const int r = j();
      std::string e = r ? kpdi::thread_error() : "";  // (thread-local: fetched on THIS thread)
      j = nullptr;  // what the job captured (a copy of the rotations, a ticket) goes before `idle` is announced
      lk.lock();
      if (r && !rc) {
        rc = r;
        err = std::move(e);
      }
      running = false;
      cv.notify_all();
    }
  }
  // `max_queued` > 0: wait while that many jobs are already queued (back-pressure: a caller that produces chunks
  // faster than the member sweeps them would otherwise queue - and keep alive - the whole dictionary)
  void post(std::function<int()> j, size_t max_queued = 0) {
    {
      std::unique_lock<std::mutex> lk(mu);
      if (max_queued) cv.wait(lk, [&] { return q.size() < max_queued; });
      q.push_back(std::move(j));
    }
    cv.notify_all();
  }
  // wait until the worker is idle; its first failure since the last collect (and forget it)
  int collect(std::string *e) {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return q.empty() && !running; });
    const int r = rc;
    *e = std::move(err);
    rc = 0;
    err.clear();
    return r;
  }
};

}  // namespace

struct kpdi_group {
  int n = 0;
  std::vector<int> dev;
  std::vector<kpdi_ctx *> ctx;
  std::vector<std::unique_ptr<Worker>> workers;  // one per member when n > 1 (a group of one runs on the caller's thread)
  int gather = KPDI_GATHER_NONE;
  std::string describe;
  bool exact64 = false;       // the problem is KPDI_COMPUTE_F64 (no async hand-over)
  int64_t npix = 0;           // detector pixels of the problem: a chunk's rows are npix elements apart
  std::vector<kpdi::ListsView> views;
  // chunk assignment (group_assign.h)
  int64_t n_total = 0;                 // dictionary size announced by kpdi_group_set_dictionary_size, 0 = unknown
  std::vector<int64_t> load, held;     // patterns every member has taken in this sweep / holds resident
  // borrowed host chunks: ticket -> pieces still to be consumed
  std::mutex tmu;
  std::condition_variable tcv;
  std::map<int64_t, int> open_tickets;
  int64_t last_ticket = 0;
  bool comm_broken = false;   // a finalize over the RCCL communicator failed half-way: the ranks are out of step
};

namespace {

std::string member_error(const kpdi_group *g, int who, const std::string &err) {
  char head[96];
  snprintf(head, sizeof head, "device %d (group member %d of %d): ", g->dev[who], who, g->n);
  return head + err;
}

// fn(i, ctx_i) on every member concurrently, joined; the first failure (lowest member) is reported with the member
// named - including one a QUEUED chunk of that member ran into earlier (its worker kept it)
template <typename F>
int run_all(kpdi_group *g, F fn) {
  if (g->n == 1) return fn(0, g->ctx[0]);
  for (int i = 1; i < g->n; ++i) g->workers[i]->post([&fn, g, i] { return fn(i, g->ctx[i]); });
  std::string err;
  int rc = g->workers[0]->collect(&err);  // member 0: on this thread, behind its queued chunks
  int who = 0;
  if (!rc) {
    rc = fn(0, g->ctx[0]);
    if (rc) err = kpdi::thread_error();
  }
  for (int i = 1; i < g->n; ++i) {
    std::string e;
    const int r = g->workers[i]->collect(&e);
    if (r && !rc) {
      rc = r;
      err = e;
      who = i;
    }
  }
  if (rc) return kpdi::fail_msg(rc, "%s", member_error(g, who, err).c_str());
  return rc;
}

// every worker idle; the first failure a queued chunk ran into
int join_workers(kpdi_group *g) {
  int rc = 0, who = 0;
  std::string err;
  for (int i = 0; i < g->n && g->n > 1; ++i) {
    std::string e;
    const int r = g->workers[i]->collect(&e);
    if (r && !rc) {
      rc = r;
      err = e;
      who = i;
    }
  }
  if (rc) return kpdi::fail_msg(rc, "%s", member_error(g, who, err).c_str());
  return KPDI_OK;
}

int bad_group() { return kpdi::fail_msg(KPDI_EINVAL, "group is NULL"); }

// One small all-gather through the members' new communicator, every member on a thread of its own, each under a timeout
// ($KPDI_COMM_TIMEOUT seconds, default 60): a communicator whose bootstrap succeeded can still hang in its first
// collective (peer access, IPC handles) - found here, where the group can still take the peer-copy gather, not in the
// finalize of the first sweep.  The multi-process form does the same in Communicator.attach (kikuchipy_amd/parallel.py).
int comm_selftest_all(kpdi_group *g, std::string *why) {
  double seconds = 60.0;
  if (const char *e = getenv("KPDI_COMM_TIMEOUT")) seconds = std::max(0.001, atof(e));
  const int timeout_ms = (int)std::min(seconds * 1000.0, 2.0e9);
  std::vector<int> rc(g->n, 0);
  std::vector<std::string> err(g->n);
  std::vector<std::thread> th;
  for (int i = 0; i < g->n; ++i)
    th.emplace_back([&, i] {
      rc[i] = getenv("KPDI_GROUP_SELFTEST_FAIL") ? kpdi::fail_msg(KPDI_ECOMM, "injected failure (KPDI_GROUP_SELFTEST_FAIL)")
                                                 : kpdi_comm_selftest(g->ctx[i], 65536, timeout_ms);
      if (rc[i]) err[i] = kpdi::thread_error();
    });
  for (auto &t : th) t.join();
  for (int i = 0; i < g->n; ++i)
    if (rc[i]) {
      *why = member_error(g, i, err[i]);
      return rc[i];
    }
  return KPDI_OK;
}

// a new sweep starts: nobody has taken anything yet
void reset_loads(kpdi_group *g) { g->load.assign(g->n, 0); }

// patterns below which a piece of a chunk is not worth a launch set of its own: two tile rounds of a member's sweep
// (KPDI_GROUP_MIN_PIECE overrides; only consulted while the dictionary size is unknown)
int64_t min_piece(const kpdi_group *g) {
  if (const char *e = getenv("KPDI_GROUP_MIN_PIECE")) return std::max<int64_t>(1, atoll(e));
  return 2 * kpdi::sweep_round_rows(g->ctx[0]);
}

// the members' lists to the root: RCCL members take part in the all-gather, peer-copy members publish their lists and
// the root queues the copies.  Afterwards a finalize of member 0 yields the merged result.
int gather_to_root(kpdi_group *g) {
  if (g->gather != KPDI_GATHER_P2P) return KPDI_OK;
  g->views.assign(g->n, kpdi::ListsView{});
  int rc = run_all(g, [g](int i, kpdi_ctx *c) { return kpdi::member_lists_ready(c, &g->views[i]); });
  if (rc) return rc;
  if (g->views[0].n == 0) return KPDI_OK;
  hipEvent_t read_done = nullptr;
  rc = kpdi::root_gather_p2p(g->ctx[0], g->views.data(), g->n, &read_done);
  if (rc) return rc;
  for (int i = 1; i < g->n; ++i) kpdi::member_lists_borrowed(g->ctx[i], read_done);
  return KPDI_OK;
}

// finalize: `root(ctx0)` hands the merged result over; with an RCCL communicator the other members join its all-gather.
// Everything that can refuse the root's call WITHOUT touching a device (arguments, result slots, mode) is checked
// first (kpdi::finalize_precheck): once a member has queued its half of the collective the root must queue its own,
// or the next finalize would pair the root's all-gather with the members' stale one.
template <typename F>
int finalize_all(kpdi_group *g, int kind, F root) {
  if (g->comm_broken)
    return kpdi::fail_msg(KPDI_ECOMM, "an earlier finalize of this group failed between the members' halves of the RCCL "
                                      "all-gather: its communicator is out of step - destroy the group and create a new one");
  int rc = join_workers(g);  // (queued chunks: a failure there is reported before anything is gathered)
  if (rc) return rc;
  rc = kpdi::finalize_precheck(g->ctx[0], kind);
  if (rc) return rc;
  rc = gather_to_root(g);
  if (rc) {
    kpdi::gather_abandon(g->ctx[0]);
    return rc;
  }
  if (g->gather == KPDI_GATHER_RCCL && g->n > 1) {
    rc = run_all(g, [&root](int i, kpdi_ctx *c) { return i == 0 ? root(c) : kpdi::finalize_participate(c); });
    if (rc) g->comm_broken = true;
    return rc;
  }
  rc = root(g->ctx[0]);
  if (rc) kpdi::gather_abandon(g->ctx[0]);  // (lists peer-copied for THIS finalize must not be merged by a later one)
  return rc;
}

}  // namespace

extern "C" {

int kpdi_group_chunk_share(int64_t n_chunk, int i, int n_dev, int64_t *start, int64_t *end) {
  if (!start || !end || n_dev < 1 || i < 0 || i >= n_dev || n_chunk < 0)
    return kpdi::fail_msg(KPDI_EINVAL, "kpdi_group_chunk_share: bad arguments");
  kpdi::group_share(n_chunk, i, n_dev, start, end);
  return KPDI_OK;
}

int kpdi_group_assign_chunk(int n_dev, int64_t n_total, int64_t min_piece, int64_t *load, int64_t n_chunk, int max_pieces,
                            int *member_out, int64_t *row0_out, int64_t *rows_out, int *n_pieces) {
  if (n_dev < 1 || !load || n_chunk <= 0 || n_total < 0 || max_pieces < 0 || !n_pieces ||
      (max_pieces > 0 && (!member_out || !row0_out || !rows_out)))
    return kpdi::fail_msg(KPDI_EINVAL, "kpdi_group_assign_chunk: bad arguments");
  const std::vector<kpdi::ChunkPiece> pieces = kpdi::group_assign_chunk(n_dev, n_total, min_piece, load, n_chunk);
  *n_pieces = (int)pieces.size();
  for (int i = 0; i < (int)pieces.size() && i < max_pieces; ++i) {
    member_out[i] = pieces[i].member;
    row0_out[i] = pieces[i].row0;
    rows_out[i] = pieces[i].rows;
  }
  return KPDI_OK;
}

int kpdi_group_create(const int *device_ids, int n_dev, int gather, kpdi_group **out) {
  if (!out) return kpdi::fail_msg(KPDI_EINVAL, "out is NULL");
  *out = nullptr;
  if (!device_ids || n_dev < 1) return kpdi::fail_msg(KPDI_EINVAL, "a group needs at least one device");
  if (n_dev > 64) return kpdi::fail_msg(KPDI_EINVAL, "a group holds at most 64 members");
  if (gather < KPDI_GATHER_AUTO || gather > KPDI_GATHER_P2P) return kpdi::fail_msg(KPDI_EINVAL, "unknown gather mode %d", gather);
  std::unique_ptr<kpdi_group> g(new kpdi_group());
  g->n = n_dev;
  g->dev.assign(device_ids, device_ids + n_dev);
  auto destroy_members = [&] {
    for (kpdi_ctx *c : g->ctx) kpdi_destroy(c);
  };
  for (int i = 0; i < n_dev; ++i) {
    kpdi_ctx *c = nullptr;
    const int rc = kpdi_create(device_ids[i], &c);
    if (rc) {
      const std::string e = kpdi::thread_error();
      destroy_members();
      return kpdi::fail_msg(rc, "group member %d: %s", i, e.c_str());
    }
    g->ctx.push_back(c);
  }
  bool duplicates = false;
  for (int i = 0; i < n_dev; ++i)
    for (int j = 0; j < i; ++j) duplicates |= device_ids[i] == device_ids[j];
  std::string note;
  int mode = gather;
  if (mode == KPDI_GATHER_AUTO) {
    const char *e = getenv("KPDI_GATHER");
    if (e && !strcmp(e, "p2p")) mode = KPDI_GATHER_P2P;
    else if (e && !strcmp(e, "rccl")) mode = KPDI_GATHER_RCCL;
    else if (e && *e) {
      destroy_members();
      return kpdi::fail_msg(KPDI_EINVAL, "KPDI_GATHER must be \"rccl\" or \"p2p\", not \"%s\"", e);
    }
  }
  if (mode == KPDI_GATHER_AUTO) {
    if (n_dev == 1) mode = KPDI_GATHER_NONE;
    else if (duplicates) mode = KPDI_GATHER_P2P, note = " (a device appears twice: RCCL refuses duplicate devices)";
    else {
      // a communicator of the process's own: no unique id to pass around, no sockets, no environment
      if (kpdi::comm_init_all(g->ctx.data(), n_dev) == KPDI_OK) {
        std::string why;
        if (comm_selftest_all(g.get(), &why) == KPDI_OK) {
          mode = KPDI_GATHER_RCCL;
        } else {  // (every member drops: a communicator that only some members hold would hang the first finalize)
          for (kpdi_ctx *c : g->ctx) (void)kpdi_comm_drop(c);
          mode = KPDI_GATHER_P2P, note = " (RCCL's first all-gather failed: " + why + ")";
        }
      } else {
        mode = KPDI_GATHER_P2P, note = std::string(" (RCCL unavailable: ") + kpdi::thread_error() + ")";
      }
    }
  } else if (mode == KPDI_GATHER_RCCL) {
    int rc = kpdi::comm_init_all(g->ctx.data(), n_dev);  // (also with one device: keeps the path testable)
    std::string e = rc ? kpdi::thread_error() : "";
    if (!rc) rc = comm_selftest_all(g.get(), &e);  // asked for by name: a failure is an error, not a fallback
    if (rc) {
      for (kpdi_ctx *c : g->ctx) (void)kpdi_comm_drop(c);
      destroy_members();
      return kpdi::fail_msg(rc, "KPDI_GATHER_RCCL: %s%s", e.c_str(),
                            duplicates ? " - several members share a device; use KPDI_GATHER_P2P for that" : "");
    }
  } else if (n_dev == 1) {
    mode = KPDI_GATHER_NONE;  // P2P asked for, nothing to gather
  }
  g->gather = mode;
  if (mode == KPDI_GATHER_P2P) {
    // direct xGMI copies where the devices allow it (hipMemcpyPeerAsync works without, through the host)
    for (int i = 1; i < n_dev; ++i)
      if (device_ids[i] != device_ids[0] && hipSetDevice(device_ids[0]) == hipSuccess) {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, device_ids[0], device_ids[i]) == hipSuccess && can)
          (void)hipDeviceEnablePeerAccess(device_ids[i], 0);  // (hipErrorPeerAccessAlreadyEnabled is fine)
        (void)hipGetLastError();
      }
  }
  g->describe = std::to_string(n_dev) + (n_dev == 1 ? " device [" : " devices [");
  for (int i = 0; i < n_dev; ++i) g->describe += (i ? "," : "") + std::to_string(device_ids[i]);
  g->describe += std::string("], gather ") +
                 (mode == KPDI_GATHER_RCCL ? "rccl" : mode == KPDI_GATHER_P2P ? "p2p" : "none") + note;
  g->workers.resize(n_dev);
  g->load.assign(n_dev, 0);
  g->held.assign(n_dev, 0);
  for (int i = 0; i < n_dev && n_dev > 1; ++i) {
    g->workers[i].reset(new Worker());
    Worker *w = g->workers[i].get();
    w->th = std::thread([w] { w->loop(); });
  }
  *out = g.release();
  return KPDI_OK;
}

int kpdi_group_destroy(kpdi_group *g) {
  if (!g) return KPDI_OK;
  for (int i = 0; i < g->n && g->n > 1; ++i) {
    Worker *w = g->workers[i].get();
    {
      std::lock_guard<std::mutex> lk(w->mu);
      w->stop = true;  // (the loop leaves once its queue has drained)
    }
    w->cv.notify_all();
    w->th.join();
  }
  // (every stream is synchronised before any member goes: a member's lists may still be read by the root's copies)
  for (kpdi_ctx *c : g->ctx) (void)kpdi_synchronize(c);
  for (kpdi_ctx *c : g->ctx) kpdi_destroy(c);
  delete g;
  return KPDI_OK;
}

int kpdi_group_size(const kpdi_group *g) { return g ? g->n : 0; }
int kpdi_group_gather(const kpdi_group *g) { return g ? g->gather : KPDI_GATHER_NONE; }
const char *kpdi_group_describe(const kpdi_group *g) { return g ? g->describe.c_str() : ""; }
kpdi_ctx *kpdi_group_member(kpdi_group *g, int i) { return g && i >= 0 && i < g->n ? g->ctx[i] : nullptr; }

int kpdi_group_synchronize(kpdi_group *g) {
  if (!g) return bad_group();
  return run_all(g, [](int, kpdi_ctx *c) { return kpdi_synchronize(c); });
}

int kpdi_group_set_problem(kpdi_group *g, int sy, int sx, const uint8_t *signal_mask, int metric, int compute_dtype,
                           int keep_n) {
  if (!g) return bad_group();
  const int rc = run_all(g, [=](int, kpdi_ctx *c) { return kpdi_set_problem(c, sy, sx, signal_mask, metric, compute_dtype, keep_n); });
  if (!rc) {
    g->exact64 = compute_dtype == KPDI_COMPUTE_F64;
    g->npix = (int64_t)sy * sx;
    reset_loads(g);
  }
  return rc;
}

int kpdi_group_set_keep_n(kpdi_group *g, int keep_n) {
  if (!g) return bad_group();
  reset_loads(g);
  return run_all(g, [=](int, kpdi_ctx *c) { return kpdi_set_keep_n(c, keep_n); });
}

int kpdi_group_set_dictionary_size(kpdi_group *g, int64_t n_total) {
  if (!g) return bad_group();
  if (n_total < 0) return kpdi::fail_msg(KPDI_EINVAL, "dictionary size must be >= 0 (0 = unknown)");
  g->n_total = n_total;
  reset_loads(g);
  return KPDI_OK;
}

int kpdi_group_set_experimental(kpdi_group *g, const void *patterns, int dtype, int64_t m_all, const uint8_t *nav_mask) {
  if (!g) return bad_group();
  // ONE host copy, n uploads in parallel (every device has its own link to the host); the members synchronise their
  // upload before they return, so the caller's buffer is free again
  reset_loads(g);
  return run_all(g, [=](int, kpdi_ctx *c) {
    const int rc = kpdi_set_experimental(c, patterns, dtype, m_all, nav_mask);
    return rc ? rc : kpdi_synchronize(c);
  });
}

int kpdi_group_set_experimental_dev(kpdi_group *g, const void *const *d_patterns, int dtype, int64_t m_all,
                                    const uint8_t *nav_mask) {
  if (!g) return bad_group();
  if (!d_patterns) return kpdi::fail_msg(KPDI_EINVAL, "d_patterns is NULL");
  reset_loads(g);
  return run_all(g, [=](int i, kpdi_ctx *c) { return kpdi_set_experimental_dev(c, d_patterns[i], dtype, m_all, nav_mask); });
}

int64_t kpdi_group_n_experimental(kpdi_group *g) { return g ? kpdi_n_experimental(g->ctx[0]) : 0; }

int kpdi_group_remove_static_background(kpdi_group *g, const float *static_bg, int operation, int scale_bg) {
  if (!g) return bad_group();
  return run_all(g, [=](int, kpdi_ctx *c) { return kpdi_remove_static_background(c, static_bg, operation, scale_bg); });
}

int kpdi_group_remove_dynamic_background(kpdi_group *g, int operation, int filter_domain, double std, double truncate) {
  if (!g) return bad_group();
  return run_all(g, [=](int, kpdi_ctx *c) { return kpdi_remove_dynamic_background(c, operation, filter_domain, std, truncate); });
}

int kpdi_group_get_experimental(kpdi_group *g, void *patterns_out) {
  if (!g) return bad_group();
  const int rc = join_workers(g);
  return rc ? rc : kpdi_get_experimental(g->ctx[0], patterns_out);
}

}  // extern "C"

namespace {

int chunk_row_bytes(kpdi_group *g, int dtype, size_t *row_bytes) {
  const size_t es = kpdi_dtype_size(dtype);
  if (es == 0) return kpdi::fail_msg(KPDI_EINVAL, "unknown dtype %d", dtype);
  if (g->npix <= 0) return kpdi::fail_msg(KPDI_EINVAL, "kpdi_group_set_problem has not been called");
  *row_bytes = (size_t)g->npix * es;
  return KPDI_OK;
}

int check_chunk(const void *rows, int64_t n_chunk) {
  if (!rows) return kpdi::fail_msg(KPDI_EINVAL, "pointer is NULL");
  if (n_chunk <= 0) return kpdi::fail_msg(KPDI_EINVAL, "dictionary chunk must hold at least one pattern");
  return KPDI_OK;
}

constexpr size_t MAX_QUEUED = 2;  // chunks a member may have waiting behind the one it works on

// A chunk of the SWEEP: its pieces (group_assign.h) are queued on their members' workers and the call returns; `job(ctx,
// piece)` runs there.  *ticket (may be NULL): the number under which kpdi_group_chunks_consumed reports that every
// piece's job has returned - for pushes whose jobs read the caller's buffer.
template <typename J>
int queue_chunk(kpdi_group *g, int64_t n_chunk, J job, int64_t *ticket) {
  const std::vector<kpdi::ChunkPiece> pieces =
      kpdi::group_assign_chunk(g->n, g->n_total, g->n_total > 0 ? 1 : min_piece(g), g->load.data(), n_chunk);
  int64_t t = 0;
  if (ticket) {
    std::lock_guard<std::mutex> lk(g->tmu);
    t = ++g->last_ticket;
    g->open_tickets[t] = (int)pieces.size();
    *ticket = t;
  }
  auto done = [g, t] {
    if (!t) return;
    {
      std::lock_guard<std::mutex> lk(g->tmu);
      auto it = g->open_tickets.find(t);
      if (it != g->open_tickets.end() && --it->second == 0) g->open_tickets.erase(it);
    }
    g->tcv.notify_all();
  };
  if (g->n == 1) {  // a group of one: on the caller's thread
    const int rc = job(g->ctx[0], pieces[0]);
    done();
    return rc;
  }
  for (const kpdi::ChunkPiece &pc : pieces) {
    kpdi_ctx *c = g->ctx[pc.member];
    g->workers[pc.member]->post([job, done, c, pc] {
      const int rc = job(c, pc);
      done();
      return rc;
    }, MAX_QUEUED);
  }
  return KPDI_OK;
}

void wait_ticket(kpdi_group *g, int64_t t) {
  std::unique_lock<std::mutex> lk(g->tmu);
  g->tcv.wait(lk, [&] { return g->open_tickets.find(t) == g->open_tickets.end(); });
}

// A chunk that stays RESIDENT: the same assignment (over the members' resident patterns), joined before returning
template <typename J>
int hold_chunk(kpdi_group *g, int64_t n_chunk, J job) {
  int rc = join_workers(g);
  if (rc) return rc;
  const std::vector<kpdi::ChunkPiece> pieces =
      kpdi::group_assign_chunk(g->n, g->n_total, g->n_total > 0 ? 1 : min_piece(g), g->held.data(), n_chunk);
  // A member may take SEVERAL pieces of one chunk: what exceeds the announced dictionary size goes whole to the
  // least-loaded member (group_assign.h), which may hold an earlier, non-adjacent piece already - e.g. a chunk of 6
  // against an announced size of 4 on two members: (0: 0-1) (1: 2-3) (0: 4-5).  Every piece of a member runs in its job,
  // in row order.
  std::vector<std::vector<const kpdi::ChunkPiece *>> mine(g->n);
  for (const kpdi::ChunkPiece &pc : pieces) mine[pc.member].push_back(&pc);
  return run_all(g, [&](int i, kpdi_ctx *c) {
    for (const kpdi::ChunkPiece *pc : mine[i]) {
      const int r = job(c, *pc);
      if (r) return r;
    }
    return (int)KPDI_OK;
  });
}

}  // namespace

extern "C" {

int kpdi_group_push_dictionary_chunk_borrowed(kpdi_group *g, const void *patterns, int dtype, int64_t n_chunk,
                                              int64_t global_start, int64_t *ticket) {
  if (!g) return bad_group();
  if (!ticket) return kpdi::fail_msg(KPDI_EINVAL, "ticket is NULL");
  size_t rb = 0;
  int rc = chunk_row_bytes(g, dtype, &rb);
  if (rc) return rc;
  rc = check_chunk(patterns, n_chunk);
  if (rc) return rc;
  // every member uploads its rows through its own staging buffers and copy stream (its own link to the host); a piece
  // counts as consumed when that upload has read it - its sweep runs on
  return queue_chunk(g, n_chunk, [=](kpdi_ctx *c, const kpdi::ChunkPiece &pc) {
    return kpdi_push_dictionary_chunk(c, (const char *)patterns + (size_t)pc.row0 * rb, dtype, pc.rows, global_start + pc.row0);
  }, ticket);
}

int kpdi_group_chunks_consumed(kpdi_group *g, int64_t *ticket) {
  if (!g) return bad_group();
  if (!ticket) return kpdi::fail_msg(KPDI_EINVAL, "ticket is NULL");
  std::lock_guard<std::mutex> lk(g->tmu);
  *ticket = g->open_tickets.empty() ? g->last_ticket : g->open_tickets.begin()->first - 1;
  return KPDI_OK;
}

int kpdi_group_push_dictionary_chunk(kpdi_group *g, const void *patterns, int dtype, int64_t n_chunk,
                                     int64_t global_start) {
  int64_t t = 0;
  const int rc = kpdi_group_push_dictionary_chunk_borrowed(g, patterns, dtype, n_chunk, global_start, &t);
  if (rc) return rc;
  wait_ticket(g, t);  // the caller's buffer is free again (a failure of the queued work is reported by the next joining call)
  return KPDI_OK;
}

int kpdi_group_push_dictionary_chunk_dev(kpdi_group *g, const void *const *d_patterns, int dtype, const int64_t *n_chunk,
                                         const int64_t *global_start) {
  if (!g) return bad_group();
  if (!d_patterns || !n_chunk || !global_start) return kpdi::fail_msg(KPDI_EINVAL, "NULL argument");
  for (int i = 0; i < g->n; ++i) g->load[i] += std::max<int64_t>(n_chunk[i], 0);  // (the caller's own partition)
  return run_all(g, [=](int i, kpdi_ctx *c) {
    if (n_chunk[i] <= 0) return (int)KPDI_OK;
    return kpdi_push_dictionary_chunk_dev(c, d_patterns[i], dtype, n_chunk[i], global_start[i]);
  });
}

int kpdi_group_set_master_pattern(kpdi_group *g, const void *upper, const void *lower, int dtype, int npx, int npy) {
  if (!g) return bad_group();
  return run_all(g, [=](int, kpdi_ctx *c) { return kpdi_set_master_pattern(c, upper, lower, dtype, npx, npy); });
}

int kpdi_group_set_detector(kpdi_group *g, const double *gnomonic_bounds, double pcz, int nrows, int ncols,
                            const double *om_detector_to_sample) {
  if (!g) return bad_group();
  return run_all(g, [=](int, kpdi_ctx *c) { return kpdi_set_detector(c, gnomonic_bounds, pcz, nrows, ncols, om_detector_to_sample); });
}

// the dictionary is SIMULATED where it is matched: a member projects the rotations of its chunks in its own HBM.  The
// rotations (32 bytes per pattern) are copied into the queued job: nothing of the caller's is read after return.
int kpdi_group_push_rotations_chunk(kpdi_group *g, const double *rotations, int64_t n, int64_t global_start, int rescale,
                                    double out_min, double out_max) {
  if (!g) return bad_group();
  const int rc = check_chunk(rotations, n);
  if (rc) return rc;
  auto rot = std::make_shared<std::vector<double>>(rotations, rotations + (size_t)n * 4);
  return queue_chunk(g, n, [=](kpdi_ctx *c, const kpdi::ChunkPiece &pc) {
    return kpdi_push_rotations_chunk(c, rot->data() + (size_t)pc.row0 * 4, pc.rows, global_start + pc.row0, rescale, out_min, out_max);
  }, nullptr);
}

int kpdi_group_hold_dictionary_chunk(kpdi_group *g, const void *patterns, int dtype, int64_t n_chunk,
                                     int64_t global_start) {
  if (!g) return bad_group();
  size_t rb = 0;
  int rc = chunk_row_bytes(g, dtype, &rb);
  if (rc) return rc;
  rc = check_chunk(patterns, n_chunk);
  if (rc) return rc;
  return hold_chunk(g, n_chunk, [=](kpdi_ctx *c, const kpdi::ChunkPiece &pc) {
    return kpdi_hold_dictionary_chunk(c, (const char *)patterns + (size_t)pc.row0 * rb, dtype, pc.rows, global_start + pc.row0);
  });
}

int kpdi_group_hold_rotations_chunk(kpdi_group *g, const double *rotations, int64_t n, int64_t global_start, int rescale,
                                    double out_min, double out_max) {
  if (!g) return bad_group();
  const int rc = check_chunk(rotations, n);
  if (rc) return rc;
  return hold_chunk(g, n, [=](kpdi_ctx *c, const kpdi::ChunkPiece &pc) {
    return kpdi_hold_rotations_chunk(c, rotations + (size_t)pc.row0 * 4, pc.rows, global_start + pc.row0, rescale, out_min, out_max);
  });
}

int kpdi_group_sweep_held(kpdi_group *g) {
  if (!g) return bad_group();
  return run_all(g, [](int, kpdi_ctx *c) {
    int64_t n = 0;
    const int rc = kpdi_held_size(c, &n, nullptr);
    return rc ? rc : (n > 0 ? kpdi_sweep_held(c) : (int)KPDI_OK);  // (a member whose share of every chunk was empty)
  });
}

int kpdi_group_release_held(kpdi_group *g) {
  if (!g) return bad_group();
  g->held.assign(g->n, 0);
  return run_all(g, [](int, kpdi_ctx *c) { return kpdi_release_held(c); });
}

int kpdi_group_held_size(kpdi_group *g, int64_t *n_patterns, int64_t *n_bytes) {
  if (!g) return bad_group();
  std::vector<int64_t> np(g->n, 0), nb(g->n, 0);
  const int rc = run_all(g, [&](int i, kpdi_ctx *c) { return kpdi_held_size(c, &np[i], &nb[i]); });
  if (rc) return rc;
  int64_t p = 0, b = 0;
  for (int i = 0; i < g->n; ++i) p += np[i], b += nb[i];
  if (n_patterns) *n_patterns = p;
  if (n_bytes) *n_bytes = b;
  return KPDI_OK;
}

int kpdi_group_reset_topk(kpdi_group *g) {
  if (!g) return bad_group();
  reset_loads(g);
  return run_all(g, [](int, kpdi_ctx *c) { return kpdi_reset_topk(c); });
}

int kpdi_group_finalize(kpdi_group *g, float *scores_out, int64_t *indices_out) {
  if (!g) return bad_group();
  if (!scores_out || !indices_out) return kpdi::fail_msg(KPDI_EINVAL, "output pointer is NULL");
  return finalize_all(g, kpdi::FINALIZE_F32, [=](kpdi_ctx *c) { return kpdi_finalize(c, scores_out, indices_out); });
}

int kpdi_group_finalize_f64(kpdi_group *g, double *scores_out, int64_t *indices_out) {
  if (!g) return bad_group();
  if (!scores_out || !indices_out) return kpdi::fail_msg(KPDI_EINVAL, "output pointer is NULL");
  return finalize_all(g, kpdi::FINALIZE_F64, [=](kpdi_ctx *c) { return kpdi_finalize_f64(c, scores_out, indices_out); });
}

// the merged result on its way to the host while the NEXT map is queued on every member (kpdi_finalize_async): nothing
// here waits for a device - the members record events, the root queues copies / the collective, merge and hand-over
int kpdi_group_finalize_async(kpdi_group *g, int *ticket) {
  if (!g) return bad_group();
  if (!ticket) return kpdi::fail_msg(KPDI_EINVAL, "ticket is NULL");
  if (g->exact64) return kpdi::fail_msg(KPDI_EINVAL, "kpdi_group_finalize_async: not available in float64 arithmetic (use kpdi_group_finalize_f64)");
  return finalize_all(g, kpdi::FINALIZE_ASYNC, [=](kpdi_ctx *c) { return kpdi_finalize_async(c, ticket); });
}

int kpdi_group_finalize_wait(kpdi_group *g, int ticket, float *scores_out, int64_t *indices_out) {
  if (!g) return bad_group();
  const int rc = join_workers(g);  // (member 0's worker may be queuing the next map's chunks)
  return rc ? rc : kpdi_finalize_wait(g->ctx[0], ticket, scores_out, indices_out);
}

int kpdi_group_pending_result_size(kpdi_group *g, int ticket, int64_t *n) {
  if (!g) return bad_group();
  return kpdi_pending_result_size(g->ctx[0], ticket, n);
}

int kpdi_group_set_profiling(kpdi_group *g, int on) {
  if (!g) return bad_group();
  return run_all(g, [=](int, kpdi_ctx *c) { return kpdi_set_profiling(c, on); });
}

int kpdi_group_reset_counters(kpdi_group *g) {
  if (!g) return bad_group();
  return run_all(g, [](int, kpdi_ctx *c) { return kpdi_reset_counters(c); });
}

}  // extern "C"
