// group.hip - kpdi_group: several MI355X driven from ONE thread of ONE process (include/kpdi.h, "a group of contexts").
//
// The reference's dictionary_indexing() is one call in one interpreter (signals/ebsd.py:1827-1984; the chunk loop of
// indexing/_dictionary_indexing.py:100-128 never leaves the process).  A group keeps that call shape on a node with
// several GPUs: one kpdi_ctx per device, one host thread per member (member 0 runs on the caller's thread), every entry
// point the per-context one fanned out, every dictionary chunk block-assigned to the members, and ONE merged result.
//
// Host code only: everything that touches a device goes through the per-context C ABI (api.hip) or the few hooks of
// group_hooks.h (in-process RCCL communicator, peer-copy gather).
#include "../../include/kpdi.h"
#include "group_hooks.h"

#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

// one host thread per member beyond the first: kernel launches, uploads and RCCL calls of the members run concurrently
// (a push is ~10 launches = ~0.1 ms of host time; one thread driving 8 devices in turn would serialise 0.8 ms per chunk)
struct Worker {
  std::thread th;
  std::mutex mu;
  std::condition_variable cv;
  std::function<int()> job;
  bool pending = false, stop = false;
  int rc = 0;
  std::string err;

  void loop() {
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      cv.wait(lk, [&] { return pending || stop; });
      if (stop) return;
      std::function<int()> j = std::move(job);
      lk.unlock();
      const int r = j();
      std::string e = r ? kpdi::thread_error() : "";  // (thread-local: fetched on THIS thread)
      lk.lock();
      rc = r;
      err = std::move(e);
      pending = false;
      cv.notify_all();
    }
  }
  void post(std::function<int()> j) {
    {
      std::lock_guard<std::mutex> lk(mu);
      job = std::move(j);
      pending = true;
    }
    cv.notify_all();
  }
  int wait(std::string *e) {
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return !pending; });
    *e = err;
    return rc;
  }
};

}  // namespace

struct kpdi_group {
  int n = 0;
  std::vector<int> dev;
  std::vector<kpdi_ctx *> ctx;
  std::vector<std::unique_ptr<Worker>> workers;  // [0] unused: member 0 runs on the calling thread
  int gather = KPDI_GATHER_NONE;
  std::string describe;
  bool exact64 = false;       // the problem is KPDI_COMPUTE_F64 (no async hand-over)
  int64_t npix = 0;           // detector pixels of the problem: a chunk's rows are npix elements apart
  std::vector<kpdi::ListsView> views;
};

namespace {

// fn(i, ctx_i) on every member concurrently; the first failure (lowest member) is reported with the member named
template <typename F>
int run_all(kpdi_group *g, F fn) {
  for (int i = 1; i < g->n; ++i) g->workers[i]->post([&fn, g, i] { return fn(i, g->ctx[i]); });
  int rc = fn(0, g->ctx[0]);
  std::string err = rc ? kpdi::thread_error() : "";
  int who = 0;
  for (int i = 1; i < g->n; ++i) {
    std::string e;
    const int r = g->workers[i]->wait(&e);
    if (r && !rc) {
      rc = r;
      err = e;
      who = i;
    }
  }
  if (rc && g->n > 1)
    return kpdi::fail_msg(rc, "device %d (group member %d of %d): %s", g->dev[who], who, g->n, err.c_str());
  return rc;
}

int bad_group() { return kpdi::fail_msg(KPDI_EINVAL, "group is NULL"); }

void share(int64_t n, int i, int n_dev, int64_t *start, int64_t *end) {
  const int64_t base = n / n_dev, rem = n % n_dev;
  *start = i * base + (i < rem ? i : rem);
  *end = *start + base + (i < rem ? 1 : 0);
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

// finalize: `root(ctx0)` hands the merged result over; with an RCCL communicator the other members join its all-gather
template <typename F>
int finalize_all(kpdi_group *g, F root) {
  int rc = gather_to_root(g);
  if (rc) return rc;
  if (g->gather == KPDI_GATHER_RCCL && g->n > 1)
    return run_all(g, [&root](int i, kpdi_ctx *c) { return i == 0 ? root(c) : kpdi::finalize_participate(c); });
  return root(g->ctx[0]);
}

}  // namespace

extern "C" {

int kpdi_group_chunk_share(int64_t n_chunk, int i, int n_dev, int64_t *start, int64_t *end) {
  if (!start || !end || n_dev < 1 || i < 0 || i >= n_dev || n_chunk < 0)
    return kpdi::fail_msg(KPDI_EINVAL, "kpdi_group_chunk_share: bad arguments");
  share(n_chunk, i, n_dev, start, end);
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
      if (kpdi::comm_init_all(g->ctx.data(), n_dev) == KPDI_OK) mode = KPDI_GATHER_RCCL;
      else mode = KPDI_GATHER_P2P, note = std::string(" (RCCL unavailable: ") + kpdi::thread_error() + ")";
    }
  } else if (mode == KPDI_GATHER_RCCL) {
    const int rc = kpdi::comm_init_all(g->ctx.data(), n_dev);  // (also with one device: keeps the path testable)
    if (rc) {
      const std::string e = kpdi::thread_error();
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
  for (int i = 1; i < n_dev; ++i) {
    g->workers[i].reset(new Worker());
    Worker *w = g->workers[i].get();
    w->th = std::thread([w] { w->loop(); });
  }
  *out = g.release();
  return KPDI_OK;
}

int kpdi_group_destroy(kpdi_group *g) {
  if (!g) return KPDI_OK;
  for (int i = 1; i < g->n; ++i) {
    Worker *w = g->workers[i].get();
    {
      std::lock_guard<std::mutex> lk(w->mu);
      w->stop = true;
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
  }
  return rc;
}

int kpdi_group_set_keep_n(kpdi_group *g, int keep_n) {
  if (!g) return bad_group();
  return run_all(g, [=](int, kpdi_ctx *c) { return kpdi_set_keep_n(c, keep_n); });
}

int kpdi_group_set_experimental(kpdi_group *g, const void *patterns, int dtype, int64_t m_all, const uint8_t *nav_mask) {
  if (!g) return bad_group();
  // ONE host copy, n uploads in parallel (every device has its own link to the host); the members synchronise their
  // upload before they return, so the caller's buffer is free again
  return run_all(g, [=](int, kpdi_ctx *c) {
    const int rc = kpdi_set_experimental(c, patterns, dtype, m_all, nav_mask);
    return rc ? rc : kpdi_synchronize(c);
  });
}

int kpdi_group_set_experimental_dev(kpdi_group *g, const void *const *d_patterns, int dtype, int64_t m_all,
                                    const uint8_t *nav_mask) {
  if (!g) return bad_group();
  if (!d_patterns) return kpdi::fail_msg(KPDI_EINVAL, "d_patterns is NULL");
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
  return kpdi_get_experimental(g->ctx[0], patterns_out);
}

}  // extern "C"

namespace {

// member i's part of a chunk of `n_chunk` rows of `row_bytes` each: fn(ctx, first row pointer, rows, global start)
template <typename F>
int split_rows(kpdi_group *g, const void *rows, size_t row_bytes, int64_t n_chunk, int64_t global_start, F fn) {
  if (!rows) return kpdi::fail_msg(KPDI_EINVAL, "pointer is NULL");
  if (n_chunk <= 0) return kpdi::fail_msg(KPDI_EINVAL, "dictionary chunk must hold at least one pattern");
  const int n = g->n;
  return run_all(g, [=](int i, kpdi_ctx *c) {
    int64_t a, b;
    share(n_chunk, i, n, &a, &b);
    if (a >= b) return (int)KPDI_OK;  // fewer rows than members
    return fn(c, (const void *)((const char *)rows + (size_t)a * row_bytes), b - a, global_start + a);
  });
}

int chunk_row_bytes(kpdi_group *g, int dtype, size_t *row_bytes) {
  const size_t es = kpdi_dtype_size(dtype);
  if (es == 0) return kpdi::fail_msg(KPDI_EINVAL, "unknown dtype %d", dtype);
  if (g->npix <= 0) return kpdi::fail_msg(KPDI_EINVAL, "kpdi_group_set_problem has not been called");
  *row_bytes = (size_t)g->npix * es;
  return KPDI_OK;
}

}  // namespace

extern "C" {

int kpdi_group_push_dictionary_chunk(kpdi_group *g, const void *patterns, int dtype, int64_t n_chunk,
                                     int64_t global_start) {
  if (!g) return bad_group();
  size_t rb = 0;
  int rc = chunk_row_bytes(g, dtype, &rb);
  if (rc) return rc;
  // every member uploads its rows through its own staging buffers and copy stream (its own link to the host) and
  // returns when the upload has consumed them; the sweeps run on
  return split_rows(g, patterns, rb, n_chunk, global_start, [=](kpdi_ctx *c, const void *p, int64_t n, int64_t start) {
    return kpdi_push_dictionary_chunk(c, p, dtype, n, start);
  });
}

int kpdi_group_push_dictionary_chunk_dev(kpdi_group *g, const void *const *d_patterns, int dtype, const int64_t *n_chunk,
                                         const int64_t *global_start) {
  if (!g) return bad_group();
  if (!d_patterns || !n_chunk || !global_start) return kpdi::fail_msg(KPDI_EINVAL, "NULL argument");
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

// the dictionary is SIMULATED where it is matched: member i projects its share of the rotations in its own HBM
int kpdi_group_push_rotations_chunk(kpdi_group *g, const double *rotations, int64_t n, int64_t global_start, int rescale,
                                    double out_min, double out_max) {
  if (!g) return bad_group();
  return split_rows(g, rotations, 4 * sizeof(double), n, global_start, [=](kpdi_ctx *c, const void *p, int64_t cnt, int64_t start) {
    return kpdi_push_rotations_chunk(c, (const double *)p, cnt, start, rescale, out_min, out_max);
  });
}

int kpdi_group_hold_dictionary_chunk(kpdi_group *g, const void *patterns, int dtype, int64_t n_chunk,
                                     int64_t global_start) {
  if (!g) return bad_group();
  size_t rb = 0;
  int rc = chunk_row_bytes(g, dtype, &rb);
  if (rc) return rc;
  return split_rows(g, patterns, rb, n_chunk, global_start, [=](kpdi_ctx *c, const void *p, int64_t n, int64_t start) {
    return kpdi_hold_dictionary_chunk(c, p, dtype, n, start);
  });
}

int kpdi_group_hold_rotations_chunk(kpdi_group *g, const double *rotations, int64_t n, int64_t global_start, int rescale,
                                    double out_min, double out_max) {
  if (!g) return bad_group();
  return split_rows(g, rotations, 4 * sizeof(double), n, global_start, [=](kpdi_ctx *c, const void *p, int64_t cnt, int64_t start) {
    return kpdi_hold_rotations_chunk(c, (const double *)p, cnt, start, rescale, out_min, out_max);
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
  return run_all(g, [](int, kpdi_ctx *c) { return kpdi_release_held(c); });
}

int kpdi_group_held_size(kpdi_group *g, int64_t *n_patterns, int64_t *n_bytes) {
  if (!g) return bad_group();
  int64_t np = 0, nb = 0;
  for (kpdi_ctx *c : g->ctx) {
    int64_t a = 0, b = 0;
    const int rc = kpdi_held_size(c, &a, &b);
    if (rc) return rc;
    np += a;
    nb += b;
  }
  if (n_patterns) *n_patterns = np;
  if (n_bytes) *n_bytes = nb;
  return KPDI_OK;
}

int kpdi_group_reset_topk(kpdi_group *g) {
  if (!g) return bad_group();
  return run_all(g, [](int, kpdi_ctx *c) { return kpdi_reset_topk(c); });
}

int kpdi_group_finalize(kpdi_group *g, float *scores_out, int64_t *indices_out) {
  if (!g) return bad_group();
  return finalize_all(g, [=](kpdi_ctx *c) { return kpdi_finalize(c, scores_out, indices_out); });
}

int kpdi_group_finalize_f64(kpdi_group *g, double *scores_out, int64_t *indices_out) {
  if (!g) return bad_group();
  return finalize_all(g, [=](kpdi_ctx *c) { return kpdi_finalize_f64(c, scores_out, indices_out); });
}

// the merged result on its way to the host while the NEXT map is queued on every member (kpdi_finalize_async): nothing
// here waits for a device - the members record events, the root queues copies / the collective, merge and hand-over
int kpdi_group_finalize_async(kpdi_group *g, int *ticket) {
  if (!g) return bad_group();
  if (!ticket) return kpdi::fail_msg(KPDI_EINVAL, "ticket is NULL");
  if (g->exact64) return kpdi::fail_msg(KPDI_EINVAL, "kpdi_group_finalize_async: not available in float64 arithmetic (use kpdi_group_finalize_f64)");
  return finalize_all(g, [=](kpdi_ctx *c) { return kpdi_finalize_async(c, ticket); });
}

int kpdi_group_finalize_wait(kpdi_group *g, int ticket, float *scores_out, int64_t *indices_out) {
  if (!g) return bad_group();
  return kpdi_finalize_wait(g->ctx[0], ticket, scores_out, indices_out);
}

int kpdi_group_pending_result_size(kpdi_group *g, int ticket, int64_t *n) {
  if (!g) return bad_group();
  return kpdi_pending_result_size(g->ctx[0], ticket, n);
}

int kpdi_group_set_profiling(kpdi_group *g, int on) {
  if (!g) return bad_group();
  for (kpdi_ctx *c : g->ctx) {
    const int rc = kpdi_set_profiling(c, on);
    if (rc) return rc;
  }
  return KPDI_OK;
}

int kpdi_group_reset_counters(kpdi_group *g) {
  if (!g) return bad_group();
  return run_all(g, [](int, kpdi_ctx *c) { return kpdi_reset_counters(c); });
}

}  // extern "C"
