// finalize.hip - the hand-over of a sweep's result: this rank's lists, the gather over ranks / group members (RCCL
// all-gather, peer copies) and its merge, the page-locked result slots of kpdi_finalize[_async / _wait]; the RCCL
// loader and communicators; the hooks kpdi_group (group.hip) is built on.
// (one of the host translation units api.hip was split into in round 5: context.h holds what they share)
#include "context.h"

#include <chrono>
#include <thread>

using namespace kpdi;

namespace kpdi {
Rccl g_rccl;
}

// kpdi_finalize in two halves.  finalize_enqueue: (all-gather + merge over the ranks,) the result's device-to-host copies
// into page-locked slot `slot`, an event behind them - nothing waits.  finalize_collect: wait for that event, hand the slot's
// contents to the caller.  kpdi_finalize = both; kpdi_finalize_async / kpdi_finalize_wait let a caller that indexes map
// after map queue the NEXT map's kernels before it collects this one's result (the hand-over - synchronisation, copies,
// widening the indices - is ~0.1 ms of host time per call during which the GPU otherwise idles: 3 % of one rank's 3 ms
// share of configs[1] at N = 8).
namespace kpdi {
// this rank's running lists, made presentable: a rank that pushed nothing contributes empty lists
int own_lists(kpdi_ctx *c) {
  int rc = flush_pending(c);  // (small chunks that were waiting for company: swept now)
  if (rc) return rc;
  rc = ensure_running(c);
  if (rc) return rc;
  if (c->run_empty && !c->exact64) {
    const size_t n0 = (size_t)c->m * c->keep_n;
    rc = wait_result_copy(c);
    if (!rc) rc = queue_fill_topk(c, c->run_s[c->run_cur].as<float>(), c->run_i[c->run_cur].as<int>(), n0);
    if (!rc) rc = flush_fills(c);
    if (rc) return rc;
    c->run_empty = false;
  }
  return KPDI_OK;
}

// the FINAL lists of the sweep on this rank: its own, or - with a communicator / in an in-process group - the merge of
// every rank's (RCCL all-gather, or lists that kpdi::root_gather_p2p has already peer-copied into the gather buffers)
int final_lists(kpdi_ctx *c, const float **out_s, const int **out_i) {
  int rc = own_lists(c);
  if (rc) return rc;
  const int k = c->keep_n;
  const size_t n = (size_t)c->m * k;
  const float *d_s = c->run_s[c->run_cur].as<float>();
  const int *d_i = c->run_i[c->run_cur].as<int>();
  const int ranks = c->p2p_ranks ? c->p2p_ranks : (c->comm ? c->nranks : 0);
  const bool peer_copied = c->p2p_ranks > 0;
  c->p2p_ranks = 0;
  c->cnt.gather_ranks = ranks;
  if (ranks) {  // also with one rank: keeps the RCCL path testable on a single GPU
    rc = wait_result_copy(c);  // (the merge below writes the other half of the ping-pong pair)
    if (rc) return rc;
    if (!peer_copied) {
      HIPCHK(c->gather_s.reserve(n * ranks * sizeof(float)));
      HIPCHK(c->gather_i.reserve(n * ranks * sizeof(int)));
      ncclResult_t r;
      {
        ScopedTimer t(c, &c->ev_comm);
        r = g_rccl.GroupStart();
        if (r == ncclSuccess) r = g_rccl.AllGather(d_s, c->gather_s.p, n, ncclFloat32, c->comm, c->stream);
        if (r == ncclSuccess) r = g_rccl.AllGather(d_i, c->gather_i.p, n, ncclInt32, c->comm, c->stream);
        ncclResult_t r2 = g_rccl.GroupEnd();
        if (r == ncclSuccess) r = r2;
      }
      if (r != ncclSuccess) return fail(KPDI_ECOMM, "RCCL all-gather failed: %s", g_rccl.GetErrorString(r));
    }
    const int nxt = c->run_cur ^ 1;
    kpdi::MergeLaunch mg{};
    mg.m = c->m;
    mg.k = k;
    mg.n_src = 1;
    mg.src_scores[0] = c->gather_s.as<float>();
    mg.src_idx[0] = c->gather_i.as<int>();
    mg.src_lists[0] = ranks;
    mg.src_len[0] = k;
    mg.src_row_stride[0] = k;
    mg.src_list_stride[0] = (int)n;
    mg.out_scores = c->run_s[nxt].as<float>();
    mg.out_idx = c->run_i[nxt].as<int>();
    mg.out_stride = k;
    mg.out_offset = 0;
    {
      ScopedTimer t(c, &c->ev_merge);
      HIPCHK(kpdi::launch_merge(mg, c->stream));
    }
    d_s = c->run_s[nxt].as<float>();
    d_i = c->run_i[nxt].as<int>();
    // the per-rank running list (run_cur) is left untouched: finalize is idempotent
  }
  c->final_idx = d_i;
  c->final_valid = true;
  *out_s = d_s;
  *out_i = d_i;
  return KPDI_OK;
}
}  // namespace kpdi

namespace {
int finalize_enqueue(kpdi_ctx *c, int slot, bool own_stream) {
  const float *d_s = nullptr;
  const int *d_i = nullptr;
  int rc = final_lists(c, &d_s, &d_i);
  if (rc) return rc;
  const size_t n = (size_t)c->m * c->keep_n;
  kpdi_ctx::ResultSlot &rs = c->slots[slot];
  HIPCHK(rs.pin.reserve(n * (sizeof(float) + sizeof(int))));
  if (!rs.ready) HIPCHK(hipEventCreateWithFlags(&rs.ready, hipEventDisableTiming));
  rs.n = n;
  rs.pending = true;
  c->result_i32 = nullptr;
  float *h_s = (float *)rs.pin.p;
  if (!own_stream) {  // kpdi_finalize waits right away: the hop to another stream would only add latency (+15 us measured)
    HIPCHK(hipMemcpyAsync(h_s, d_s, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(h_s + n, d_i, n * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipEventRecord(rs.ready, c->stream));
    return KPDI_OK;
  }
  if (!c->result_stream) {
    HIPCHK(hipStreamCreateWithFlags(&c->result_stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&c->result_done, hipEventDisableTiming));
  }
  HIPCHK(hipEventRecord(c->result_done, c->stream));
  HIPCHK(hipStreamWaitEvent(c->result_stream, c->result_done, 0));
  HIPCHK(hipMemcpyAsync(h_s, d_s, n * sizeof(float), hipMemcpyDeviceToHost, c->result_stream));
  HIPCHK(hipMemcpyAsync(h_s + n, d_i, n * sizeof(int), hipMemcpyDeviceToHost, c->result_stream));
  HIPCHK(hipEventRecord(rs.ready, c->result_stream));
  c->result_copy = rs.ready;
  return KPDI_OK;
}

int finalize_collect(kpdi_ctx *c, int slot, float *scores_out, int64_t *indices_out) {
  kpdi_ctx::ResultSlot &rs = c->slots[slot];
  if (!rs.pending) return fail(KPDI_EINVAL, "no result is pending in slot %d", slot);
  HIPCHK(hipEventSynchronize(rs.ready));
  rs.pending = false;
  const size_t n = rs.n;
  const float *h_s = (const float *)rs.pin.p;
  const int *h_i = (const int *)(h_s + n);
  memcpy(scores_out, h_s, n * sizeof(float));
  for (size_t i = 0; i < n; ++i) indices_out[i] = (int64_t)h_i[i];
  c->result_i32 = (const int32_t *)h_i;  // (valid until this slot is used again: two finalize calls on)
  c->result_n = (int64_t)n;
  return KPDI_OK;
}

int finalize_args(kpdi_ctx *c) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  if (!c->have_exp || !c->have_problem) return fail(KPDI_EINVAL, "nothing to finalise");
  return use_device(c);
}
}  // namespace

extern "C" {

int kpdi_finalize(kpdi_ctx *c, float *scores_out, int64_t *indices_out) {
  int rc = finalize_args(c);
  if (rc) return rc;
  if (!scores_out || !indices_out) return fail(KPDI_EINVAL, "output pointer is NULL");
  if (c->m == 0) return KPDI_OK;
  if (c->exact64) {
    rc = ensure_running(c);
    if (rc) return rc;
    return finalize64(c, nullptr, scores_out, indices_out);
  }
  // the slot of an outstanding kpdi_finalize_async ticket is never touched (its copy may still be in flight and its
  // ticket must stay collectable): take the other one, or fail like kpdi_finalize_async does
  int slot = c->next_slot;
  if (c->slots[slot].pending) slot ^= 1;
  if (c->slots[slot].pending)
    return fail(KPDI_EINVAL, "two results are already pending: collect one with kpdi_finalize_wait first");
  c->next_slot = slot ^ 1;
  rc = finalize_enqueue(c, slot, false);
  if (rc) return rc;
  return finalize_collect(c, slot, scores_out, indices_out);
}

int kpdi_finalize_async(kpdi_ctx *c, int *ticket) {
  int rc = finalize_args(c);
  if (rc) return rc;
  if (!ticket) return fail(KPDI_EINVAL, "ticket is NULL");
  if (c->exact64) return fail(KPDI_EINVAL, "kpdi_finalize_async: not available in float64 arithmetic (use kpdi_finalize_f64)");
  if (c->m == 0) return fail(KPDI_EINVAL, "no experimental patterns to finalise");
  int slot = c->next_slot;
  if (c->slots[slot].pending) slot ^= 1;
  if (c->slots[slot].pending)
    return fail(KPDI_EINVAL, "two results are already pending: collect one with kpdi_finalize_wait first");
  c->next_slot = slot ^ 1;
  rc = finalize_enqueue(c, slot, true);
  if (rc) return rc;
  *ticket = slot;
  return KPDI_OK;
}

int kpdi_finalize_wait(kpdi_ctx *c, int ticket, float *scores_out, int64_t *indices_out) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  if (!scores_out || !indices_out) return fail(KPDI_EINVAL, "output pointer is NULL");
  if (ticket < 0 || ticket > 1) return fail(KPDI_EINVAL, "bad ticket %d", ticket);
  int rc = use_device(c);
  if (rc) return rc;
  return finalize_collect(c, ticket, scores_out, indices_out);
}

int kpdi_pending_result_size(kpdi_ctx *c, int ticket, int64_t *n) {
  if (!c || !n) return fail(KPDI_EINVAL, "NULL argument");
  if (ticket < 0 || ticket > 1 || !c->slots[ticket].pending) return fail(KPDI_EINVAL, "no result is pending for ticket %d", ticket);
  *n = (int64_t)c->slots[ticket].n;
  return KPDI_OK;
}

int kpdi_result_indices_i32(kpdi_ctx *c, const int32_t **indices, int64_t *n) {
  if (!c || !indices || !n) return fail(KPDI_EINVAL, "NULL argument");
  *indices = c->result_i32;
  *n = c->result_i32 ? c->result_n : 0;
  return KPDI_OK;
}

int kpdi_comm_unique_id(uint8_t *id_out) {
  if (!id_out) return fail(KPDI_EINVAL, "id_out is NULL");
  if (!g_rccl.load()) return fail(KPDI_ECOMM, "cannot load librccl: %s", g_rccl.why.c_str());
  static_assert(sizeof(ncclUniqueId) == KPDI_UNIQUE_ID_BYTES, "ncclUniqueId size");
  ncclUniqueId id;
  ncclResult_t r = g_rccl.GetUniqueId(&id);
  if (r != ncclSuccess) return fail(KPDI_ECOMM, "ncclGetUniqueId: %s", g_rccl.GetErrorString(r));
  memcpy(id_out, &id, sizeof id);
  return KPDI_OK;
}

int kpdi_comm_init(kpdi_ctx *c, int rank, int nranks, const uint8_t *id) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail(KPDI_EINVAL, "bad rank %d / %d", rank, nranks);
  if (!id) return fail(KPDI_EINVAL, "id is NULL");
  int rc = use_device(c);
  if (rc) return rc;
  if (!g_rccl.load()) return fail(KPDI_ECOMM, "cannot load librccl: %s", g_rccl.why.c_str());
  c->comm_dropped = false;
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof uid);
  ncclComm_t comm = nullptr;
  ncclResult_t r = g_rccl.CommInitRank(&comm, nranks, uid, rank);
  if (r != ncclSuccess)
    return fail(KPDI_ECOMM, "ncclCommInitRank(rank %d of %d, device %d): %s", rank, nranks, c->device, g_rccl.GetErrorString(r));
  if (c->comm_dropped) {  // the caller gave up on this call long ago (kpdi_comm_drop from another thread)
    if (g_rccl.CommAbort) g_rccl.CommAbort(comm);
    return fail(KPDI_ECOMM, "the communicator was dropped while ncclCommInitRank was still running");
  }
  c->comm = comm;
  c->rank = rank;
  c->nranks = nranks;
  return KPDI_OK;
}

// One all-gather of `n_bytes` per rank on the context's stream, awaited for at most `timeout_ms`: a communicator whose
// bootstrap succeeded can still hang in its first collective (shared-memory segments, IPC handles, peer access) - better
// found here, where the caller can still fall back (kpdi_comm_drop + kpdi_export_lists / kpdi_import_lists), than in
// the finalize of a sweep.
int kpdi_comm_selftest(kpdi_ctx *c, int64_t n_bytes, int timeout_ms) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  if (!c->comm) return fail(KPDI_EINVAL, "no communicator attached (kpdi_comm_init)");
  if (n_bytes < 4 || n_bytes > (1ll << 30) || timeout_ms < 1) return fail(KPDI_EINVAL, "kpdi_comm_selftest: bad size / timeout");
  int rc = use_device(c);
  if (rc) return rc;
  const size_t n = (size_t)n_bytes / 4;
  // buffers and event of the test: freed on EVERY way out - except after a time-out, where the collective may still touch
  // the buffers: they pass to the context and go when kpdi_comm_drop has aborted the communicator
  struct Scratch {
    kpdi_ctx *c;
    unsigned *send = nullptr, *recv = nullptr;
    hipEvent_t done = nullptr;
    bool keep = false;
    ~Scratch() {
      if (done) (void)hipEventDestroy(done);
      if (keep) {
        c->selftest_left[0] = send;
        c->selftest_left[1] = recv;
      } else {
        if (send) (void)hipFree(send);
        if (recv) (void)hipFree(recv);
      }
    }
  } sc{c};
  HIPCHK(hipMalloc(&sc.send, n * 4));
  HIPCHK(hipMalloc(&sc.recv, n * 4 * c->nranks));
  std::vector<unsigned> h(n, 0x5eed0000u + (unsigned)c->rank);
  HIPCHK(hipMemcpyAsync(sc.send, h.data(), n * 4, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  ncclResult_t r = g_rccl.AllGather(sc.send, sc.recv, n, ncclUint32, c->comm, c->stream);
  if (r != ncclSuccess) return fail(KPDI_ECOMM, "RCCL all-gather (self-test): %s", g_rccl.GetErrorString(r));
  HIPCHK(hipEventCreateWithFlags(&sc.done, hipEventDisableTiming));
  HIPCHK(hipEventRecord(sc.done, c->stream));
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t q = hipEventQuery(sc.done);
    if (q == hipSuccess) break;
    if (q != hipErrorNotReady) return fail(KPDI_EHIP, "self-test all-gather: %s", hipGetErrorString(q));
    if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(timeout_ms)) {
      sc.keep = true;
      return fail(KPDI_ETIMEOUT, "the first RCCL all-gather (%lld bytes per rank, %d ranks) did not complete within %d ms", (long long)n_bytes,
                  c->nranks, timeout_ms);
    }
    std::this_thread::sleep_for(std::chrono::microseconds(200));
  }
  std::vector<unsigned> got(n * c->nranks);
  HIPCHK(hipMemcpy(got.data(), sc.recv, got.size() * 4, hipMemcpyDeviceToHost));
  for (int j = 0; j < c->nranks; ++j)
    if (got[(size_t)j * n] != 0x5eed0000u + (unsigned)j || got[(size_t)j * n + n - 1] != 0x5eed0000u + (unsigned)j)
      return fail(KPDI_ECOMM, "self-test all-gather: the block of rank %d arrived damaged", j);
  return KPDI_OK;
}

// Forget the communicator (aborting whatever it still has in flight): finalize then hands out this context's own lists,
// or the lists given to kpdi_import_lists.  Safe to call while a kpdi_comm_init of this context hangs on another thread.
int kpdi_comm_drop(kpdi_ctx *c) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  c->comm_dropped = true;
  ncclComm_t comm = c->comm;
  c->comm = nullptr;
  c->rank = 0;
  c->nranks = 1;
  if (comm) {
    if (g_rccl.CommAbort) g_rccl.CommAbort(comm);
    else if (g_rccl.CommDestroy) g_rccl.CommDestroy(comm);
  }
  for (void *&p : c->selftest_left) {  // (what a timed-out self-test left behind: nothing touches it any more)
    if (p && use_device(c) == KPDI_OK) (void)hipFree(p);
    p = nullptr;
  }
  return KPDI_OK;
}

// This context's OWN running best-k lists (no gather): m x keep_n scores - float, or double in KPDI_COMPUTE_F64 - and
// their int32 dictionary indices (INT32_MAX = unfilled), in ranking order.
int kpdi_export_lists(kpdi_ctx *c, void *scores_out, int32_t *indices_out) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  if (!c->have_exp || !c->have_problem) return fail(KPDI_EINVAL, "nothing to export");
  if (!scores_out || !indices_out) return fail(KPDI_EINVAL, "output pointer is NULL");
  int rc = use_device(c);
  if (rc) return rc;
  if (c->m == 0) return KPDI_OK;
  rc = own_lists(c);
  if (rc) return rc;
  const size_t n = (size_t)c->m * c->keep_n;
  const void *d_s = c->exact64 ? c->run64_s.p : c->run_s[c->run_cur].p;
  const int *d_i = c->exact64 ? c->run64_i.as<int>() : c->run_i[c->run_cur].as<int>();
  HIPCHK(hipMemcpyAsync(scores_out, d_s, n * (c->exact64 ? sizeof(double) : sizeof(float)), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(indices_out, d_i, n * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return KPDI_OK;
}

// The lists of `n_ranks` contexts (rank-major: [n_ranks][m * keep_n], as kpdi_export_lists wrote them) take the place of
// an all-gather: the NEXT finalize of this context merges them - same kernel, same total order - instead of its own.
// The host-staged gather of a multi-process job whose RCCL communicator cannot be used (kikuchipy_amd.parallel).
int kpdi_import_lists(kpdi_ctx *c, const void *scores_all, const int32_t *indices_all, int n_ranks) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  if (!c->have_exp || !c->have_problem) return fail(KPDI_EINVAL, "nothing to merge into");
  if (!scores_all || !indices_all || n_ranks < 1) return fail(KPDI_EINVAL, "kpdi_import_lists: bad arguments");
  int rc = use_device(c);
  if (rc) return rc;
  if (c->m == 0) return KPDI_OK;
  const size_t n = (size_t)c->m * c->keep_n * n_ranks;
  const size_t es = c->exact64 ? sizeof(double) : sizeof(float);
  DevBuf &gs = c->exact64 ? c->gather64_s : c->gather_s;
  DevBuf &gi = c->exact64 ? c->gather64_i : c->gather_i;
  HIPCHK(hipStreamSynchronize(c->stream));  // (a merge of the previous finalize may still read the gather buffers)
  HIPCHK(gs.reserve(n * es));
  HIPCHK(gi.reserve(n * sizeof(int)));
  HIPCHK(hipMemcpyAsync(gs.p, scores_all, n * es, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(gi.p, indices_all, n * sizeof(int), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));  // the caller's buffers are free again
  c->cnt.h2d_bytes += (double)(n * (es + sizeof(int)));
  c->p2p_ranks = n_ranks;
  return KPDI_OK;
}

}  // extern "C"

// ---- hooks for in-process groups of contexts (group.hip; declared in group_hooks.h) -------------------------------
namespace kpdi {


// one RCCL communicator over the contexts of ONE process (ncclCommInitAll: no unique id, no sockets, no environment)
int comm_init_all(kpdi_ctx *const *ctx, int n) {
  if (!ctx || n < 1) return fail(KPDI_EINVAL, "comm_init_all: no contexts");
  if (!g_rccl.load()) return fail(KPDI_ECOMM, "cannot load librccl: %s", g_rccl.why.c_str());
  std::vector<int> devs(n);
  for (int i = 0; i < n; ++i) {
    if (!ctx[i]) return fail(KPDI_EINVAL, "comm_init_all: context %d is NULL", i);
    if (ctx[i]->comm) return fail(KPDI_EINVAL, "comm_init_all: context %d already has a communicator", i);
    devs[i] = ctx[i]->device;
  }
  std::vector<ncclComm_t> comms(n, nullptr);
  ncclResult_t r = g_rccl.CommInitAll(comms.data(), n, devs.data());
  if (r != ncclSuccess) return fail(KPDI_ECOMM, "ncclCommInitAll over %d device(s): %s", n, g_rccl.GetErrorString(r));
  for (int i = 0; i < n; ++i) {
    ctx[i]->comm = comms[i];
    ctx[i]->rank = i;
    ctx[i]->nranks = n;
  }
  return KPDI_OK;
}

int finalize_precheck(kpdi_ctx *c, int kind) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  if (!c->have_exp || !c->have_problem) return fail(KPDI_EINVAL, "nothing to finalise");
  if (kind == FINALIZE_F64 && !c->exact64) return fail(KPDI_EINVAL, "kpdi_finalize_f64 needs a problem set up with KPDI_COMPUTE_F64");
  if (kind == FINALIZE_ASYNC && c->exact64)
    return fail(KPDI_EINVAL, "kpdi_finalize_async: not available in float64 arithmetic (use kpdi_finalize_f64)");
  if (kind == FINALIZE_ASYNC && c->m == 0) return fail(KPDI_EINVAL, "no experimental patterns to finalise");
  if (!c->exact64 && c->m > 0 && c->slots[0].pending && c->slots[1].pending)
    return fail(KPDI_EINVAL, "two results are already pending: collect one with kpdi_finalize_wait first");
  return KPDI_OK;
}

void gather_abandon(kpdi_ctx *c) {
  if (c) c->p2p_ranks = 0;
}

int64_t sweep_round_rows(const kpdi_ctx *c) {
  if (!c || !c->have_exp || c->m_pad <= 0) return 4096;
  return plan::round_rows(plan_env(c), c->m_pad / kpdi::TILE_EXP);
}

// RCCL gather, members other than the one that hands the result to the host: the all-gather + merge of
// kpdi_finalize without the copies (every rank of a collective has to take part in it)
int finalize_participate(kpdi_ctx *c) {
  if (!c) return fail(KPDI_EINVAL, "ctx is NULL");
  if (!c->have_exp || !c->have_problem) return fail(KPDI_EINVAL, "nothing to finalise");
  int rc = use_device(c);
  if (rc) return rc;
  if (c->m == 0) return KPDI_OK;
  if (c->exact64) {
    rc = ensure_running(c);
    return rc ? rc : finalize64(c, nullptr, nullptr, nullptr);
  }
  const float *d_s = nullptr;
  const int *d_i = nullptr;
  return final_lists(c, &d_s, &d_i);
}

// peer-copy gather, every member: its running lists are presentable and final - an event on its stream says when
int member_lists_ready(kpdi_ctx *c, ListsView *v) {
  if (!c || !v) return fail(KPDI_EINVAL, "NULL argument");
  if (!c->have_exp || !c->have_problem) return fail(KPDI_EINVAL, "nothing to finalise");
  int rc = use_device(c);
  if (rc) return rc;
  *v = ListsView{};
  v->device = c->device;
  v->f64 = c->exact64;
  v->n = (size_t)c->m * c->keep_n;
  if (c->m == 0) return KPDI_OK;
  rc = own_lists(c);
  if (rc) return rc;
  if (c->exact64) {
    v->scores = c->run64_s.p;
    v->idx = c->run64_i.as<int>();
  } else {
    v->scores = c->run_s[c->run_cur].p;
    v->idx = c->run_i[c->run_cur].as<int>();
  }
  if (!c->lists_final) HIPCHK(hipEventCreateWithFlags(&c->lists_final, hipEventDisableTiming));
  HIPCHK(hipEventRecord(c->lists_final, c->stream));
  v->ready = c->lists_final;
  return KPDI_OK;
}

// peer-copy gather, the root member: every member's lists -> the root's gather buffers (hipMemcpyPeerAsync on the
// root's stream behind the members' events; xGMI between devices, a plain device copy when members share a device).
// The root's next finalize merges them exactly like all-gathered ones.  *read_done: recorded behind the copies.
int root_gather_p2p(kpdi_ctx *c, const ListsView *v, int n, hipEvent_t *read_done) {
  if (!c || !v || n < 1 || !read_done) return fail(KPDI_EINVAL, "root_gather_p2p: bad arguments");
  int rc = use_device(c);
  if (rc) return rc;
  *read_done = nullptr;
  const size_t cnt = (size_t)c->m * c->keep_n;
  if (cnt == 0) return KPDI_OK;
  const size_t es = c->exact64 ? sizeof(double) : sizeof(float);
  DevBuf &gs = c->exact64 ? c->gather64_s : c->gather_s;
  DevBuf &gi = c->exact64 ? c->gather64_i : c->gather_i;
  for (int j = 0; j < n; ++j)
    if (v[j].n != cnt || v[j].f64 != c->exact64 || !v[j].scores || !v[j].idx)
      return fail(KPDI_EINVAL, "group member %d holds %zu list entries (%s), the root %zu (%s): the members of a group must "
                  "be set up alike", j, v[j].n, v[j].f64 ? "float64" : "float32", cnt, c->exact64 ? "float64" : "float32");
  HIPCHK(gs.reserve(cnt * n * es));
  HIPCHK(gi.reserve(cnt * n * sizeof(int)));
  {
    ScopedTimer t(c, &c->ev_comm);
    for (int j = 0; j < n; ++j) {
      HIPCHK(hipStreamWaitEvent(c->stream, v[j].ready, 0));
      HIPCHK(hipMemcpyPeerAsync((char *)gs.p + (size_t)j * cnt * es, c->device, v[j].scores, v[j].device, cnt * es, c->stream));
      HIPCHK(hipMemcpyPeerAsync((char *)gi.p + (size_t)j * cnt * sizeof(int), c->device, v[j].idx, v[j].device,
                                cnt * sizeof(int), c->stream));
    }
  }
  if (!c->peer_read) HIPCHK(hipEventCreateWithFlags(&c->peer_read, hipEventDisableTiming));
  HIPCHK(hipEventRecord(c->peer_read, c->stream));
  *read_done = c->peer_read;
  c->p2p_ranks = n;
  return KPDI_OK;
}

// peer-copy gather, the other members: whoever next writes this member's lists waits for the root's copies of them
void member_lists_borrowed(kpdi_ctx *c, hipEvent_t read_done) {
  if (c && read_done) c->result_copy = read_done;
}

int context_device(const kpdi_ctx *c) { return c ? c->device : -1; }
int context_gather_ranks(const kpdi_ctx *c) { return c ? (c->comm ? c->nranks : 0) : 0; }

}  // namespace kpdi
