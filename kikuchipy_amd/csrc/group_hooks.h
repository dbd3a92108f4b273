// Internal interface between api.hip (one context = one GPU) and group.hip (kpdi_group: several contexts driven from one
// process).  Nothing here is part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

struct kpdi_ctx;

namespace kpdi {

// message of the calling thread's last failure (= kpdi_last_error()) and the way to set it from another translation unit
const char *thread_error();
int fail_msg(int code, const char *fmt, ...);

// a member's running best-k lists as the root of a peer-copy gather sees them
struct ListsView {
  int device = 0;
  const void *scores = nullptr;  // m * keep_n float (double in float64 arithmetic), device memory of `device`
  const int *idx = nullptr;
  size_t n = 0;                  // m * keep_n
  bool f64 = false;
  hipEvent_t ready = nullptr;    // recorded on the member's stream behind the last kernel that writes the lists
};

int comm_init_all(kpdi_ctx *const *ctx, int n);
// what would make the ROOT's finalize call refuse before it touches a device (no experimental set, wrong arithmetic for
// the entry point, both result slots taken): checked before any member queues its half of a collective
enum FinalizeKind { FINALIZE_F32 = 0, FINALIZE_F64 = 1, FINALIZE_ASYNC = 2 };
int finalize_precheck(kpdi_ctx *root, int kind);
// the root's finalize did not run (or failed): lists peer-copied for it must not wait for a later one
void gather_abandon(kpdi_ctx *root);
// dictionary patterns one full round of the context's sweep covers (every CU one tile): the unit below which a piece of
// a chunk wastes most of a launch
int64_t sweep_round_rows(const kpdi_ctx *c);
int finalize_participate(kpdi_ctx *c);
int member_lists_ready(kpdi_ctx *c, ListsView *v);
int root_gather_p2p(kpdi_ctx *root, const ListsView *v, int n, hipEvent_t *read_done);
void member_lists_borrowed(kpdi_ctx *c, hipEvent_t read_done);
int context_device(const kpdi_ctx *c);
int context_gather_ranks(const kpdi_ctx *c);

}  // namespace kpdi
