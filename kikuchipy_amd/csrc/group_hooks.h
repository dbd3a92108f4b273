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
int finalize_participate(kpdi_ctx *c);
int member_lists_ready(kpdi_ctx *c, ListsView *v);
int root_gather_p2p(kpdi_ctx *root, const ListsView *v, int n, hipEvent_t *read_done);
void member_lists_borrowed(kpdi_ctx *c, hipEvent_t read_done);
int context_device(const kpdi_ctx *c);
int context_gather_ranks(const kpdi_ctx *c);

}  // namespace kpdi
