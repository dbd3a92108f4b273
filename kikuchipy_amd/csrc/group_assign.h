// group_assign.h - which member of a kpdi_group takes which rows of a pushed dictionary chunk.  Pure host arithmetic
// (no HIP, no state beyond the caller's `load` array): exported as kpdi_group_assign_chunk for planning and for the
// CPU test-suite (tests/test_group_assign.py).
//
// The reference's loop hands the metric one chunk of `n_per_iteration` patterns at a time
// (indexing/_dictionary_indexing.py:100-128; the tutorial's call uses a tenth of the dictionary, 3044 patterns,
// doc/tutorials/pattern_matching.ipynb:582).  Cutting EVERY such chunk n_dev ways (rounds 3-4) left a member 380 patterns
// per launch - 3 of the 16 workgroup slots of a row block for one tile-time.  The rule here keeps chunks whole wherever
// the call is chunked and still reproduces the contiguous n_dev-way split of a single-pass call:
//
//   dictionary size known (kpdi_group_set_dictionary_size; the Python layer always knows it):
//     member i has a QUOTA = the i-th of n_dev near-equal parts of the dictionary (as the multi-process form's
//     shard_range).  A chunk goes to the member that has taken the fewest patterns so far (ties: lowest index) - but
//     never beyond that member's quota: what does not fit spills to the next least-loaded member.  Hence
//       one chunk = the whole dictionary  -> member i gets the contiguous i-th part (the old split, exactly);
//       33 chunks of 3044 on 8 members    -> chunks 0..7 to members 0..7, 8..15 again, ... (whole chunks, round robin;
//                                            consecutive chunks land on different members, so the upload / generation
//                                            of one overlaps the sweep of the other), and the last, shorter chunk is
//                                            cut so that every member ends on exactly its quota;
//       4 chunks of 25 000 on 8 members   -> halves: (0,1) (2,3) (4,5) (6,7).
//   size unknown (plain C callers that never said): a chunk is cut into min(n_dev, n_chunk / min_piece) near-equal
//     pieces (at least one) for the least-loaded members - `min_piece` = two tile rounds of a member's sweep, below
//     which a piece wastes most of its launch.
//
// The merge of the members' lists is a total order (score desc, index asc), so the assignment never changes a result.
#pragma once
#include <cstdint>
#include <vector>

namespace kpdi {

struct ChunkPiece {
  int member;
  int64_t row0, rows;  // rows [row0, row0 + rows) of the chunk
};

inline void group_share(int64_t n, int i, int n_dev, int64_t *start, int64_t *end) {
  const int64_t base = n / n_dev, rem = n % n_dev;
  *start = i * base + (i < rem ? i : rem);
  *end = *start + base + (i < rem ? 1 : 0);
}

// `load[i]`: patterns member i has taken so far in this sweep (updated).  Pieces come out in row order.
inline std::vector<ChunkPiece> group_assign_chunk(int n_dev, int64_t n_total, int64_t min_piece, int64_t *load,
                                                  int64_t n_chunk) {
  std::vector<ChunkPiece> out;
  if (n_dev < 1 || n_chunk <= 0) return out;
  auto least_loaded = [&](bool with_room) {
    int best = -1;
    for (int i = 0; i < n_dev; ++i) {
      if (with_room) {
        int64_t a, b;
        group_share(n_total, i, n_dev, &a, &b);
        if (load[i] >= b - a) continue;
      }
      if (best < 0 || load[i] < load[best]) best = i;
    }
    return best;
  };
  if (n_total > 0) {
    int64_t row0 = 0, left = n_chunk;
    while (left > 0) {
      int j = least_loaded(true);
      int64_t take = left;
      if (j >= 0) {
        int64_t a, b;
        group_share(n_total, j, n_dev, &a, &b);
        take = left < (b - a) - load[j] ? left : (b - a) - load[j];
      } else {
        j = least_loaded(false);  // more patterns than announced: the remainder whole to the least-loaded member
      }
      if (!out.empty() && out.back().member == j) out.back().rows += take;
      else out.push_back({j, row0, take});
      load[j] += take;
      row0 += take;
      left -= take;
    }
    return out;
  }
  // size unknown: as many near-equal pieces as are still worth a launch set each, on the least-loaded members
  if (min_piece < 1) min_piece = 1;
  int p = (int)(n_chunk / min_piece < (int64_t)n_dev ? n_chunk / min_piece : (int64_t)n_dev);
  if (p < 1) p = 1;
  std::vector<char> chosen(n_dev, 0);
  for (int k = 0; k < p; ++k) {
    int best = -1;
    for (int i = 0; i < n_dev; ++i)
      if (!chosen[i] && (best < 0 || load[i] < load[best])) best = i;
    chosen[best] = 1;
  }
  int part = 0;
  for (int i = 0; i < n_dev; ++i) {
    if (!chosen[i]) continue;
    int64_t a, b;
    group_share(n_chunk, part++, p, &a, &b);
    out.push_back({i, a, b - a});
    load[i] += b - a;
  }
  return out;
}

}  // namespace kpdi
