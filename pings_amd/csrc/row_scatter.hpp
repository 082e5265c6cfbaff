// Deterministic scatter-add of gradient rows into a table (internal API; C-ABI: pings_rows_scatter_add).
//
//   out[r, 0:F] = sum over the pairs p with key[p] == r, in ascending p, of  w[p] * src[src_row[p] * ld + 0:F]
//   out[r, 0:F] = 0 for rows no pair points at
//
// This is the backward of the feature gather of `NeuralPoints.query_feature` (model/neural_gaussians.py:565-579:
// `feats[idx]` -> autograd index_put / scatter_add with atomics in the reference, i.e. run-to-run different bits).
// Here: counting sort of the pairs by destination row — histogram with INTEGER atomics, one exclusive scan,
// placement into the row's bucket, ranking inside the bucket by pair id — and one gather pass in which a group of F/4
// lanes owns a row, adds the row's pairs in ascending pair id and stores the row once, 16 B per lane.  Two forms of the
// pass: over the whole table (empty rows are stored as zeros by the same pass: no memset) when there are at least as
// many pairs as rows; over the BUCKETS (a memset of the table, then only the first position of every bucket works)
// when the table has more than four rows per pair — a 16,384-query step on a 1M-row table touches < 10 % of the rows.
// 5-6 launches instead of the 20-odd of a library merge sort, bitwise reproducible.
#pragma once
#include "common.hpp"

namespace pings_rows {

struct Plan {
  uint32_t* count;    // [rows + 1]  histogram, then consumed by the placement pass
  uint32_t* offset;   // [rows + 2]  exclusive scan of count
  uint32_t* bucket;   // [n]         pair ids grouped by destination row (arrival order inside a row's bucket)
  uint32_t* sorted;   // [n]         the same, every bucket in ascending pair id
  uint32_t* first;    // [n]         destination row at the first position of every bucket, NO_ROW elsewhere
  int64_t n;          // pairs the plan was carved for
  void* temp;         // scan scratch
  size_t temp_bytes;
  size_t total;
};

Plan carve(void* base, int64_t n_pairs, int64_t rows);

// keys[n]: destination row of every pair; keys >= rows are skipped.  Fills plan.offset / plan.bucket.
int build(const Plan& p, const uint32_t* keys, int64_t n, int64_t rows, hipStream_t st);

// F <= 64.  src_row == nullptr: pair p reads row p.  w == nullptr: weight 1.  `out` is [rows, F], every row written.
int gather_sum(const Plan& p, int64_t rows, int F, const float* src, int64_t ld, const uint32_t* src_row,
               const float* w, float* out, hipStream_t st);

}  // namespace pings_rows
