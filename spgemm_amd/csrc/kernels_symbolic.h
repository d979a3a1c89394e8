// Symbolic kernels of the window, LDS-hash and global-scratch paths.
// Part of the gfx950 SpGEMM kernels; included through kernels.h.
#pragma once
#include "kernels_walk.h"

namespace sg {

// ---- symbolic, window path: one wavefront per row ------------------------------------------------
// LDS per wavefront: a bitmap of `words` 32-bit words covering columns [cmin & ~31, cmax] and a list
// of the words that hold a column, 16 bits each (6 bytes per word: the LDS per wavefront sets
// how many of them a CU holds, and a hull of 65,536 columns went from 9 to 13 wavefronts per CU).
//
// Every product ORs its bit into the bitmap (lanes of one instruction that share a word are
// serialised by the LDS; combining them first in registers was measured and cost more than it
// saved).  The sorted column list is then written word by word of the words that hold a column.
// FLAT: flat walk (short B rows) instead of the scalar-unit walk; adds FLAT_BYTES of LDS per wavefront.
// same_bits (scalar walk; may be null): RowInfo::same of B - consecutive A entries that select consecutive rows
// of B with one column list are walked once.
// LDS per wavefront: bits u32[words] | list of non-empty words u16[words] (rounded up to 16 bytes) | flat-walk scratch
__host__ __device__ inline size_t sym_window_region(uint32_t words, bool flat) {
  return (((size_t)words * 6u + 15u) & ~(size_t)15u) + (flat ? FLAT_BYTES : 0u) + (((size_t)(words + 31u) / 32u * 4u + 15u) & ~(size_t)15u);
}
// (the last part: one bit per word of the bitmap, for k_sym_window_runs, behind everything else)
__host__ __device__ inline size_t sym_window_top_offset(uint32_t words, bool flat) {
  return (((size_t)words * 6u + 15u) & ~(size_t)15u) + (flat ? FLAT_BYTES : 0u);
}
// The sorted column list of a row from its bitmap; returns the number of columns.
// The words that hold a column (a hull of tens of thousands of columns holds a few hundred) are
// listed first: a round of 64 words costs one read and a ballot, no scan and nothing that the next
// round waits for.  The listed words are then counted, scanned and written out 64 at a time, every
// lane the columns of its own word: neighbouring lanes write neighbouring entries.
__device__ __forceinline__ uint32_t sym_window_emit(const uint32_t *bm, uint16_t *nzl /* one entry per word at most */, uint32_t nw,
                                                    uint32_t base, uint32_t *__restrict__ out, uint32_t lane) {
  uint32_t nz = 0;
  for (uint32_t r0 = 0; r0 < nw; r0 += 64) {
    const uint32_t wi = r0 + lane;
    const uint32_t word = wi < nw ? bm[wi] : 0u;
    const uint64_t mask = __ballot(word != 0u);
    if (word) nzl[nz + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = (uint16_t)wi;
    nz += (uint32_t)__popcll(mask);
  }
  lds_fence();
  uint32_t carry = 0;
  for (uint32_t s0 = 0; s0 < nz; s0 += 64) {
    const uint32_t si = s0 + lane;
    const uint32_t wi = si < nz ? (uint32_t)nzl[si] : 0u;
    uint32_t word = si < nz ? bm[wi] : 0u;
    const uint32_t pc = __popc(word);
    const uint32_t incl = group_incl_scan<64>(pc);
    uint32_t at = carry + incl - pc;
    carry += group_last<64>(incl);
    const uint32_t cbase = base + (wi << 5);
    while (word) {
      out[at++] = cbase + (uint32_t)__ffs(word) - 1u;
      word &= word - 1u;
    }
  }
  return carry;
}
// The same from a bitmap that comes with a second level (top: one bit per word that holds a column,
// set by the lane that found the word empty): the words that hold a column are read off the second
// level - one round for a hull of 65,536 columns instead of 32 over the bitmap - and both levels are
// left cleared for the next row, which then has nothing to clear either.
__device__ __forceinline__ uint32_t sym_window_emit_top(uint32_t *bm, uint32_t *top, uint16_t *nzl, uint32_t nw, uint32_t base,
                                                        uint32_t *__restrict__ out, uint32_t lane) {
  const uint32_t ntop = (nw + 31u) >> 5;
  uint32_t nz = 0;
  for (uint32_t t0 = 0; t0 < ntop; t0 += 64) {
    const uint32_t ti = t0 + lane;
    uint32_t tw = ti < ntop ? top[ti] : 0u;
    const uint32_t pc = __popc(tw);
    const uint32_t incl = group_incl_scan<64>(pc);
    uint32_t at = nz + incl - pc;
    nz += group_last<64>(incl);
    if (tw) top[ti] = 0u;
    while (tw) {
      nzl[at++] = (uint16_t)(ti * 32u + (uint32_t)__ffs(tw) - 1u);
      tw &= tw - 1u;
    }
  }
  lds_fence();
  uint32_t carry = 0;
  for (uint32_t s0 = 0; s0 < nz; s0 += 64) {
    const uint32_t si = s0 + lane;
    const uint32_t wi = si < nz ? (uint32_t)nzl[si] : 0u;
    uint32_t word = si < nz ? bm[wi] : 0u;
    if (si < nz) bm[wi] = 0u;
    const uint32_t pc = __popc(word);
    const uint32_t incl = group_incl_scan<64>(pc);
    uint32_t at = carry + incl - pc;
    carry += group_last<64>(incl);
    const uint32_t cbase = base + (wi << 5);
    while (word) {
      out[at++] = cbase + (uint32_t)__ffs(word) - 1u;
      word &= word - 1u;
    }
  }
  return carry;
}
// The group's rows share the pattern: same nnz; they are accumulated together when the group kernel
// has a tier for them, else one by one (each reading the leader's column list).
__device__ __forceinline__ void sym_window_finish(const Csr &A, const RowMeta &m, uint32_t row, uint32_t n, uint32_t lane) {
  const uint32_t cnt = m.gcnt[row];
  if (lane < cnt) {
    m.nnz[row + lane] = n;
    const uint32_t len_a = (uint32_t)min(A.ptr[row + 1] - A.ptr[row], (uint64_t)0xFFFFFFFFu);
    const uint32_t gkey = num_key_of(m.sym_key[row], n, m.hash_num_max, cnt, len_a);
    const bool grouped = gkey >= (uint32_t)NUM_GRP0 && gkey != KEY_NONE;
    m.num_key[row + lane] = (uint8_t)(grouped ? (lane == 0 ? gkey : KEY_NONE) : num_key_of(m.sym_key[row], n, m.hash_num_max));
  }
}
#ifndef SG_SYM_SLOTS
#define SG_SYM_SLOTS 3
#endif
constexpr int SYM_RUN_SLOTS = SG_SYM_SLOTS; // units of the run walk whose columns are in flight
template <bool FLAT, bool TAIL = false> // TAIL: see RowWalk (scalar walk only)
__global__ void __launch_bounds__(256)
    k_sym_window(const uint32_t *__restrict__ rows, uint32_t nrows, Csr A, Csr B, RowMeta m,
                 uint32_t words, const uint32_t *__restrict__ same_bits) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // wave-uniform on the scalar unit
  uint32_t idx = xcd_remap(blockIdx.x, gridDim.x) * wpb + w;
  if (idx >= nrows) return;
  const size_t region = sym_window_region(words, FLAT);
  uint32_t *bm = reinterpret_cast<uint32_t *>(smem + w * region);             // words x 32 columns
  uint16_t *pre = reinterpret_cast<uint16_t *>(smem + w * region + (size_t)words * 4u); // the words that hold a column
  const uint32_t row = rows[idx];
  const uint32_t base = m.cmin[row] & ~31u;
  const uint32_t nw = ((m.cmax[row] - base) >> 5) + 1u;
  for (uint32_t i = lane; i < nw; i += 64) bm[i] = 0u;
  lds_fence();
  auto mark = [&](uint32_t col, double, double, uint32_t) {
    if (col == EMPTY_KEY) return;
    uint32_t c = col - base;
    atomicOr(&bm[c >> 5], 1u << (c & 31u));
  };
  if constexpr (FLAT) {
    flat_walk<false>(A, B, row, lane, smem + w * region + sym_window_region(words, false), mark);
  } else {
    const uint64_t a0 = scalar_view(A.ptr)[row], a1 = scalar_view(A.ptr)[row + 1];
    if (same_bits && a1 - a0 <= 128u) { // wave-uniform
      // B has runs of rows with one column list and the A row is short: its description goes to
      // registers (RowRuns), only the first row of every run is walked, and the columns of
      // SYM_RUN_SLOTS - 1 of them are in flight while one is ORed in
      RowRuns<true> runs;
      runs.load<255>(A, B, same_bits, a0, (uint32_t)(a1 - a0), lane);
      RunPipe<255, false, SYM_RUN_SLOTS, true> pipe;
      pipe.begin(B, runs, lane);
      pipe.run(B, runs, lane, [&](const RunInfo &, const RunVec<1> &v) { mark(v.ok ? v.col : EMPTY_KEY, 0.0, 0.0, 0u); });
    } else {
      wave_row_walk<false, false, TAIL>(A, B, row, lane, mark, same_bits); // (an entry that repeats a column list is skipped)
    }
  }
  lds_fence();
  const uint32_t n = sym_window_emit(bm, pre, nw, base, m.temp_cols + m.t_off[row], lane);
  sym_window_finish(A, m, row, n, lane);
}

// The same for a B with runs of rows with one column list (same_bits), scalar walk: a wavefront takes
// rpw (<= 64) consecutive rows of its bin, one after the other.  What a row needs before its first
// column sits behind a chain of dependent loads (row id -> row state -> column indices of the A row ->
// bounds of the B rows -> B entries); the chain is taken off the critical path as in k_num_grp_run:
// lane i holds row i's state, the description of the next row is loaded while this one is walked.
// PAIR: the units are 128 columns of a B row, two per lane (for a B whose rows have more than 64 entries)
template <bool PAIR>
__global__ void __launch_bounds__(256)
    k_sym_window_runs(const uint32_t *__restrict__ rows, uint32_t nrows, Csr A, Csr B, RowMeta m, uint32_t words,
                      const uint32_t *__restrict__ same_bits, uint32_t wpb, uint32_t rpw) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint64_t first64 = ((uint64_t)xcd_remap(blockIdx.x, gridDim.x) * wpb + w) * rpw;
  if (w >= wpb || first64 >= nrows) return;
  const uint32_t first = (uint32_t)first64, ng = min(rpw, nrows - first);
  const size_t region = sym_window_region(words, false);
  uint32_t *bm = reinterpret_cast<uint32_t *>(smem + w * region);
  uint16_t *pre = reinterpret_cast<uint16_t *>(smem + w * region + (size_t)words * 4u);
  uint32_t *top = reinterpret_cast<uint32_t *>(smem + w * region + sym_window_top_offset(words, false));
  // both levels of the bitmap start empty, and every row leaves them so
  for (uint32_t i = lane; i < words; i += 64) bm[i] = 0u;
  for (uint32_t i = lane; i < (words + 31u) / 32u; i += 64) top[i] = 0u;
  const uint32_t st_row = rows[first + min(lane, ng - 1u)];
  const uint32_t st_cmin = m.cmin[st_row], st_cmax = m.cmax[st_row];
  const uint64_t st_toff = m.t_off[st_row], st_a0 = A.ptr[st_row];
  const uint64_t st_len64 = A.ptr[(uint64_t)st_row + 1u] - st_a0;
  const uint32_t st_len = (uint32_t)min(st_len64, (uint64_t)128u); // what the registers describe
  const uint32_t st_long = st_len64 > 128u ? 1u : 0u;              // such a row is walked through scalar loads
  auto rl32 = [](uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); };
  RowRuns<true, true, PAIR> cur, nxt;
  cur.issue_a(A, readlane64(st_a0, 0), max(rl32(st_len, 0), 1u), lane);
  {
    const uint32_t g1 = min(1u, ng - 1u);
    nxt.issue_a(A, readlane64(st_a0, g1), max(rl32(st_len, g1), 1u), lane);
  }
  cur.issue_b(B, same_bits);
  for (uint32_t g = 0; g < ng; ++g) {
    const uint32_t g2 = min(g + 2u, ng - 1u);
    const uint32_t row = rl32(st_row, g);
    const uint32_t base = rl32(st_cmin, g) & ~31u;
    const uint32_t nw = ((rl32(st_cmax, g) - base) >> 5) + 1u;
    const bool is_long = rl32(st_long, g) != 0u;
    cur.len_a = rl32(st_len, g); // (a row without entries cannot be here; the loads above assumed one entry at least)
    cur.template finish<255>(lane);
    nxt.issue_b(B, same_bits);
    RowRuns<true, true, PAIR> aft;
    aft.issue_a(A, readlane64(st_a0, g2), max(rl32(st_len, g2), 1u), lane);
    lds_fence();
    // (the LDS takes the lanes of one instruction that share a word one after the other: exactly one finds
    //  it empty and sets the word's bit in the second level - when the NEXT unit is marked, so that no
    //  unit waits for what its own atomic returns)
    uint32_t pend_w = 0, pend_old = 1u;
    auto settle = [&]() {
      if (pend_old == 0u) atomicOr(&top[pend_w >> 5], 1u << (pend_w & 31u));
      pend_old = 1u;
    };
    auto mark = [&](uint32_t col, double, double, uint32_t) {
      settle();
      if (col == EMPTY_KEY) return;
      const uint32_t c = col - base;
      pend_w = c >> 5;
      pend_old = atomicOr(&bm[pend_w], 1u << (c & 31u));
    };
    if (!is_long) { // wave-uniform
      RunPipe<255, false, SYM_RUN_SLOTS, true, PAIR> pipe;
      pipe.begin(B, cur, lane);
      pipe.run(B, cur, lane, [&](const RunInfo &, const RunVec<1, PAIR> &v) {
        mark(v.ok ? v.col : EMPTY_KEY, 0.0, 0.0, 0u);
        if constexpr (PAIR) mark(v.ok2 ? v.col2 : EMPTY_KEY, 0.0, 0.0, 0u);
      });
    } else {
      wave_row_walk<false, false, false>(A, B, row, lane, mark, same_bits); // (an entry that repeats a column list is skipped)
    }
    settle();
    lds_fence();
    const uint32_t n = sym_window_emit_top(bm, top, pre, nw, base, m.temp_cols + readlane64(st_toff, g), lane);
    sym_window_finish(A, m, row, n, lane);
    lds_fence(); // (both levels are empty again)
    cur = nxt;
    nxt = aft;
  }
}

// ---- symbolic, hash path: one wavefront per row ----------------------------------------------
// LDS per wavefront: keys[S] | cnt[S/2] | tk[S/2]  (uint32 each) = 8*S bytes, S = 1 << log2s.
// S >= 2 * nprod, so the table is at most half full.  Sorting is a bucket rank sort: a monotone
// map of the column into S/2 buckets, counting sort into bucket order, then each key finds its
// rank among the (few) keys of its bucket.
template <bool FLAT>
__global__ void __launch_bounds__(256)
    k_sym_hash(const uint32_t *__restrict__ rows, uint32_t nrows, Csr A, Csr B, RowMeta m,
               uint32_t log2s) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t idx = xcd_remap(blockIdx.x, gridDim.x) * wpb + w;
  if (idx >= nrows) return;
  const uint32_t S = 1u << log2s, NB = S >> 1, mask = S - 1u;
  const size_t region = (size_t)S * 8u + (FLAT ? FLAT_BYTES : 0u);
  uint32_t *keys = reinterpret_cast<uint32_t *>(smem + w * region);
  uint32_t *cnt = keys + S;
  uint32_t *tk = cnt + NB;
  const uint32_t row = rows[idx];
  for (uint32_t i = lane; i < S; i += 64) keys[i] = EMPTY_KEY;
  for (uint32_t i = lane; i < NB; i += 64) cnt[i] = 0u;
  lds_fence();
  uint32_t mine = 0;
  auto insert = [&](uint32_t c, double, double, uint32_t) {
    if (c == EMPTY_KEY) return;
    uint32_t h = hash_slot(c, log2s);
    for (uint32_t it = 0; it < S; ++it) { // bounded: the table is at most half full
      uint32_t old = atomicCAS(&keys[h], EMPTY_KEY, c);
      if (old == EMPTY_KEY) { ++mine; break; }
      if (old == c) break;
      h = (h + 1u) & mask;
    }
  };
  if constexpr (FLAT) flat_walk<false>(A, B, row, lane, smem + w * region + (size_t)S * 8u, insert);
  else wave_row_walk<false>(A, B, row, lane, insert);
  lds_fence();
  const uint32_t n = group_sum<64>(mine);
  const uint32_t cmin = m.cmin[row];
  const float scale = (float)NB / (float)(m.cmax[row] - cmin + 1u);
  // pass A: bucket counts
  for (uint32_t s = lane; s < S; s += 64) {
    uint32_t key = keys[s];
    if (key != EMPTY_KEY) {
      uint32_t b = min(NB - 1u, (uint32_t)((float)(key - cmin) * scale));
      atomicAdd(&cnt[b], 1u);
    }
  }
  lds_fence();
  // exclusive scan of cnt
  uint32_t carry = 0;
  for (uint32_t r0 = 0; r0 < NB; r0 += 64) {
    uint32_t v = cnt[r0 + lane]; // NB is a multiple of 64
    uint32_t incl = group_incl_scan<64>(v);
    cnt[r0 + lane] = carry + incl - v;
    carry += group_last<64>(incl);
  }
  lds_fence();
  // pass B: scatter into bucket order; afterwards cnt[b] = end of bucket b
  for (uint32_t s = lane; s < S; s += 64) {
    uint32_t key = keys[s];
    if (key != EMPTY_KEY) {
      uint32_t b = min(NB - 1u, (uint32_t)((float)(key - cmin) * scale));
      uint32_t p = atomicAdd(&cnt[b], 1u);
      tk[p] = key;
    }
  }
  lds_fence();
  // pass C: rank inside the bucket, write sorted
  uint32_t *out = m.temp_cols + m.t_off[row];
  for (uint32_t p = lane; p < n; p += 64) {
    uint32_t key = tk[p];
    uint32_t b = min(NB - 1u, (uint32_t)((float)(key - cmin) * scale));
    uint32_t start = b ? cnt[b - 1] : 0u, end = cnt[b];
    uint32_t rank = start;
    for (uint32_t q = start; q < end; ++q) rank += (tk[q] < key) ? 1u : 0u;
    out[rank] = key;
  }
  if (lane == 0) {
    m.nnz[row] = n;
    m.num_key[row] = (uint8_t)num_key_of(m.sym_key[row], n, m.hash_num_max);
  }
}

// ---- symbolic, hash path, large tables: one 256-thread workgroup per row ---------------------------
// The same algorithm as k_sym_hash with the four wavefronts of a workgroup sharing one table (the
// pattern of a row does not depend on the order in which its products are seen): a table of 8192
// slots takes 64 KiB of LDS, and one wavefront per table would leave two wavefronts on a CU.
// The wavefronts take the A entries round robin, except those whose B row has more than
// HWG_LONG entries: these (up to HWG_LONG_CAP of them) are put aside and then walked by all four
// wavefronts together - such rows have a few A entries with hub rows of B among them, and the
// wavefront that drew the hub kept the other three waiting for half of the kernel's time.
// LDS: keys[S] | cnt[S/2] | tk[S/2] = 8*S bytes.
constexpr uint32_t HWG_LONG = 256, HWG_LONG_CAP = 64;
// THREADS: 256, or 512 for the largest table (two workgroups of it fit a CU: 16 instead of 8 wavefronts)
template <int THREADS>
__global__ void __launch_bounds__(THREADS)
    k_sym_hash_wg(const uint32_t *__restrict__ rows, uint32_t nrows, Csr A, Csr B, RowMeta m, uint32_t log2s) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ uint32_t s_n, s_wsum[THREADS / 64], s_nlong, s_longk[HWG_LONG_CAP];
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const uint32_t S = 1u << log2s, NB = S >> 1, mask = S - 1u;
  uint32_t *keys = reinterpret_cast<uint32_t *>(smem);
  uint32_t *cnt = keys + S;
  uint32_t *tk = cnt + NB;
  const uint32_t row = scalar_view(rows)[xcd_remap(blockIdx.x, gridDim.x)];
  for (uint32_t i = tid; i < S; i += THREADS) keys[i] = EMPTY_KEY;
  for (uint32_t i = tid; i < NB; i += THREADS) cnt[i] = 0u;
  if (tid == 0) { s_n = 0u; s_nlong = 0u; }
  __syncthreads();
  const uint64_t a0 = scalar_view(A.ptr)[row], a1 = scalar_view(A.ptr)[row + 1];
  uint32_t mine = 0;
  auto insert = [&](uint32_t c) {
    uint32_t h = hash_slot(c, log2s);
    for (uint32_t it = 0; it < S; ++it) { // bounded: the table is at most half full
      const uint32_t old = atomicCAS(&keys[h], EMPTY_KEY, c);
      if (old == EMPTY_KEY) { ++mine; break; }
      if (old == c) break;
      h = (h + 1u) & mask;
    }
  };
  for (uint64_t ka = a0 + w; ka < a1; ka += THREADS / 64) {
    const uint32_t k = scalar_view(A.col)[ka];
    const uint64_t b0 = scalar_view(B.ptr)[k], b1 = scalar_view(B.ptr)[k + 1];
    if (b1 - b0 > HWG_LONG) { // wave-uniform
      uint32_t slot = 0;
      if (lane == 0) slot = atomicAdd(&s_nlong, 1u);
      slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
      if (slot < HWG_LONG_CAP) {
        if (lane == 0) s_longk[slot] = k;
        continue; // taken by all wavefronts below
      }
    }
    for (uint64_t j = b0 + lane; j < b1; j += 64) insert(B.col[j]);
  }
  __syncthreads();
  const uint32_t nlong = min(s_nlong, HWG_LONG_CAP);
  for (uint32_t i = 0; i < nlong; ++i) {
    const uint32_t k = s_longk[i];
    const uint64_t b0 = B.ptr[k], b1 = B.ptr[k + 1];
    for (uint64_t j = b0 + tid; j < b1; j += THREADS) insert(B.col[j]);
  }
  mine = group_last<64>(group_incl_scan<64>(mine));
  if (lane == 0 && mine) atomicAdd(&s_n, mine);
  __syncthreads();
  const uint32_t n = s_n;
  const uint32_t cmin = m.cmin[row];
  const float scale = (float)NB / (float)(m.cmax[row] - cmin + 1u);
  // pass A: bucket counts (buckets are monotone in the column)
  for (uint32_t s = tid; s < S; s += THREADS) {
    const uint32_t key = keys[s];
    if (key != EMPTY_KEY) atomicAdd(&cnt[min(NB - 1u, (uint32_t)((float)(key - cmin) * scale))], 1u);
  }
  __syncthreads();
  // exclusive scan of cnt: NB / THREADS consecutive buckets per thread (NB is a multiple of THREADS here)
  const uint32_t per = NB / THREADS;
  uint32_t local = 0;
  for (uint32_t i = 0; i < per; ++i) local += cnt[tid * per + i];
  const uint32_t incl = group_incl_scan<64>(local);
  if (lane == 63) s_wsum[w] = incl;
  __syncthreads();
  uint32_t run = incl - local;
  for (uint32_t i = 0; i < w; ++i) run += s_wsum[i];
  for (uint32_t i = 0; i < per; ++i) {
    const uint32_t v = cnt[tid * per + i];
    cnt[tid * per + i] = run;
    run += v;
  }
  __syncthreads();
  // pass B: scatter into bucket order; afterwards cnt[b] = end of bucket b
  for (uint32_t s = tid; s < S; s += THREADS) {
    const uint32_t key = keys[s];
    if (key != EMPTY_KEY) tk[atomicAdd(&cnt[min(NB - 1u, (uint32_t)((float)(key - cmin) * scale))], 1u)] = key;
  }
  __syncthreads();
  // pass C: rank inside the bucket, write sorted
  uint32_t *out = m.temp_cols + m.t_off[row];
  for (uint32_t p = tid; p < n; p += THREADS) {
    const uint32_t key = tk[p];
    const uint32_t b = min(NB - 1u, (uint32_t)((float)(key - cmin) * scale));
    const uint32_t start = b ? cnt[b - 1] : 0u, end = cnt[b];
    uint32_t rank = start;
    for (uint32_t q = start; q < end; ++q) rank += (tk[q] < key) ? 1u : 0u;
    out[rank] = key;
  }
  if (tid == 0) {
    m.nnz[row] = n;
    m.num_key[row] = (uint8_t)num_key_of(m.sym_key[row], n, m.hash_num_max);
  }
}

// ---- symbolic, global path: one 256-thread workgroup per row, bitmap in global scratch ---------
// scratch: per workgroup `words_total` uint32 words, all zero on entry and on exit.
__global__ void __launch_bounds__(256)
    k_sym_global(const uint32_t *__restrict__ rows, uint32_t nrows, Csr A, Csr B, RowMeta m,
                 uint32_t *__restrict__ scratch, uint64_t words_total) {
  __shared__ uint64_t s_wave[4];
  __shared__ uint32_t s_carry;
  uint32_t *bm = scratch + (size_t)blockIdx.x * words_total;
  const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (uint32_t idx = blockIdx.x; idx < nrows; idx += gridDim.x) {
    const uint32_t row = rows[idx];
    const uint32_t w0 = m.cmin[row] >> 5, w1 = m.cmax[row] >> 5;
    const uint64_t a0 = A.ptr[row], a1 = A.ptr[row + 1];
    for (uint64_t ka = a0 + w; ka < a1; ka += 4) {
      uint32_t k = A.col[ka];
      uint64_t b0 = B.ptr[k], b1 = B.ptr[k + 1];
      for (uint64_t jb = b0 + lane; jb < b1; jb += 64) {
        uint32_t c = B.col[jb];
        atomicOr(&bm[c >> 5], 1u << (c & 31u));
      }
    }
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    uint32_t *out = m.temp_cols + m.t_off[row];
    for (uint32_t r0 = w0; r0 <= w1; r0 += 256) {
      uint32_t wi = r0 + threadIdx.x;
      uint32_t bits = 0;
      if (wi <= w1) bits = atomicExch(&bm[wi], 0u); // read through L2 and clear
      uint32_t pc = __popc(bits);
      uint64_t tot;
      uint32_t ex = (uint32_t)block_excl_scan64(pc, s_wave, &tot);
      uint32_t pos = s_carry + ex;
      uint32_t cbase = wi << 5;
      while (bits) {
        uint32_t b = __ffs(bits) - 1;
        out[pos++] = cbase + b;
        bits &= bits - 1u;
      }
      __syncthreads();
      if (threadIdx.x == 0) s_carry += (uint32_t)tot;
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      uint32_t n = s_carry;
      m.nnz[row] = n;
      m.num_key[row] = (uint8_t)num_key_of(m.sym_key[row], n, m.hash_num_max);
    }
    __syncthreads();
  }
}

} // namespace sg
