// Numeric kernels of the window, row-group, LDS-hash and global-scratch paths; dropping exact zeros.
// Part of the gfx950 SpGEMM kernels; included through kernels.h.
#pragma once
#include "kernels_walk.h"

namespace sg {

// ---- numeric -----------------------------------------------------------------------------------
struct NumOut {
  const uint64_t *__restrict__ c_ptr;
  uint32_t *__restrict__ c_col;
  double *__restrict__ c_val;
  unsigned long long *zero_count; // entries whose accumulated value is exactly 0.0
};

// numeric, window path: one wavefront per row.
// LDS per wavefront: bmp[words] of {lo: bitmap word, hi: number of set bits before the word}
//                    | vals[cap] doubles | FLAT only: owner[cap] uint32 + flat-walk scratch.
template <bool FLAT>
__global__ void __launch_bounds__(256)
    k_num_window(const uint32_t *__restrict__ rows, uint32_t nrows, Csr A, Csr B, RowMeta m, NumOut o,
                 uint32_t words, uint32_t cap, uint32_t wpb, uint32_t hw_order) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // one row per wavefront: scalar row state
  uint32_t idx = xcd_remap(blockIdx.x, gridDim.x) * wpb + w;
  if (w >= wpb || idx >= nrows) return;
  const size_t region = (size_t)words * 8u + (size_t)cap * 8u + (FLAT ? (size_t)cap * 4u + FLAT_BYTES : 0u);
  uint64_t *bmp = reinterpret_cast<uint64_t *>(smem + w * region);
  uint32_t *bmp32 = reinterpret_cast<uint32_t *>(bmp);
  double *vals = reinterpret_cast<double *>(bmp + words);
  uint32_t *owner = reinterpret_cast<uint32_t *>(vals + cap);
  const uint32_t row = rows[idx];
  const uint32_t base = m.cmin[row] & ~31u;
  const uint32_t nw = ((m.cmax[row] - base) >> 5) + 1u;
  const uint32_t n = m.nnz[row];
  const uint64_t off = o.c_ptr[row];
  const uint32_t *tin = m.temp_cols + m.t_off[m.lead[row]];
  for (uint32_t i = lane; i < nw; i += 64) bmp[i] = 0ull;
  for (uint32_t i = lane; i < n; i += 64) {
    vals[i] = 0.0;
    if constexpr (FLAT) if (!hw_order) owner[i] = 0u;
  }
  lds_fence();
  for (uint32_t p = lane; p < n; p += 64) {
    uint32_t c = tin[p];
    o.c_col[off + p] = c;
    uint32_t cc = c - base;
    atomicOr(&bmp32[(cc >> 5) * 2u], 1u << (cc & 31u));
  }
  lds_fence();
  uint32_t carry = 0;
  for (uint32_t r0 = 0; r0 < nw; r0 += 64) {
    uint32_t wi = r0 + lane;
    uint32_t pc = wi < nw ? __popc(bmp32[wi * 2u]) : 0u;
    uint32_t incl = group_incl_scan<64>(pc);
    if (wi < nw) bmp32[wi * 2u + 1u] = carry + incl - pc;
    carry += group_last<64>(incl);
  }
  lds_fence();
  auto position = [&](uint32_t c) {
    uint32_t cc = c - base;
    uint64_t e = bmp[cc >> 5];
    return (uint32_t)(e >> 32) + __popc((uint32_t)e & ((1u << (cc & 31u)) - 1u));
  };
  if constexpr (FLAT) {
    uint32_t stamp = 0;
    flat_walk<true>(A, B, row, lane, reinterpret_cast<unsigned char *>(owner + cap),
                    [&](uint32_t c, double bv, double av, uint32_t) {
                      const bool live = c != EMPTY_KEY;
                      const uint32_t pos = live ? position(c) : 0u;
                      if (stamp > CLAIM_STAMP_LIMIT) { // wave-uniform: start the tags over before the stamp field overflows
                        lds_fence();
                        for (uint32_t i = lane; i < n; i += 64) owner[i] = 0u;
                        stamp = 0;
                        lds_fence();
                      }
                      claim_in_lane_order(vals, owner, stamp, lane, live && pos < n, pos, __dmul_rn(av, bv), hw_order != 0);
                    });
  } else {
    wave_row_walk<true>(A, B, row, lane, [&](uint32_t c, double bv, double av, uint32_t) {
      const uint32_t pos = position(c);
      if (pos < n) lds_add_f64(&vals[pos], __dmul_rn(av, bv));
    });
  }
  lds_fence();
  uint32_t nz = 0;
  for (uint32_t p = lane; p < n; p += 64) {
    double v = vals[p];
    o.c_val[off + p] = v;
    nz += (v == 0.0) ? 1u : 0u;
  }
  nz = group_sum<64>(nz);
  if (lane == 0 && nz) atomicAdd(o.zero_count, (unsigned long long)nz);
}

// numeric, window path, row groups: one wavefront per group of up to GRP_MAX consecutive rows of
// A that share their column list.  Every B entry is loaded once and its output position is found
// once; it is then multiplied by each row's own A value (staged in LDS) and added to that row's
// accumulator.  Each row still receives its products in the order of its A row, as on the CPU.
// Column -> output position, two forms:
// COMPACT = false: a flat bitmap over the row's column hull, {bits, outputs before} per 32 columns;
//   a look-up is one 8-byte LDS read and a population count.
// COMPACT = true: a bitmap in two levels, for a hull that is wide and almost empty (a 3-D stencil:
//   375 outputs in a hull of 65,536 columns, where the flat bitmap is 16 KiB of a 31-KiB region and
//   the LDS a wavefront takes decides how many of them a CU holds): top[] has one bit per 32-column
//   word of the hull that holds an output, with the count of such words before it; cw[] holds those
//   words only ({bits, outputs before the word}).  A look-up is two dependent 8-byte LDS reads and
//   two population counts.  dielFilterV3real-like rows: 8 instead of 5 wavefronts per CU, 13.6 ms
//   instead of 20.0; af_shell10-like rows (hull of 16,384 columns, kernel bound by instruction
//   issue): 3.6 ms instead of 3.2, so the host picks the form per launch (grp_use_compact).
// The A values of the group's rows stay in registers, entry t of a row in lane t % 64 (the walk
// takes one A entry at a time, wave-uniformly, so a v_readlane hands the value to the scalar unit;
// staged in LDS they cost a read per row and B entry and 2.5 KiB per wavefront).
// LDS per wavefront: bitmap (flat: bmp[words]; compact: top[grp_top_words(words)] | cw[cap]) |
// vals[gmax][cap]; the A rows have at most GRP_ALEN entries for TIER 0, GRP_ALEN_MID for TIER 1 and
// GRP_ALEN_WIDE for TIER 2.
__host__ __device__ inline uint32_t grp_top_words(uint32_t words) { return (words + 31u) / 32u; }
// gmax: rows a group of the launch has at most (from the analysis, <= GRP_MAX / GRP_MAX_WIDE)
__host__ __device__ inline size_t grp_region(bool compact, uint32_t words, uint32_t cap, uint32_t gmax, uint32_t alen) {
  const size_t bitmap = compact ? (size_t)grp_top_words(words) * 8u + (size_t)cap * 8u : (size_t)words * 8u;
  (void)alen;
  return bitmap + (size_t)gmax * cap * 8u;
}
// the two-level form where the flat bitmap would be more than a third of the wavefront's LDS
__host__ __device__ inline bool grp_use_compact(uint32_t words, uint32_t cap, uint32_t gmax, uint32_t alen) {
  return (size_t)words * 8u * 3u > grp_region(false, words, cap, gmax, alen);
}
// The rank table (column -> output position) of a group, built from its SORTED column list: lane l of
// register u holds the column at position p = 64 u + l.  Columns of one 32-column word sit in
// neighbouring lanes, so the first lane of a word knows the word's "outputs before" (its own p), the
// number of words that began before it is a population count over the lanes (its index among the
// words with outputs, for the two-level form), and the first lane of a top word knows that word's
// "words before": one round of LDS stores and ORs, no scan over the table and no look-up while it
// is built.  top[] / cw[] must be zero (compact: cw[0 .. n), flat: all words of the hull) and fenced.
// emit(p, c) is called by the lanes that hold a column.  Returns the number of words with outputs.
template <bool COMPACT, uint32_t TCN, class Emit>
__device__ __forceinline__ uint32_t grp_rank_build(const uint32_t (&tc)[TCN], uint32_t n, uint32_t base, uint32_t lane,
                                                   uint32_t *top32, uint32_t *cw32, Emit emit) {
  uint32_t ncw = 0, prev_wi = 0xFFFFFFFFu; // wave-uniform: words so far, word of the last column so far
  const uint64_t le = (2ull << lane) - 1ull; // lanes up to this one
#pragma unroll
  for (uint32_t u = 0; u < TCN; ++u) {
    if (64u * u < n) { // wave-uniform
      const uint32_t p = 64u * u + lane;
      const bool act = p < n;
      const uint32_t c = tc[u];
      if (act) emit(p, c);
      const uint32_t cc = c - base, wi = cc >> 5;
      const uint32_t wi_before = dpp_move<0x138>(prev_wi, wi); // the lane before (wave_shr:1); lane 0: the register before
      const bool word_start = act && wi != wi_before;
      const uint64_t bal = __ballot(word_start);
      if constexpr (COMPACT) {
        const uint32_t ci = ncw + (uint32_t)__popcll(bal & le) - 1u; // (the first column starts a word: never below zero)
        if (act) atomicOr(&cw32[ci * 2u], 1u << (cc & 31u));
        if (word_start) {
          cw32[ci * 2u + 1u] = p;
          atomicOr(&top32[(wi >> 5) * 2u], 1u << (wi & 31u));
          if ((wi >> 5) != (wi_before >> 5)) top32[(wi >> 5) * 2u + 1u] = ci;
        }
      } else {
        if (act) atomicOr(&cw32[wi * 2u], 1u << (cc & 31u));
        if (word_start) cw32[wi * 2u + 1u] = p;
      }
      ncw += (uint32_t)__popcll(bal);
      prev_wi = (uint32_t)__builtin_amdgcn_readlane((int)wi, 63);
    }
  }
  return ncw;
}
template <int TIER, bool COMPACT, bool TAIL = false> // TAIL: see RowWalk
__global__ void __launch_bounds__(256)
    k_num_window_grp(const uint32_t *__restrict__ rows, uint32_t nrows, Csr A, Csr B, RowMeta m, NumOut o,
                     uint32_t words, uint32_t cap, uint32_t wpb, uint32_t gmax) {
  constexpr uint32_t alen = TIER == 0 ? GRP_ALEN : TIER == 1 ? GRP_ALEN_MID : GRP_ALEN_WIDE;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t idx = xcd_remap(blockIdx.x, gridDim.x) * wpb + w;
  if (w >= wpb || idx >= nrows) return;
  const size_t region = grp_region(COMPACT, words, cap, gmax, alen);
  const uint32_t topw = COMPACT ? grp_top_words(words) : 0u;
  uint64_t *top = reinterpret_cast<uint64_t *>(smem + w * region); // {words with an output (bits), such words before}
  uint32_t *top32 = reinterpret_cast<uint32_t *>(top);
  uint64_t *cw = top + topw; // compact: {outputs in the word (bits), outputs before} of the words with outputs; flat: of all words
  uint32_t *cw32 = reinterpret_cast<uint32_t *>(cw);
  double *vals = reinterpret_cast<double *>(cw + (COMPACT ? cap : words));
  constexpr uint32_t gmaxc = TIER ? GRP_MAX_WIDE : GRP_MAX; // rows a group of this tier can have
  // row state through the scalar unit (all of it was written by earlier kernels)
  const uint32_t row = scalar_view(rows)[idx];
  const uint32_t cnt = min((uint32_t)scalar_view(m.gcnt)[row], gmax);
  const uint32_t base = scalar_view(m.cmin)[row] & ~31u;
  const uint32_t nw = ((scalar_view(m.cmax)[row] - base) >> 5) + 1u;
  const uint32_t n = scalar_view((const uint32_t *)m.nnz)[row];
  const uint32_t *tin = m.temp_cols + scalar_view(m.t_off)[row];
  // first loads of the B walk, overlapped with the LDS set-up below
  RowWalk<true, false, TAIL> walk;
  walk.begin(A, B, row, lane);
  const uint32_t len_a = (uint32_t)(walk.a1 - walk.a0);
  // lane j holds the output offset and the A offset of the group's j-th row
  uint64_t my_off = 0, my_a = 0;
  if (lane < cnt) {
    my_off = o.c_ptr[row + lane];
    my_a = A.ptr[row + lane];
  }
  const uint32_t ntop = (nw + 31u) >> 5;
  if constexpr (COMPACT) {
    for (uint32_t i = lane; i < ntop; i += 64) top[i] = 0ull;
    for (uint32_t i = lane; i < n; i += 64) cw[i] = 0ull; // at most one word per output
  } else {
    for (uint32_t i = lane; i < nw; i += 64) cw[i] = 0ull;
  }
  double av_lo[gmaxc], av_hi[gmaxc]; // A values of row j: entries lane and 64 + lane
#pragma unroll
  for (uint32_t j = 0; j < gmaxc; ++j) {
    av_lo[j] = 0.0;
    av_hi[j] = 0.0;
    if (j < cnt) {
      for (uint32_t i = lane; i < n; i += 64) vals[j * cap + i] = 0.0;
      const double *av = A.val + readlane64(my_a, j);
      if (lane < len_a) av_lo[j] = av[lane];
      if constexpr (alen > 64)
        if (64u + lane < len_a) av_hi[j] = av[64u + lane];
    }
  }
  // the group's column list (at most GRP_CAP_TOP entries), loaded in one go: it is used twice below, and
  // a loop with a load per round pays the round trip every time
  constexpr uint32_t TCN = GRP_CAP_TOP / 64;
  uint32_t tc[TCN];
#pragma unroll
  for (uint32_t u = 0; u < TCN; ++u) tc[u] = 64u * u + lane < n ? tin[64u * u + lane] : 0u;
  lds_fence();
  grp_rank_build<COMPACT, TCN>(tc, n, base, lane, top32, cw32, [&](uint32_t p, uint32_t c) {
#pragma unroll
    for (uint32_t j = 0; j < GRP_MAX; ++j)
      if (j < cnt) o.c_col[readlane64(my_off, j) + p] = c;
  });
  // index in cw[] of the 32-column word wi (compact: a word that holds an output)
  auto compact_word = [&](uint32_t wi) {
    if constexpr (!COMPACT) return wi;
    const uint64_t t = top[wi >> 5];
    return (uint32_t)(t >> 32) + __popc((uint32_t)t & ((1u << (wi & 31u)) - 1u));
  };
  lds_fence();
  walk.run(A, B, lane, [&](uint32_t c, double bv, double, uint32_t apos) {
    const uint32_t cc = c - base;
    const uint64_t e = cw[compact_word(cc >> 5)];
    const uint32_t pos = (uint32_t)(e >> 32) + __popc((uint32_t)e & ((1u << (cc & 31u)) - 1u));
    const uint32_t ap = (uint32_t)__builtin_amdgcn_readfirstlane((int)apos); // the A entry: the same for every lane
#pragma unroll
    for (uint32_t j = 0; j < gmaxc; ++j)
      if (j < cnt) { // wave-uniform
        double a;
        if constexpr (alen > 64) a = ap < 64u ? readlane_f64(av_lo[j], ap) : readlane_f64(av_hi[j], ap - 64u);
        else a = readlane_f64(av_lo[j], ap);
        lds_add_f64(&vals[j * cap + pos], __dmul_rn(a, bv));
      }
  });
  lds_fence();
  uint32_t nz = 0;
#pragma unroll
  for (uint32_t j = 0; j < GRP_MAX; ++j) {
    if (j < cnt) {
      const uint64_t off = readlane64(my_off, j);
      for (uint32_t p = lane; p < n; p += 64) {
        double v = vals[j * cap + p];
        o.c_val[off + p] = v;
        nz += (v == 0.0) ? 1u : 0u;
      }
    }
  }
  nz = group_sum<64>(nz);
  if (lane == 0 && nz) atomicAdd(o.zero_count, (unsigned long long)nz);
}

// The same accumulation with the A row walked run by run (RowRuns / RunPipe): where consecutive A
// entries select consecutive rows of B with ONE column list (the unknowns of a mesh node: 5 rows in
// a row for af_shell10, 3 for dielFilterV3real), the output position of a column is found once per
// run instead of once per B row, and every accumulator it feeds is read from LDS once, takes the
// run's products one after the other in registers - the order of the CPU loop - and is written
// back once: two plain LDS accesses per row of the group and run instead of one LDS floating-point
// add (a read-modify-write that keeps the LDS busy five times as long) per row and B entry.  A run
// of one B row takes the single LDS add as before.  The row's description sits in registers, so
// the B entries of two units are in flight while a third is consumed.
// WIDE: A rows of up to 128 entries, else 64.  GMAX: rows a group of the launch has at most; the LDS
// holds GMAX accumulator rows per wavefront, and the inner loops run over all GMAX of them without
// a test per row (a group of fewer rows multiplies zeros into rows that are never written out).
#ifndef SG_GRP_SLOTS
#define SG_GRP_SLOTS 2
#endif
#ifndef SG_GRP_ALDS
#define SG_GRP_ALDS 0
#endif
#ifndef SG_GRP_WPE
#define SG_GRP_WPE 1
#endif
constexpr int GRP_RUN_SLOTS = SG_GRP_SLOTS;
// SG_GRP_ALDS: the A values of the group's rows are staged in LDS and read back as broadcasts (one LDS
// read per row of the group and B row) instead of being handed from a lane to the scalar unit (two
// lane reads): fewer vector instructions, more LDS traffic and 8 * GMAX * 64 (128) bytes of LDS more.
constexpr bool GRP_RUN_ALDS = SG_GRP_ALDS != 0;
// SG_GRP_ASCALAR: 1: the A values come straight from memory through the scalar unit (the address is
// wave-uniform) instead of two lane reads of the vector unit per value; 0: never; default: for the
// wide groups (rows of 65 to 128 entries; measured: the numeric pass of the 3-D grid rows 6.5 -> 6.1 ms,
// of the shell rows 1.14 -> 1.18 ms, so the narrow groups keep the lane reads).  Issuing the loads of
// the next B row behind the first use of this one's, so that they land during its additions, lost
// on both (scalar loads return in any order: every wait is for all that are out).
#ifndef SG_GRP_ASCALAR
#define SG_GRP_ASCALAR 2
#endif
__host__ __device__ inline size_t grp_run_region(bool compact, uint32_t words, uint32_t cap, uint32_t gmax, bool wide) {
  return grp_region(compact, words, cap, gmax, 0) + (GRP_RUN_ALDS ? (size_t)gmax * (wide ? 128u : 64u) * 8u : 0u);
}
// TCN: registers of 64 columns that hold a group's column list (cap <= 64 * TCN)
// (the narrow groups need little LDS: registers for six wavefronts per SIMD keep the LDS the limit)
template <bool WIDE, bool COMPACT, int RMAX, int GMAX, int TCN_>
__global__ void __launch_bounds__(256, WIDE ? 1 : SG_GRP_WPE)
    k_num_grp_run(const uint32_t *__restrict__ rows, uint32_t nrows, Csr A, Csr B, const uint32_t *__restrict__ same_bits,
                  RowMeta m, NumOut o, uint32_t words, uint32_t cap, uint32_t wpb, uint32_t rpw) {
  // A wavefront takes rpw (<= 64) consecutive groups of its bin, one after the other.  What a group
  // needs before its first product sits behind a chain of dependent loads (row id -> row state ->
  // column indices of the A row -> bounds of the B rows -> B entries); that chain is taken off the
  // critical path: the row state of ALL the wavefront's groups is loaded once (lane i holds group i's),
  // the description of the next group's A row is loaded while this group is accumulated (the column
  // indices of the group after it as well), and so is the next group's column list.
  constexpr uint32_t gmax = GMAX;
#ifdef SG_GRP_STAMPS
  // diagnostic build: cycles per phase, summed over the wavefronts into o.zero_count[8..]
  unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = __builtin_amdgcn_s_memtime();
#define GR_STAMP(i) { const unsigned long long tn = __builtin_amdgcn_s_memtime(); tph[i] += tn - tlast; tlast = tn; }
#else
#define GR_STAMP(i)
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint64_t first64 = ((uint64_t)xcd_remap(blockIdx.x, gridDim.x) * wpb + w) * rpw;
  if (w >= wpb || first64 >= nrows) return;
  const uint32_t first = (uint32_t)first64, ng = min(rpw, nrows - first);
  const size_t region = grp_run_region(COMPACT, words, cap, gmax, WIDE);
  constexpr uint32_t ALEN = WIDE ? 128u : 64u;
  const uint32_t topw = COMPACT ? grp_top_words(words) : 0u;
  uint64_t *top = reinterpret_cast<uint64_t *>(smem + w * region);
  uint32_t *top32 = reinterpret_cast<uint32_t *>(top);
  uint64_t *cw = top + topw;
  uint32_t *cw32 = reinterpret_cast<uint32_t *>(cw);
  double *vals = reinterpret_cast<double *>(cw + (COMPACT ? cap : words));
  double *avals = vals + (size_t)gmax * cap; // (SG_GRP_ALDS) A values: avals[j * ALEN + t]
  constexpr uint32_t gmaxc = GMAX;
  static_assert(GMAX <= (int)(WIDE ? GRP_MAX_WIDE : GRP_MAX), "more rows than a group can have");
  // row state of the wavefront's groups: lane i holds group i's (the lanes past the last group repeat it)
  const uint32_t st_row = rows[first + min(lane, ng - 1u)];
  const uint32_t st_cmin = m.cmin[st_row], st_cmax = m.cmax[st_row];
  const uint64_t st_toff = m.t_off[st_row], st_a0 = A.ptr[st_row], st_coff = o.c_ptr[st_row];
  // rows of the group | entries of its A rows << 8 | outputs << 16, in one register
  const uint32_t st_pack = min((uint32_t)m.gcnt[st_row], gmax) | ((uint32_t)min(A.ptr[(uint64_t)st_row + 1u] - st_a0, (uint64_t)ALEN) << 8) |
                           ((uint32_t)m.nnz[st_row] << 16);
  auto rl32 = [](uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); };
  constexpr uint32_t TCN = TCN_;
  // the group's column list; a lane past its end re-reads the last entry (n >= 1 for every row of this bin)
  auto load_tc = [&](uint32_t g, uint32_t (&tc)[TCN]) {
    const uint32_t *tin = m.temp_cols + readlane64(st_toff, g);
    const uint32_t n = rl32(st_pack, g) >> 16;
#pragma unroll
    for (uint32_t u = 0; u < TCN; ++u) tc[u] = tin[min(64u * u + lane, n - 1u)];
  };
  using Runs = RowRuns<WIDE, WIDE, WIDE>; // (the narrow groups: units enumerated on the scalar unit; the wide ones: units of 128 columns, two per lane)
  Runs cur, nxt;
  auto len_of = [&](uint32_t g) { return (rl32(st_pack, g) >> 8) & 0xFFu; };
  cur.issue_a(A, readlane64(st_a0, 0), max(len_of(0), 1u), lane);
  {
    const uint32_t g1 = min(1u, ng - 1u);
    nxt.issue_a(A, readlane64(st_a0, g1), max(len_of(g1), 1u), lane);
  }
  cur.issue_b(B, same_bits);
  uint32_t tc[TCN];
  load_tc(0u, tc);
  uint32_t nz = 0;
  GR_STAMP(0) // row state of all groups, first descriptions issued
  for (uint32_t g = 0; g < ng; ++g) {
    const uint32_t g1 = min(g + 1u, ng - 1u), g2 = min(g + 2u, ng - 1u);
    // this group's scalars
    const uint32_t pack = rl32(st_pack, g);
    const uint32_t cnt = pack & 0xFFu;
    const uint32_t base = rl32(st_cmin, g) & ~31u;
    const uint32_t nw = ((rl32(st_cmax, g) - base) >> 5) + 1u;
    const uint32_t n = pack >> 16;
    const uint64_t a0 = readlane64(st_a0, g), coff = readlane64(st_coff, g);
    const uint32_t len_a = (pack >> 8) & 0xFFu;
    cur.len_a = len_a; // (a row without entries cannot be here; the loads above assumed one entry at least)
    // Where the A values come through the scalar unit, its loads sit on the path of every unit (a row's
    // values, a wait, the row's additions): the values must not come from HBM then.  One vector load
    // touches every 128-byte line of the group's values (its rows lie back to back: 64 lines at most,
    // one per lane) while the group is set up; its result is only waited for, after the units.
    uint32_t a_touch = 0;
    if constexpr (SG_GRP_ASCALAR == 1 || (SG_GRP_ASCALAR == 2 && WIDE)) {
      const uint32_t dwords = cnt * len_a * 2u; // (>= 2)
      a_touch = reinterpret_cast<const uint32_t *>(A.val + a0)[min(lane * 32u, dwords - 1u)];
    }
    cur.template finish<RMAX>(lane);
    // the next group's description: bounds of its B rows now, column indices of the group after it
    nxt.issue_b(B, same_bits);
    Runs aft;
    aft.issue_a(A, readlane64(st_a0, g2), max(len_of(g2), 1u), lane);
    GR_STAMP(1) // waited for this group's description
    const uint32_t ntop = (nw + 31u) >> 5;
    if constexpr (COMPACT) {
      for (uint32_t i = lane; i < ntop; i += 64) top[i] = 0ull;
      for (uint32_t i = lane; i < n; i += 64) cw[i] = 0ull;
    } else {
      for (uint32_t i = lane; i < nw; i += 64) cw[i] = 0ull;
    }
    // A values of row j: entries lane and 64 + lane (the rows of a group lie back to back in A and in C)
    double av_lo[gmaxc], av_hi[gmaxc];
#pragma unroll
    for (uint32_t j = 0; j < gmaxc; ++j) {
      av_lo[j] = 0.0;
      av_hi[j] = 0.0;
      if (j < cnt) {
        for (uint32_t i = lane; i < n; i += 64) vals[j * cap + i] = 0.0;
        const double *av = A.val + a0 + (uint64_t)j * len_a;
        if (lane < len_a) av_lo[j] = av[lane];
        if constexpr (WIDE)
          if (64u + lane < len_a) av_hi[j] = av[64u + lane];
      }
      if constexpr (GRP_RUN_ALDS) { // (rows past the group's end hold zeros)
        avals[j * ALEN + lane] = av_lo[j];
        if constexpr (WIDE) avals[j * ALEN + 64u + lane] = av_hi[j];
      }
    }
    // the first units' B entries: in flight during the rest of the set-up
    RunPipe<RMAX, true, GRP_RUN_SLOTS, WIDE, WIDE> pipe;
    pipe.begin(B, cur, lane);
    lds_fence();
    // (a row's base on the scalar unit, the position as a 32-bit lane offset: no 64-bit address per lane and row)
    typedef uint32_t __attribute__((address_space(1))) *global_u32_ptr;
    grp_rank_build<COMPACT, TCN>(tc, n, base, lane, top32, cw32, [&](uint32_t p, uint32_t c) {
#pragma unroll
      for (uint32_t j = 0; j < gmaxc; ++j)
        if (j < cnt) ((global_u32_ptr)(uintptr_t)(o.c_col + coff + (uint64_t)j * n))[p & 0xFFFFu] = c;
    });
    auto compact_word = [&](uint32_t wi) {
      if constexpr (!COMPACT) return wi;
      const uint64_t t = top[wi >> 5];
      return (uint32_t)(t >> 32) + __popc((uint32_t)t & ((1u << (wi & 31u)) - 1u));
    };
    lds_fence();
    // the next group's column list (this group's is used up)
    load_tc(g1, tc);
    GR_STAMP(2) // bitmap and prefix counts built, columns of C stored
    // value of A entry t of the group's row j (t wave-uniform)
    constexpr bool GRP_RUN_ASCALAR = SG_GRP_ASCALAR == 1 || (SG_GRP_ASCALAR == 2 && WIDE);
    // the rows' values through the scalar unit: one base address per row (a row past the group's end
    // reads row 0: its sums are never written out) and a 32-bit entry number, which the compiler turns
    // into loads of the form base + scalar offset + constant: the values of a run's entries, which lie
    // side by side, arrive in one or two loads per row, without an address computation each
    const_as_ptr<double> a_row[gmaxc];
#pragma unroll
    for (uint32_t j = 0; j < gmaxc; ++j) a_row[j] = scalar_view(A.val + a0 + (uint64_t)(j < cnt ? j : 0u) * len_a);
    auto a_of = [&](uint32_t j, uint32_t t) {
      if constexpr (GRP_RUN_ASCALAR) return a_row[j][(uint32_t)__builtin_amdgcn_readfirstlane((int)t) & 0xFFu]; // (t < 128)
      else if constexpr (GRP_RUN_ALDS) return avals[j * ALEN + t];
      else if constexpr (WIDE) return t < 64u ? readlane_f64(av_lo[j], t) : readlane_f64(av_hi[j], t - 64u);
      else return readlane_f64(av_lo[j], t);
    };
    pipe.run(B, cur, lane, [&](const RunInfo &ri, const RunVec<RMAX, WIDE> &v) {
      lds_fence(); // the unit before has left its sums in LDS
      // a whole run of a small group: the values of ALL rows are asked for first (a dozen at most:
      // they fit the scalar registers), then the rows' sums advance side by side - row after row,
      // each row's loads waited for the sums of the row before it, and the chains of additions
      // (six dependent operations a row) ran one after the other
      const bool whole_small = GRP_RUN_ASCALAR && ri.rl == (uint32_t)RMAX && GMAX * RMAX <= 12; // wave-uniform
      double a[gmaxc][RMAX];
      if (whole_small) {
        const uint32_t tb = (uint32_t)__builtin_amdgcn_readfirstlane((int)ri.t0) & 0x7Fu; // (t0 < 128)
#pragma unroll
        for (uint32_t j = 0; j < gmaxc; ++j)
#pragma unroll
          for (int r = 0; r < RMAX; ++r) a[j][r] = a_row[j][tb + (uint32_t)r];
      }
      // one column per lane of the unit (a wide group's unit has two: the second 64 columns of the B rows)
      auto half = [&](bool ok, uint32_t col, const double (&bv)[RMAX]) {
      if (ok) {
        const uint32_t cc = col - base;
        const uint64_t e = cw[compact_word(cc >> 5)];
        const uint32_t pos = (uint32_t)(e >> 32) + __popc((uint32_t)e & ((1u << (cc & 31u)) - 1u));
        double *acc_at = vals + pos;
        if (ri.rl == 1u) { // wave-uniform
#pragma unroll
          for (uint32_t j = 0; j < gmaxc; ++j) lds_add_f64(acc_at + j * cap, __dmul_rn(a_of(j, ri.t0), bv[0]));
        } else if (whole_small) {
          double acc[gmaxc];
#pragma unroll
          for (uint32_t j = 0; j < gmaxc; ++j) acc[j] = acc_at[j * cap];
#pragma unroll
          for (int r = 0; r < RMAX; ++r)
#pragma unroll
            for (uint32_t j = 0; j < gmaxc; ++j) acc[j] = __dadd_rn(acc[j], __dmul_rn(a[j][r], bv[r]));
#pragma unroll
          for (uint32_t j = 0; j < gmaxc; ++j) acc_at[j * cap] = acc[j];
        } else if (GRP_RUN_ASCALAR && ri.rl == (uint32_t)RMAX) { // a whole run: row after row, each row's values in one go
          const uint32_t tb = (uint32_t)__builtin_amdgcn_readfirstlane((int)ri.t0) & 0x7Fu; // (t0 < 128)
#pragma unroll
          for (uint32_t j = 0; j < gmaxc; ++j) {
            double a[RMAX];
#pragma unroll
            for (int r = 0; r < RMAX; ++r) a[r] = a_row[j][tb + (uint32_t)r]; // (one masked entry number + constants: adjacent loads merge)
            double acc = acc_at[j * cap];
#pragma unroll
            for (int r = 0; r < RMAX; ++r) acc = __dadd_rn(acc, __dmul_rn(a[r], bv[r]));
            acc_at[j * cap] = acc;
          }
        } else {
          double acc[gmaxc];
#pragma unroll
          for (uint32_t j = 0; j < gmaxc; ++j) acc[j] = acc_at[j * cap];
#pragma unroll
          for (int r = 0; r < RMAX; ++r)
            if ((uint32_t)r < ri.rl) { // wave-uniform
#pragma unroll
              for (uint32_t j = 0; j < gmaxc; ++j) acc[j] = __dadd_rn(acc[j], __dmul_rn(a_of(j, ri.t0 + (uint32_t)r), bv[r]));
            }
#pragma unroll
          for (uint32_t j = 0; j < gmaxc; ++j) acc_at[j * cap] = acc[j];
        }
      }
      };
      half(v.ok, v.col, v.bv);
      if constexpr (WIDE) half(v.ok2, v.col2, v.bv2);
    });
    lds_fence();
    asm volatile("" ::"v"(a_touch)); // (keeps the load above)
    GR_STAMP(3) // the units
#pragma unroll
    for (uint32_t j = 0; j < gmaxc; ++j) {
      if (j < cnt) {
        const uint64_t off = coff + (uint64_t)j * n;
        for (uint32_t p = lane; p < n; p += 64) {
          double v = vals[j * cap + p];
          o.c_val[off + p] = v;
          nz += (v == 0.0) ? 1u : 0u;
        }
      }
    }
    lds_fence();
    cur = nxt;
    nxt = aft;
    GR_STAMP(4) // values written
  }
  nz = group_sum<64>(nz);
  if (lane == 0 && nz) atomicAdd(o.zero_count, (unsigned long long)nz);
#ifdef SG_GRP_STAMPS
  if (lane == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(&o.zero_count[8 + i], tph[i]);
#endif
#undef GR_STAMP
}

// numeric, hash path: one wavefront per row.
// LDS per wavefront: vals[S/2] double | keys[S] uint32 | pos[S] uint16 (= 10*S bytes; nnz <= S/2)
//                    | FLAT only: owner[S/2] uint32 + flat-walk scratch.
// The sorted structural columns are inserted first (slot -> output position), so the
// accumulate loop only probes (no compare-and-swap) and the output needs no sort.
template <bool FLAT>
__global__ void __launch_bounds__(256)
    k_num_hash(const uint32_t *__restrict__ rows, uint32_t nrows, Csr A, Csr B, RowMeta m, NumOut o,
               uint32_t log2s, uint32_t wpb, uint32_t hw_order) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t idx = xcd_remap(blockIdx.x, gridDim.x) * wpb + w;
  if (w >= wpb || idx >= nrows) return;
  const uint32_t S = 1u << log2s, mask = S - 1u;
  const size_t region = (size_t)S * 10u + (FLAT ? (size_t)S * 2u + FLAT_BYTES : 0u);
  unsigned char *reg = smem + w * region;
  double *vals = reinterpret_cast<double *>(reg);                 // S/2 doubles = 4*S bytes
  uint32_t *keys = reinterpret_cast<uint32_t *>(reg + 4u * S);    // S uint32
  uint16_t *pos = reinterpret_cast<uint16_t *>(reg + 8u * S);     // S uint16
  uint32_t *owner = reinterpret_cast<uint32_t *>(reg + 10u * S);  // S/2 uint32 (FLAT)
  const uint32_t row = rows[idx];
  const uint32_t n = m.nnz[row];
  const uint64_t off = o.c_ptr[row];
  const uint32_t *tin = m.temp_cols + m.t_off[m.lead[row]];
  for (uint32_t i = lane; i < S; i += 64) keys[i] = EMPTY_KEY;
  for (uint32_t i = lane; i < n; i += 64) {
    vals[i] = 0.0;
    if constexpr (FLAT) if (!hw_order) owner[i] = 0u;
  }
  lds_fence();
  for (uint32_t p = lane; p < n; p += 64) {
    uint32_t c = tin[p];
    o.c_col[off + p] = c;
    uint32_t h = hash_slot(c, log2s);
    uint32_t it = 0;
    while (atomicCAS(&keys[h], EMPTY_KEY, c) != EMPTY_KEY && it < S) { h = (h + 1u) & mask; ++it; }
    if (it < S) pos[h] = (uint16_t)p;
  }
  lds_fence();
  // slot of column c (S when absent, which cannot happen for a consistent symbolic result)
  auto find = [&](uint32_t c) {
    uint32_t h = hash_slot(c, log2s);
    uint32_t it = 0;
    while (keys[h] != c && it < S) { h = (h + 1u) & mask; ++it; }
    return it < S ? h : S;
  };
  if constexpr (FLAT) {
    uint32_t stamp = 0;
    flat_walk<true>(A, B, row, lane, reg + 12u * S, [&](uint32_t c, double bv, double av, uint32_t) {
      const uint32_t h = c != EMPTY_KEY ? find(c) : S;
      const bool live = h < S;
      if (stamp > CLAIM_STAMP_LIMIT) { // wave-uniform: start the tags over before the stamp field overflows
        lds_fence();
        for (uint32_t i = lane; i < n; i += 64) owner[i] = 0u;
        stamp = 0;
        lds_fence();
      }
      claim_in_lane_order(vals, owner, stamp, lane, live, live ? (uint32_t)pos[h] : 0u, __dmul_rn(av, bv), hw_order != 0);
    });
  } else {
    wave_row_walk<true>(A, B, row, lane, [&](uint32_t c, double bv, double av, uint32_t) {
      const uint32_t h = find(c);
      if (h < S) lds_add_f64(&vals[pos[h]], __dmul_rn(av, bv));
    });
  }
  lds_fence();
  uint32_t nz = 0;
  for (uint32_t p = lane; p < n; p += 64) {
    double v = vals[p];
    o.c_val[off + p] = v;
    nz += (v == 0.0) ? 1u : 0u;
  }
  nz = group_sum<64>(nz);
  if (lane == 0 && nz) atomicAdd(o.zero_count, (unsigned long long)nz);
}

// numeric, global path: ONE wavefront per row (the k order must be kept), accumulating straight
// into the row's slice of c_val through L2 atomics.  scratch: per workgroup `words_total`
// {bitmap word, prefix} pairs, all zero on entry and on exit.
__global__ void __launch_bounds__(64)
    k_num_global(const uint32_t *__restrict__ rows, uint32_t nrows, Csr A, Csr B, RowMeta m, NumOut o,
                 unsigned long long *__restrict__ scratch, uint64_t words_total) {
  unsigned long long *bmp = scratch + (size_t)blockIdx.x * words_total;
  uint32_t *bmp32 = reinterpret_cast<uint32_t *>(bmp);
  const uint32_t lane = threadIdx.x;
  for (uint32_t idx = blockIdx.x; idx < nrows; idx += gridDim.x) {
    const uint32_t row = rows[idx];
    const uint32_t n = m.nnz[row];
    const uint64_t off = o.c_ptr[row];
    const uint32_t *tin = m.temp_cols + m.t_off[m.lead[row]];
    const uint32_t w0 = m.cmin[row] >> 5, w1 = m.cmax[row] >> 5;
    for (uint32_t p = lane; p < n; p += 64) {
      uint32_t c = tin[p];
      o.c_col[off + p] = c;
      __hip_atomic_store(&o.c_val[off + p], 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      atomicOr(&bmp32[(size_t)(c >> 5) * 2u], 1u << (c & 31u));
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
    uint32_t carry = 0;
    for (uint32_t r0 = w0; r0 <= w1; r0 += 64) {
      uint32_t wi = r0 + lane;
      uint32_t bits = 0;
      if (wi <= w1) bits = __hip_atomic_load(&bmp32[(size_t)wi * 2u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      uint32_t pc = __popc(bits);
      uint32_t incl = group_incl_scan<64>(pc);
      if (wi <= w1 && bits)
        __hip_atomic_store(&bmp32[(size_t)wi * 2u + 1u], carry + incl - pc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      carry += group_last<64>(incl);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
    wave_row_walk<true>(A, B, row, lane, [&](uint32_t c, double bv, double av, uint32_t) {
      unsigned long long e = __hip_atomic_load(&bmp[c >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      uint32_t pos = (uint32_t)(e >> 32) + __popc((uint32_t)e & ((1u << (c & 31u)) - 1u));
      if (pos < n) unsafeAtomicAdd(&o.c_val[off + pos], __dmul_rn(av, bv));
    });
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
    uint32_t nz = 0;
    for (uint32_t p = lane; p < n; p += 64) {
      double v = __hip_atomic_load(&o.c_val[off + p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      nz += (v == 0.0) ? 1u : 0u;
      uint32_t c = tin[p];
      __hip_atomic_store(&bmp[c >> 5], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    nz = group_sum<64>(nz);
    if (lane == 0 && nz) atomicAdd(o.zero_count, (unsigned long long)nz);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
  }
}

// ---- dropping exact zeros (rare) ----------------------------------------------------------------
__global__ void __launch_bounds__(256)
    k_count_nonzero(uint32_t rows, const uint64_t *__restrict__ c_ptr, const double *__restrict__ c_val,
                    uint32_t *__restrict__ keep) {
  constexpr int G = 8;
  const uint64_t row = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
  uint32_t lg = threadIdx.x & (G - 1);
  uint32_t cnt = 0;
  if (row < rows)
    for (uint64_t j = c_ptr[row] + lg; j < c_ptr[row + 1]; j += G) cnt += (c_val[j] != 0.0) ? 1u : 0u;
  cnt = group_sum<G>(cnt);
  if (row < rows && lg == 0) keep[row] = cnt;
}

// one wavefront per row, order-preserving
__global__ void __launch_bounds__(256)
    k_compact_rows(uint32_t rows, const uint64_t *__restrict__ old_ptr, const uint32_t *__restrict__ old_col,
                   const double *__restrict__ old_val, const uint64_t *__restrict__ new_ptr,
                   uint32_t *__restrict__ new_col, double *__restrict__ new_val) {
  const uint64_t row = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  uint32_t lane = threadIdx.x & 63;
  if (row >= rows) return;
  uint64_t j0 = old_ptr[row], j1 = old_ptr[row + 1];
  uint64_t out = new_ptr[row];
  for (uint64_t jb = j0; jb < j1; jb += 64) {
    uint64_t j = jb + lane;
    double v = j < j1 ? old_val[j] : 0.0;
    bool keepit = j < j1 && v != 0.0;
    unsigned long long mball = __ballot(keepit);
    if (keepit) {
      uint64_t p = out + __popcll(mball & ((1ull << lane) - 1ull));
      new_col[p] = old_col[j];
      new_val[p] = v;
    }
    out += __popcll(mball);
  }
}

} // namespace sg
