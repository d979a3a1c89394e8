// Per-row context and the two ways to walk the products of a row (scalar-unit walk, flat walk), in-order claims.
// Part of the gfx950 SpGEMM kernels; included through kernels.h.
#pragma once
#include "kernels_common.h"

namespace sg {

// ---- common per-row context ------------------------------------------------------------------
struct Csr {
  const uint64_t *__restrict__ ptr;
  const uint32_t *__restrict__ col;
  const double *__restrict__ val;
};

struct RowMeta {
  const uint32_t *__restrict__ cmin;
  const uint32_t *__restrict__ cmax;
  const uint8_t *__restrict__ sym_key;
  const uint64_t *__restrict__ t_off; // offset of the row's slot in temp_cols
  uint32_t *__restrict__ temp_cols;   // sorted structural columns per row (written by symbolic)
  uint32_t *__restrict__ nnz;         // structural nnz per row
  uint8_t *__restrict__ num_key;      // numeric bin key per row
  const uint8_t *__restrict__ gcnt;   // rows in the group led by this row (0: follower)
  const uint32_t *__restrict__ lead;  // leader of the row's group (the row itself when ungrouped)
  uint32_t hash_num_max;              // see num_key_of
  const uint64_t *__restrict__ a_p0;  // short rows: per A entry, start of its B row ...
  const uint32_t *__restrict__ a_len; // ... and the B row's length (written by the analysis)
  const uint64_t *__restrict__ a_pk;  // ... and the start of its padded {column, value} records (may be null)
};

// ---- walking the products of one row ----------------------------------------------------------
// Wave-uniform broadcast of a 64-bit lane value (index uniform across the wavefront).
__device__ __forceinline__ uint64_t readlane64(uint64_t v, uint32_t l) {
  uint32_t lo = __builtin_amdgcn_readlane((int)(uint32_t)v, (int)l);
  uint32_t hi = __builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)l);
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ double readlane_f64(double v, uint32_t l) {
  return __longlong_as_double((long long)readlane64((uint64_t)__double_as_longlong(v), l));
}

#ifndef SG_PF
#define SG_PF 4
#endif
constexpr int PF = SG_PF; // B rows whose first 64 entries are fetched before any of them is consumed

// Wave-per-row walk with the A row handled on the scalar unit.  `row` must be wave-uniform
// (derived through readfirstlane): the A entries, their values and the bounds of the B rows
// are then fetched by scalar loads and live in SGPRs, the B row base is a scalar pointer and a
// lane addresses its entry as base + lane * size, so the vector ALU does no index arithmetic
// at all.  The first 64 entries of PF consecutive B rows are fetched back to back; each B row
// is consumed completely, in A order, before the next one (the order of the CPU loop).
// f(col, bval, aval): bval/aval are only loaded when WITH_VAL.
// Read-only kernel inputs seen through the constant address space: a load whose address is
// wave-uniform is then issued on the scalar unit (s_load) instead of the vector memory path.
template <class T> using const_as_ptr = const T __attribute__((address_space(4))) *;
template <class T> __device__ __forceinline__ const_as_ptr<T> scalar_view(const T *p) {
  return (const_as_ptr<T>)(uintptr_t)p;
}

// f(col, bval, aval, apos): apos = position of the A entry inside its row (wave-uniform).
//
// The walk is software-pipelined over groups of PF A entries: while group g is consumed from
// registers, the vector loads of group g+1 are in flight and the scalar loads of group g+2 (A
// entries, then B row bounds) have been issued, so neither latency is exposed.
template <bool WITH_AVAL> struct WalkInfo {
  uint64_t p0[PF];   // start of the B row in B.col / B.val
  uint32_t len[PF];  // entries of the B row (0 past the end of the A row)
  double av[PF];     // value of the A entry
};

// same_bits (may be null): one bit per row of B, set when the row has the column list of the row
// before it.  A pattern-only walk then skips an A entry that selects the row after the one its
// predecessor selected when that bit is set: the entry adds no new column (a0: first entry of the row).
template <bool WITH_AVAL>
__device__ __forceinline__ void walk_scalar_stage(const Csr &A, const Csr &B, uint64_t ka, uint64_t a1,
                                                  WalkInfo<WITH_AVAL> &s, const uint32_t *same_bits = nullptr,
                                                  uint64_t a0 = 0) {
  const auto a_col = scalar_view(A.col);
  const auto a_val = scalar_view(A.val);
  const auto b_ptr = scalar_view(B.ptr);
  if (ka >= a1) {
#pragma unroll
    for (int u = 0; u < PF; ++u) { s.len[u] = 0; s.p0[u] = 0; s.av[u] = 0.0; }
    return;
  }
  uint32_t k[PF];
  if (ka + PF <= a1) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      k[u] = a_col[ka + u];
      if constexpr (WITH_AVAL) s.av[u] = a_val[ka + u];
    }
  } else { // the last group re-reads the row's last entry for its missing slots
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const uint64_t kk = min(ka + u, a1 - 1);
      k[u] = a_col[kk];
      if constexpr (WITH_AVAL) s.av[u] = a_val[kk];
    }
  }
  uint64_t p1[PF];
#pragma unroll
  for (int u = 0; u < PF; ++u) {
    s.p0[u] = b_ptr[k[u]];
    p1[u] = b_ptr[k[u] + 1];
  }
#pragma unroll
  for (int u = 0; u < PF; ++u) s.len[u] = (ka + u < a1) ? (uint32_t)(p1[u] - s.p0[u]) : 0u;
  if (same_bits) { // wave-uniform
    const auto sb = scalar_view(same_bits);
    uint32_t kprev = ka > a0 ? a_col[ka - 1] : 0xFFFFFFFEu; // (no entry before the first: never k - 1)
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const uint32_t w = sb[k[u] >> 5];
      if (k[u] == kprev + 1u && ((w >> (k[u] & 31u)) & 1u)) s.len[u] = 0u;
      kprev = k[u];
    }
  }
}

// TAIL: entries 64..127 of every B row are fetched here as well (c2, v2) instead of by the consumer.
template <bool WITH_VAL, bool WITH_AVAL, bool TAIL>
__device__ __forceinline__ void walk_vector_stage(const Csr &B, const WalkInfo<WITH_AVAL> &s, uint32_t lane,
                                                  uint32_t (&c)[PF], double (&v)[PF], uint32_t (&c2)[PF], double (&v2)[PF]) {
#pragma unroll
  for (int u = 0; u < PF; ++u) {
    const bool ok = lane < s.len[u];
    c[u] = ok ? (B.col + s.p0[u])[lane] : EMPTY_KEY;
    if constexpr (WITH_VAL) v[u] = ok ? (B.val + s.p0[u])[lane] : 0.0;
    if constexpr (TAIL) {
      const bool ok2 = lane + 64u < s.len[u];
      c2[u] = ok2 ? (B.col + s.p0[u])[lane + 64u] : EMPTY_KEY;
      if constexpr (WITH_VAL) v2[u] = ok2 ? (B.val + s.p0[u])[lane + 64u] : 0.0;
    }
  }
}

// begin() issues the first loads of the walk (two scalar stages and one vector stage) so that a
// kernel can overlap them with its LDS set-up; run() consumes the row.
// TAIL: entries 64..127 of a B row are fetched ahead with the first 64 (8 or 24 more registers); without
// it the consumer loads everything past the first 64 itself, with the latency exposed.  Pays where B's
// rows are mostly longer than a wavefront.
template <bool WITH_VAL, bool WITH_AVAL = WITH_VAL, bool TAIL = false> struct RowWalk {
  uint64_t a0, a1;
  WalkInfo<WITH_AVAL> s_cur, s_next;
  uint32_t c_cur[PF], c2_cur[PF];
  double v_cur[PF], v2_cur[PF];
  const uint32_t *same_bits = nullptr; // pattern-only walks: skip entries that repeat a column list (see walk_scalar_stage)

  __device__ __forceinline__ void begin(const Csr &A, const Csr &B, uint32_t row, uint32_t lane) {
    const auto a_ptr = scalar_view(A.ptr);
    a0 = a_ptr[row];
    a1 = a_ptr[row + 1];
    walk_scalar_stage<WITH_AVAL>(A, B, a0, a1, s_cur, same_bits, a0);
    walk_vector_stage<WITH_VAL, WITH_AVAL, TAIL>(B, s_cur, lane, c_cur, v_cur, c2_cur, v2_cur);
    walk_scalar_stage<WITH_AVAL>(A, B, a0 + PF, a1, s_next, same_bits, a0);
  }

  template <class F> __device__ __forceinline__ void run(const Csr &A, const Csr &B, uint32_t lane, F f) {
    WalkInfo<WITH_AVAL> s_after;
    uint32_t c_next[PF], c2_next[PF];
    double v_next[PF], v2_next[PF];
    for (uint64_t ka = a0; ka < a1; ka += PF) {
      walk_vector_stage<WITH_VAL, WITH_AVAL, TAIL>(B, s_next, lane, c_next, v_next, c2_next, v2_next);
      walk_scalar_stage<WITH_AVAL>(A, B, ka + 2 * PF, a1, s_after, same_bits, a0);
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        const uint32_t apos = (uint32_t)(ka - a0) + u;
        if (c_cur[u] != EMPTY_KEY) f(c_cur[u], WITH_VAL ? v_cur[u] : 0.0, WITH_AVAL ? s_cur.av[u] : 0.0, apos);
        if (s_cur.len[u] > 64) { // wave-uniform: the rest of a long B row, before the next A entry
          const uint32_t *pc = B.col + s_cur.p0[u];
          const double *pv = B.val + s_cur.p0[u];
          uint32_t j0 = 64;
          if constexpr (TAIL) {
            if (c2_cur[u] != EMPTY_KEY) f(c2_cur[u], WITH_VAL ? v2_cur[u] : 0.0, WITH_AVAL ? s_cur.av[u] : 0.0, apos);
            j0 = 128;
          }
          for (uint32_t j = j0 + lane; j < s_cur.len[u]; j += 64)
            f(pc[j], WITH_VAL ? pv[j] : 0.0, WITH_AVAL ? s_cur.av[u] : 0.0, apos);
        }
      }
      s_cur = s_next;
      s_next = s_after;
#pragma unroll
      for (int u = 0; u < PF; ++u) {
        c_cur[u] = c_next[u];
        v_cur[u] = v_next[u];
        if constexpr (TAIL) {
          c2_cur[u] = c2_next[u];
          v2_cur[u] = v2_next[u];
        }
      }
    }
  }
};

template <bool WITH_VAL, bool WITH_AVAL = WITH_VAL, bool TAIL = false, class F>
__device__ __forceinline__ void wave_row_walk(const Csr &A, const Csr &B, uint32_t row, uint32_t lane, F f,
                                              const uint32_t *same_bits = nullptr) {
  RowWalk<WITH_VAL, WITH_AVAL, TAIL> w;
  w.same_bits = same_bits;
  w.begin(A, B, row, lane);
  w.run(A, B, lane, f);
}

// ---- walking an A row run by run -----------------------------------------------------------------
// For A rows of up to 128 entries (the row groups of the window path).  A run = up to RMAX
// consecutive entries of the A row that select CONSECUTIVE rows k, k + 1, ... of B which all have
// one column list (RowInfo::same).  Such rows lie back to back in B's arrays, so entry j of row
// k + r sits at p0 + r * L + j (p0: start of row k, L: the common length).  A unit of the walk is one
// chunk of 64 columns of one run: lane j holds column p0[chunk * 64 + j] once and the value of that
// column in each of the run's rows.  The consumer finds the column's output position once per unit
// and adds the run's products to it row after row - the order of the CPU loop, since the entries of
// a run are consecutive in the A row and the lanes of a unit hold distinct columns.
//
// The description of the whole A row lives in registers: lane t holds, for the entries t and 64 + t,
// the start and the length of the B row and whether the entry continues a run; it is loaded once,
// when the kernel starts.  No load of the walk depends on another load after that, so the B
// entries of DEPTH units are in flight while one is consumed (the kernels that walk a row through
// dependent scalar loads have one stage in flight and are bound by the latency of that chain).
//
// The units are laid out by the vector unit, 64 at a time, lane u holding unit u (its rows' start,
// length, first entry, row count and chunk): a prefix sum of the chunk counts of the entries that
// begin a unit's rows, and a search of every lane for its entry over those sums.  Handing out a unit
// is then four lane reads by a counter.  (Enumerated on the scalar unit - next entry that begins a
// run, its length, chunk counter, two units of state in flight - the walk cost about a hundred scalar
// instructions per unit, twice what the vector unit spent on the unit's products: these kernels
// are bound by the instructions they issue, the scalar ones included.)
struct RunInfo { // wave-uniform
  uint64_t p0;  // start of the first B row of the run
  uint32_t L;   // entries of each of its B rows
  uint32_t rl;  // B rows (= A entries) in the run, 1..RMAX
  uint32_t t0;  // position of its first entry in the A row
  uint32_t chunk;
  bool valid;
};
// PAIR: a unit is 128 columns of its rows, two per lane (col2 / ok2 / bv2: the column 64 further on): half
// as many units for rows of B with 65 to 128 entries, whose second 64 columns fill a quarter of the lanes
template <int NV, bool PAIR = false> struct RunVec {
  uint32_t col;     // the lane's column of this unit
  bool ok;          // ... if it holds one
  double bv[NV];    // the column's value in the run's rows (rows past the run's end repeat the last one); pattern walks: unused
  uint32_t col2;
  bool ok2;
  double bv2[NV];
};
// TABLE: the units are laid out by the vector unit (below); else they are enumerated on the scalar unit,
// one at a time (the kernels of the narrow groups: their walk is short, and the tables' registers
// cost them a fifth of their wavefronts per CU)
template <bool WIDE, bool TABLE = true, bool PAIR = false> struct RowRuns {
  static_assert(TABLE || !PAIR, "units of 128 columns come with the unit tables");
  static constexpr bool table = TABLE;
  // per lane: entry t (lo) and 64 + t (hi, WIDE only)
  uint64_t p0_lo, p0_hi;
  uint32_t len_lo, len_hi;
  uint64_t start_lo, start_hi; // wave-uniform: entries that begin a run
  uint32_t len_a;
  // rs_lo / rs_hi (wave-uniform): the entries that begin a unit's rows: every entry that begins a
  // run and, in a run of more than RMAX rows, every RMAX-th entry after it
  uint64_t rs_lo, rs_hi;
  // the units: of one half of the entries (ph = 0: entries 0..63, 1: 64..127) at a time, 64 per table
  uint32_t ph, ph_tot, ph_done; // wave-uniform: half, its units, those of them laid out so far
  uint64_t u_p0;                // per lane (unit of the table): start of the first of its B rows
  uint32_t u_L, u_meta;         // entries per row; first entry | rows << 8 | chunk << 16
  uint32_t nu, ui;              // wave-uniform: units in the table, next one to hand out
  // !TABLE: the iterator on the scalar unit
  uint32_t t, run_end, rl, L, chunk;
  uint64_t p0;
  bool done;

  // The description is loaded in two dependent steps, which a kernel that walks several rows one
  // after the other issues a row (and two rows) ahead: issue_a() loads the row's column indices k,
  // issue_b() the bounds of the B rows k and their `same` bits, finish() (once those have landed)
  // finds the runs.  Every lane issues every load (a lane past the row's end re-reads the row's
  // last entry): the count of loads in flight does not depend on the row.  len >= 1.
  uint32_t k_lo, k_hi;
  uint32_t bits_lo, bits_hi; // the words of same_bits that hold the lane's rows
  uint64_t p1_lo, p1_hi;     // ends of the lane's B rows
  __device__ __forceinline__ void issue_a(const Csr &A, uint64_t a0, uint32_t len, uint32_t lane) {
    len_a = len;
    k_lo = A.col[a0 + min(lane, len - 1u)];
    k_hi = 0;
    if constexpr (WIDE) k_hi = A.col[a0 + min(64u + lane, len - 1u)];
  }
  __device__ __forceinline__ void issue_b(const Csr &B, const uint32_t *__restrict__ same_bits) {
    p0_lo = B.ptr[k_lo];
    p1_lo = B.ptr[(uint64_t)k_lo + 1u];
    bits_lo = same_bits[k_lo >> 5];
    p0_hi = p1_hi = 0; bits_hi = 0;
    if constexpr (WIDE) {
      p0_hi = B.ptr[k_hi];
      p1_hi = B.ptr[(uint64_t)k_hi + 1u];
      bits_hi = same_bits[k_hi >> 5];
    }
  }
  // RMAX: rows a unit takes at most (128 or more: a run is never cut)
  template <int RMAX> __device__ __forceinline__ void finish(uint32_t lane) {
    const uint32_t len = len_a;
    len_lo = lane < len ? (uint32_t)min(p1_lo - p0_lo, (uint64_t)0xFFFFFFFFu) : 0u;
    // an entry continues a run when it selects the row after the one its predecessor selects and
    // that row repeats the predecessor's column list
    uint32_t kp_lo = __shfl_up(k_lo, 1, 64);
    if (lane == 0) kp_lo = k_lo; // (the first entry never continues one)
    const bool cont_lo = lane < len && ((bits_lo >> (k_lo & 31u)) & 1u) && k_lo == kp_lo + 1u;
    start_lo = __ballot(lane < len && !cont_lo);
    start_hi = 0;
    len_hi = 0;
    if constexpr (WIDE) {
      len_hi = 64u + lane < len ? (uint32_t)min(p1_hi - p0_hi, (uint64_t)0xFFFFFFFFu) : 0u;
      uint32_t kp_hi = __shfl_up(k_hi, 1, 64);
      const uint32_t k63 = (uint32_t)__builtin_amdgcn_readlane((int)k_lo, 63);
      if (lane == 0) kp_hi = k63;
      const bool cont_hi = 64u + lane < len && ((bits_hi >> (k_hi & 31u)) & 1u) && k_hi == kp_hi + 1u;
      start_hi = __ballot(64u + lane < len && !cont_hi);
    }
    if constexpr (!TABLE) {
      t = 0; run_end = 0; rl = 0; L = 0; chunk = 0; p0 = 0;
      done = len == 0;
    }
    rs_lo = start_lo;
    rs_hi = start_hi;
    if constexpr (RMAX < 128) {
      // where the lane's run began (entry 0 begins one, so the masks below are never empty for an entry of the row)
      const uint64_t le = (2ull << lane) - 1ull;
      const uint32_t s_lo = 63u - (uint32_t)__clzll((long long)((start_lo & le) | 1ull));
      rs_lo = __ballot(lane < len && (lane - s_lo) % (uint32_t)RMAX == 0u);
      if constexpr (WIDE) {
        const uint64_t mh = start_hi & le;
        const uint32_t s_hi = mh ? 127u - (uint32_t)__clzll((long long)mh) : 63u - (uint32_t)__clzll((long long)(start_lo | 1ull));
        rs_hi = __ballot(64u + lane < len && (64u + lane - s_hi) % (uint32_t)RMAX == 0u);
      }
    }
    ph = 0; ph_tot = 0; ph_done = 0; nu = 0; ui = 0;
    if constexpr (TABLE)
      if (len) refill(lane);
  }
  // prefix sums of the chunk counts of half ph's entries that begin a unit's rows (base: units before the
  // entry's; rows: rows of its unit).  Recomputed by every refill: a second one is rare (more than 64
  // units in a half), and the two registers would otherwise be held through the whole walk.
  __device__ __forceinline__ void half_sums(uint32_t lane, uint32_t &base, uint32_t &rows) {
    const bool hi = WIDE && ph != 0u; // wave-uniform
    const uint64_t rs = hi ? rs_hi : rs_lo;
    const uint32_t off = hi ? 64u : 0u;
    // the entry after the unit's rows: the next one that begins rows, in this half, the other one, or the row's end
    const uint32_t tail = ((!hi && WIDE && rs_hi) ? 64u + (uint32_t)__builtin_ctzll(rs_hi) : len_a) - off;
    const uint64_t after = lane < 63u ? rs >> (lane + 1u) : 0ull;
    const uint32_t tn = after ? lane + 1u + (uint32_t)__builtin_ctzll(after) : tail;
    rows = tn - lane;
    const uint32_t len = hi ? len_hi : len_lo;
    const uint32_t nch = ((rs >> lane) & 1ull) ? (PAIR ? (len + 127u) >> 7 : (len + 63u) >> 6) : 0u;
    const uint32_t incl = group_incl_scan<64>(nch);
    base = incl - nch;
    ph_tot = group_last<64>(incl);
  }
  // the next table of up to 64 units; nu == 0: the row has no more
  __device__ __forceinline__ void refill(uint32_t lane) {
    ui = 0;
    nu = 0;
    uint32_t ph_base, ph_rl;
    half_sums(lane, ph_base, ph_rl);
    if (ph_done >= ph_tot) { // (wave-uniform) this half is used up
      if (!WIDE || ph != 0u || rs_hi == 0ull) return;
      ph = 1u;
      ph_done = 0;
      half_sums(lane, ph_base, ph_rl);
      if (ph_tot == 0u) return;
    }
    const bool hi = WIDE && ph != 0u;
    const uint32_t u = ph_done + lane;
    // the unit's entry: the last one with at most u units before it (the sums do not decrease, and an
    // entry that begins no rows, or rows without entries, has the sum of the entry after it)
    uint32_t t = 0;
#pragma unroll
    for (uint32_t s = 32; s > 0; s >>= 1) {
      const uint32_t b = __shfl(ph_base, (int)(t + s), 64);
      if (b <= u) t += s;
    }
    const uint32_t chunk = u - __shfl(ph_base, (int)t, 64);
    const uint32_t rows = __shfl(ph_rl, (int)t, 64);
    if (hi) {
      u_p0 = __shfl(p0_hi, (int)t, 64);
      u_L = __shfl(len_hi, (int)t, 64);
    } else {
      u_p0 = __shfl(p0_lo, (int)t, 64);
      u_L = __shfl(len_lo, (int)t, 64);
    }
    u_meta = (t + (hi ? 64u : 0u)) | (rows << 8) | (chunk << 16);
    nu = min(64u, ph_tot - ph_done);
    ph_done += nu;
  }
  // the whole description at once, for A row [a0, a0 + len); len <= 64 (128 when WIDE); len == 0: no unit
  template <int RMAX>
  __device__ __forceinline__ void load(const Csr &A, const Csr &B, const uint32_t *__restrict__ same_bits, uint64_t a0,
                                       uint32_t len, uint32_t lane) {
    if (len == 0) { // wave-uniform
      len_a = 0; start_lo = start_hi = 0; rs_lo = rs_hi = 0; ph = 0; ph_tot = 0; ph_done = 0; nu = 0; ui = 0;
      done = true; rl = 0; L = 0; chunk = 0; t = 0; run_end = 0; p0 = 0;
      p0_lo = p0_hi = 0; len_lo = len_hi = 0;
      return;
    }
    issue_a(A, a0, len, lane);
    issue_b(B, same_bits);
    finish<RMAX>(lane);
  }
  // more units after those of the table
  __device__ __forceinline__ bool more() const { return TABLE && (ph_done < ph_tot || (WIDE && ph == 0u && rs_hi != 0ull)); }
  // !TABLE: first entry >= from that begins a run (len_a when there is none)
  __device__ __forceinline__ uint32_t next_start(uint32_t from) const {
    if (from < 64u) {
      const uint64_t m = start_lo >> from;
      if (m) return from + (uint32_t)__builtin_ctzll(m);
      from = 64u;
    }
    if constexpr (WIDE) {
      if (from < 128u) {
        const uint64_t m = start_hi >> (from - 64u);
        if (m) return from + (uint32_t)__builtin_ctzll(m);
      }
    }
    return len_a;
  }
  // the next unit (RMAX: as given to finish()).  TABLE: of the table; the caller refills an empty one
  template <int RMAX> __device__ __forceinline__ RunInfo next() {
    RunInfo r{0, 0, 1, 0, 0, false};
    if constexpr (!TABLE && !WIDE) {
      // rows of up to 64 entries: the entries that begin a unit's rows are the bits of rs_lo; the next
      // unit's entry is the lowest one, its rows end at the one after it
      if (done) return r;
      if (rl != 0 && (chunk + 1u) * 64u < L) { // more columns of the same rows
        ++chunk;
      } else {
        if (rs_lo == 0ull) { done = true; return r; }
        const uint32_t tt = (uint32_t)__builtin_ctzll(rs_lo);
        rs_lo &= rs_lo - 1ull;
        t = tt;
        rl = (rs_lo ? (uint32_t)__builtin_ctzll(rs_lo) : len_a) - tt;
        chunk = 0;
        p0 = readlane64(p0_lo, tt);
        L = (uint32_t)__builtin_amdgcn_readlane((int)len_lo, (int)tt);
      }
      r.p0 = p0; r.L = L; r.rl = rl; r.t0 = t; r.chunk = chunk; r.valid = true;
      return r;
    }
    if constexpr (!TABLE) {
      if (done) return r;
      if (rl != 0 && (chunk + 1u) * 64u < L) { // more columns of the same rows
        ++chunk;
      } else {
        t += rl; // (0 before the first unit)
        if (t >= len_a) { done = true; return r; }
        if (t >= run_end) run_end = next_start(t + 1u);
        rl = min(run_end - t, (uint32_t)RMAX);
        chunk = 0;
        if constexpr (WIDE) {
          if (t < 64u) { p0 = readlane64(p0_lo, t); L = (uint32_t)__builtin_amdgcn_readlane((int)len_lo, (int)t); }
          else { p0 = readlane64(p0_hi, t - 64u); L = (uint32_t)__builtin_amdgcn_readlane((int)len_hi, (int)(t - 64u)); }
        } else {
          p0 = readlane64(p0_lo, t);
          L = (uint32_t)__builtin_amdgcn_readlane((int)len_lo, (int)t);
        }
      }
      r.p0 = p0; r.L = L; r.rl = rl; r.t0 = t; r.chunk = chunk; r.valid = true;
      return r;
    }
    if (ui >= nu) return r; // wave-uniform
    const uint32_t i = ui++;
    const uint32_t meta = (uint32_t)__builtin_amdgcn_readlane((int)u_meta, (int)i);
    r.p0 = readlane64(u_p0, i);
    r.L = (uint32_t)__builtin_amdgcn_readlane((int)u_L, (int)i);
    r.t0 = meta & 0xFFu;
    r.rl = (meta >> 8) & 0xFFu;
    r.chunk = meta >> 16;
    r.valid = true;
    return r;
  }
};
// Issues the loads of a unit.  Every lane issues every load (a lane past the end of the B rows,
// a row past the end of the run and an exhausted walk read an address that is always valid: the
// run's first entry, or entry 0 of B): the number of loads in flight is then the same on every path
// through the loop, which lets the compiler wait for exactly the unit it is about to consume.
template <int RMAX, bool WITH_VAL, bool PAIR = false>
__device__ __forceinline__ void run_fetch(const Csr &B, const RunInfo &s, uint32_t lane, RunVec<WITH_VAL ? RMAX : 1, PAIR> &v) {
  const uint32_t j = s.chunk * (PAIR ? 128u : 64u) + lane;
  v.ok = s.valid && j < s.L;
  const uint64_t base = (s.valid && s.L) ? s.p0 : 0ull;
  // (the rows of B walked here are strictly increasing rows of a window row: at most 65,536 entries.
  //  The mask tells the compiler that the byte offset fits 32 bits: scalar base + 32-bit lane offset.)
  const uint32_t jj = (v.ok ? j : 0u) & 0xFFFFFu;
  v.col = (B.col + base)[jj];
  uint32_t jj2 = 0;
  v.ok2 = false;
  v.col2 = 0;
  if constexpr (PAIR) {
    v.ok2 = s.valid && j + 64u < s.L;
    jj2 = (v.ok2 ? j + 64u : 0u) & 0xFFFFFu;
    v.col2 = (B.col + base)[jj2];
  }
  if constexpr (WITH_VAL) {
    // (the lane reads say so to the compiler: scalar base + 32-bit lane offset, no 64-bit address per lane and row)
    const uint64_t vbp = (uint64_t)(uintptr_t)(B.val + base);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)vbp);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(vbp >> 32));
    typedef const double __attribute__((address_space(1))) *global_f64_ptr; // (global memory, not a generic address)
    const uint64_t vb = ((uint64_t)hi << 32) | lo;
#pragma unroll
    for (int r = 0; r < RMAX; ++r) { // a row of the run: its own base on the scalar unit, the lanes' offset is the same for all rows
      const uint32_t rr = min((uint32_t)r, s.rl - 1u);
      v.bv[r] = ((global_f64_ptr)(uintptr_t)(vb + (uint64_t)(rr * s.L) * 8u))[jj];
      if constexpr (PAIR) v.bv2[r] = ((global_f64_ptr)(uintptr_t)(vb + (uint64_t)(rr * s.L) * 8u))[jj2];
    }
  }
}

// The units of a row, SLOTS of them in flight: begin() issues the first SLOTS fetches (to be
// overlapped with the kernel's set-up), run() consumes a unit and refills its slot.
// f(info, vec) is called once per unit by all lanes.
template <int RMAX, bool WITH_VAL, int SLOTS, bool WIDE, bool PAIR = false> struct RunPipe {
  RunInfo info[SLOTS];
  RunVec<WITH_VAL ? RMAX : 1, PAIR> vec[SLOTS];
  template <class RR> __device__ __forceinline__ void begin(const Csr &B, RR &rr, uint32_t lane) {
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      info[s] = rr.template next<RMAX>();
      run_fetch<RMAX, WITH_VAL, PAIR>(B, info[s], lane, vec[s]);
    }
  }
  template <class RR, class F> __device__ __forceinline__ void run(const Csr &B, RR &rr, uint32_t lane, F f) {
    if constexpr (!RR::table) { // units enumerated one by one: no table to drain at
      for (;;) {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
          if (!info[s].valid) return; // the units come in order: the first empty slot ends the walk
          f(info[s], vec[s]);
          info[s] = rr.template next<RMAX>();
          run_fetch<RMAX, WITH_VAL, PAIR>(B, info[s], lane, vec[s]);
        }
      }
    }
    for (;;) {
      // the units of one table; the pipeline drains at its end (once per 64 units)
      for (bool open = true; open;) {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
          if (!info[s].valid) { open = false; break; } // the units come in order: the first empty slot ends the table
          f(info[s], vec[s]);
          info[s] = rr.template next<RMAX>();
          run_fetch<RMAX, WITH_VAL, PAIR>(B, info[s], lane, vec[s]);
        }
      }
      if (!rr.more()) return;
      rr.refill(lane);
      if (rr.nu == 0u) return;
      begin(B, rr, lane);
    }
  }
};

// ---- flat walk ---------------------------------------------------------------------------------
// For rows whose B rows are short, one A entry per step leaves most lanes idle.  The flat walk
// lays the products of up to 64 A entries end to end (A order, then B row order) and hands 64
// consecutive products to the 64 lanes per step, whatever B rows they come from: lane l of a step
// holds product q = q0 + l and finds its A entry by a binary search over the exclusive prefix
// sums of the B row lengths (kept in LDS).  Products of one step are in CPU order by lane, and
// steps are sequential, so a consumer that resolves two lanes hitting one accumulator in lane
// order (claim_in_lane_order) still adds in the CPU order.
// LDS scratch per wavefront: st[64] u32 | p0s[64] u64 | avs[64] f64 = FLAT_BYTES.
// f(col, bval, aval, apos) is called by ALL lanes every step; col == EMPTY_KEY marks an idle lane.
constexpr uint32_t FLAT_BYTES = 1280;
#ifndef SG_FLAT_PF
#define SG_FLAT_PF 3
#endif
constexpr int FLAT_PF = SG_FLAT_PF; // steps of the flat walk whose loads are in flight ahead of the consumed one

template <bool WITH_VAL, class F>
__device__ __forceinline__ void flat_walk(const Csr &A, const Csr &B, uint32_t row, uint32_t lane,
                                          unsigned char *scratch, F f) {
  uint32_t *st = reinterpret_cast<uint32_t *>(scratch);
  uint64_t *p0s = reinterpret_cast<uint64_t *>(scratch + 256);
  double *avs = reinterpret_cast<double *>(scratch + 768);
  const uint64_t a0 = scalar_view(A.ptr)[row], a1 = scalar_view(A.ptr)[row + 1];
  for (uint64_t ab = a0; ab < a1; ab += 64) {
    const uint64_t ka = ab + lane;
    uint32_t len = 0;
    uint64_t p0 = 0;
    double av = 0.0;
    if (ka < a1) {
      const uint32_t k = A.col[ka];
      p0 = B.ptr[k];
      len = (uint32_t)(B.ptr[k + 1] - p0);
      if constexpr (WITH_VAL) av = A.val[ka];
    }
    const uint32_t incl = group_incl_scan<64>(len);
    const uint32_t total = group_last<64>(incl);
    lds_fence(); // the previous chunk has been consumed
    st[lane] = incl - len;
    p0s[lane] = p0;
    if constexpr (WITH_VAL) avs[lane] = av;
    lds_fence();
    auto fetch = [&](uint32_t q, uint32_t &t, uint32_t &c, double &bv) {
      const bool valid = q < total;
      const uint32_t qq = valid ? q : 0u;
      t = 0;
#pragma unroll
      for (uint32_t s = 32; s > 0; s >>= 1)
        if (st[t + s] <= qq) t += s; // largest t with st[t] <= q: the entry that holds product q
      const uint64_t p = p0s[t] + (qq - st[t]);
      c = valid ? B.col[p] : EMPTY_KEY;
      if constexpr (WITH_VAL) bv = valid ? B.val[p] : 0.0;
    };
    // FLAT_PF steps are in flight ahead of the one that is consumed
    uint32_t t_q[FLAT_PF], c_q[FLAT_PF];
    double bv_q[FLAT_PF];
#pragma unroll
    for (int d = 0; d < FLAT_PF; ++d) {
      t_q[d] = 0; c_q[d] = EMPTY_KEY; bv_q[d] = 0.0;
      if (64u * d < total) fetch(64u * d + lane, t_q[d], c_q[d], bv_q[d]);
    }
    for (uint32_t q0 = 0; q0 < total; q0 += 64) {
      const uint32_t t_cur = t_q[0], c_cur = c_q[0];
      const double bv_cur = bv_q[0];
#pragma unroll
      for (int d = 0; d + 1 < FLAT_PF; ++d) { t_q[d] = t_q[d + 1]; c_q[d] = c_q[d + 1]; bv_q[d] = bv_q[d + 1]; }
      t_q[FLAT_PF - 1] = 0; c_q[FLAT_PF - 1] = EMPTY_KEY; bv_q[FLAT_PF - 1] = 0.0;
      if (q0 + 64u * FLAT_PF < total) fetch(q0 + 64u * FLAT_PF + lane, t_q[FLAT_PF - 1], c_q[FLAT_PF - 1], bv_q[FLAT_PF - 1]);
      double av_cur = 0.0;
      if constexpr (WITH_VAL) av_cur = avs[t_cur];
      f(c_cur, bv_cur, av_cur, (uint32_t)(ab - a0) + t_cur);
    }
  }
}

// Add `prod` to vals[p] for every lane with pending == true so that lanes of this call that share
// one p add in lane order.  owner[] (one uint32 per accumulator, zero at kernel start) takes
// claims (stamp << 6 | 63 - lane) by atomic max: the lowest pending lane of the newest round wins
// its accumulator, adds, and the others try again in the next round.  stamp is wave-uniform.
// Callers zero owner[] and the stamp again once the stamp passes CLAIM_STAMP_LIMIT: a tag is
// stamp << 6 | lane, and a wrapped stamp could never win an atomic max again.
//
// hw_order (wave-uniform): the device applies the lanes of ONE ds_add_f64 instruction that share an
// address in ascending lane order (k_probe_lane_order checked that when the context was made): the
// whole call is then that single instruction, and owner[] and the stamp are not touched.
constexpr uint32_t CLAIM_STAMP_LIMIT = 1u << 25;
__device__ __forceinline__ void claim_in_lane_order(double *vals, uint32_t *owner, uint32_t &stamp, uint32_t lane,
                                                    bool pending, uint32_t p, double prod, bool hw_order) {
  if (hw_order) {
    if (pending) lds_add_f64(&vals[p], prod);
    return;
  }
  for (;;) {
    ++stamp;
    const uint32_t tag = (stamp << 6) | (63u - lane);
    if (pending) atomicMax(&owner[p], tag);
    lds_fence();
    if (pending && owner[p] == tag) {
      lds_add_f64(&vals[p], prod);
      pending = false;
    }
    if (!__any(pending)) break;
    lds_fence();
  }
}

// r-th (0-based) set bit of a 32-bit word that has more than r bits set
__device__ __forceinline__ uint32_t select_bit(uint32_t bits, uint32_t r) {
  uint32_t base = 0, c;
  c = __popc(bits & 0xFFFFu); if (r >= c) { r -= c; bits >>= 16; base += 16; }
  c = __popc(bits & 0xFFu);   if (r >= c) { r -= c; bits >>= 8;  base += 8; }
  c = __popc(bits & 0xFu);    if (r >= c) { r -= c; bits >>= 4;  base += 4; }
  c = __popc(bits & 0x3u);    if (r >= c) { r -= c; bits >>= 2;  base += 2; }
  c = bits & 1u;              if (r >= c) { base += 1; }
  return base;
}

} // namespace sg
