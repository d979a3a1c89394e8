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

template <bool WITH_AVAL>
__device__ __forceinline__ void walk_scalar_stage(const Csr &A, const Csr &B, uint64_t ka, uint64_t a1,
                                                  WalkInfo<WITH_AVAL> &s) {
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

  __device__ __forceinline__ void begin(const Csr &A, const Csr &B, uint32_t row, uint32_t lane) {
    const auto a_ptr = scalar_view(A.ptr);
    a0 = a_ptr[row];
    a1 = a_ptr[row + 1];
    walk_scalar_stage<WITH_AVAL>(A, B, a0, a1, s_cur);
    walk_vector_stage<WITH_VAL, WITH_AVAL, TAIL>(B, s_cur, lane, c_cur, v_cur, c2_cur, v2_cur);
    walk_scalar_stage<WITH_AVAL>(A, B, a0 + PF, a1, s_next);
  }

  template <class F> __device__ __forceinline__ void run(const Csr &A, const Csr &B, uint32_t lane, F f) {
    WalkInfo<WITH_AVAL> s_after;
    uint32_t c_next[PF], c2_next[PF];
    double v_next[PF], v2_next[PF];
    for (uint64_t ka = a0; ka < a1; ka += PF) {
      walk_vector_stage<WITH_VAL, WITH_AVAL, TAIL>(B, s_next, lane, c_next, v_next, c2_next, v2_next);
      walk_scalar_stage<WITH_AVAL>(A, B, ka + 2 * PF, a1, s_after);
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
__device__ __forceinline__ void wave_row_walk(const Csr &A, const Csr &B, uint32_t row, uint32_t lane, F f) {
  RowWalk<WITH_VAL, WITH_AVAL, TAIL> w;
  w.begin(A, B, row, lane);
  w.run(A, B, lane, f);
}

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
