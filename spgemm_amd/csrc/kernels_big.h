// Big rows of a sorted B: symbolic with LDS bitmap windows per workgroup, numeric in output slices.
// Part of the gfx950 SpGEMM kernels; included through kernels.h.
#pragma once
#include "kernels_numeric.h"

namespace sg {

// ---- symbolic, big rows: one workgroup of BIG_THREADS per row, bitmap windows in LDS --------------
// Needs B rows with strictly increasing columns.  The row's column hull is cut into windows of
// win_words * 32 columns.  Every A entry keeps {position, end} of what is left of its B row (in
// LDS for rows of up to BIG_CUR entries, else in g_state, two uint64 per entry of A); in a window
// the part of the B row that falls into it (a "piece": contiguous, the B row is sorted) is ORed
// into the LDS bitmap and the position moves on.  One wavefront takes the first BIG_MID entries of
// a piece, four pieces at a time, all their loads in flight together; a piece whose first BIG_MID
// entries all lay inside the window goes on: all wavefronts take BIG_THREADS of its entries per
// step, four such pieces at a time, and a wavefront stops at its first block that leaves the window
// (handing every B row with more than BIG_MID entries left to all wavefronts at once cost 10 %
// more on R-MAT 22: most of those pieces have a few dozen entries in a window).
// The bitmap is then written out in blocks of 256 words (four per lane, the blocks dealt to the
// wavefronts round robin): block totals, a scan over the blocks, and every lane writes the
// columns of its own four words, at most 128.
// LDS: win_words u32 bitmap | BIG_CUR u64 pos | BIG_CUR u64 end | BIG_CUR u32 list of mid pieces |
//      BIG_CUR u32 list of long pieces | BIG_CUR u32 entries taken from long pieces | win_words / 64 u32 (block offsets; a quarter used).
// 512 threads and windows of up to 16384 words: two workgroups fit the 160 KiB of a CU, and one works
// while the other waits at a barrier (1024 threads with twice the window: 10 % slower on R-MAT 22;
// 256 threads with half of it: 15 % slower, the rows are passed over once per window).
#ifndef SG_BIG_THREADS
#define SG_BIG_THREADS 512
#endif
#ifndef SG_BIG_MID
#define SG_BIG_MID 256
#endif
constexpr uint32_t BIG_THREADS = SG_BIG_THREADS, BIG_WAVES = BIG_THREADS / 64, BIG_CUR = BIG_THREADS, BIG_MID = SG_BIG_MID;
constexpr int BIG_MID_LOADS = BIG_MID / 64; // loads per lane that cover the first BIG_MID entries of a piece
static_assert(BIG_CUR == BIG_THREADS, "a batch of A entries is one entry per thread, and the piece lists hold one batch");
constexpr uint32_t BIG_WIN_MAX = 32 * BIG_THREADS;        // words of the largest window
constexpr uint32_t BIG_BPW = BIG_WIN_MAX / 256 / BIG_WAVES; // blocks of 256 words per wavefront and window
__host__ __device__ inline size_t sym_big_lds(uint32_t win_words) {
  return (size_t)win_words * 4 + (size_t)BIG_CUR * (8 + 8 + 4 + 4 + 4) + (size_t)(win_words / 64) * 4;
}

// two workgroups per CU: 2 * BIG_WAVES / 4 wavefronts per SIMD have to fit the register file
__global__ void __launch_bounds__(BIG_THREADS) __attribute__((amdgpu_waves_per_eu(2 * BIG_WAVES / 4, 2 * BIG_WAVES / 4)))
    k_sym_big(const uint32_t *__restrict__ rows, uint32_t nrows, Csr A, Csr B, RowMeta m, uint32_t win_words,
              uint64_t *__restrict__ g_state, uint64_t nnz_a, unsigned long long *__restrict__ stamps) {
#ifdef SG_SYM_STAMPS
  // diagnostic build: cycles per phase, summed over the wavefronts into stamps[0..7]
  unsigned long long tph[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = __builtin_amdgcn_s_memtime();
#define SY_STAMP(i) { const unsigned long long tn = __builtin_amdgcn_s_memtime(); tph[i] += tn - tlast; tlast = tn; }
#else
#define SY_STAMP(i)
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t *bits = reinterpret_cast<uint32_t *>(smem);
  uint64_t *s_pos = reinterpret_cast<uint64_t *>(smem + (size_t)win_words * 4);
  uint64_t *s_end = s_pos + BIG_CUR;
  uint32_t *s_mid = reinterpret_cast<uint32_t *>(s_end + BIG_CUR);
  uint32_t *s_long = s_mid + BIG_CUR;
  uint32_t *s_cnt = s_long + BIG_CUR;
  uint32_t *s_blk = s_cnt + BIG_CUR; // per block of 64 words: set bits, then offset of its first column
  __shared__ uint32_t s_n[2]; // mid pieces, long pieces of the current batch
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const uint32_t row = scalar_view(rows)[blockIdx.x];
  const uint64_t a0 = scalar_view(A.ptr)[row], a1 = scalar_view(A.ptr)[row + 1];
  const uint32_t len_a = (uint32_t)(a1 - a0);
  const uint32_t w0 = m.cmin[row] >> 5, w1 = m.cmax[row] >> 5;
  uint64_t *pos = len_a <= BIG_CUR ? s_pos : g_state + a0;
  uint64_t *end = len_a <= BIG_CUR ? s_end : g_state + nnz_a + a0;
  uint32_t *out = m.temp_cols + m.t_off[row];
  uint32_t carry = 0;
  if (tid < 2) s_n[tid] = 0u;
  __syncthreads();
  for (uint32_t wb = w0; wb <= w1; wb += win_words) {
    const uint32_t nw = min(win_words, w1 + 1u - wb);
    const uint32_t nblk = (nw + 255u) >> 8;
    const bool first = wb == w0, last = wb + nw > w1;
    const uint64_t chi = (uint64_t)(wb + nw) << 5;
    if (first) { // later windows find the bitmap cleared by the write-out of the one before
      for (uint32_t w = tid * 4u; w < win_words; w += BIG_THREADS * 4u)
        *reinterpret_cast<uint4 *>(&bits[w]) = make_uint4(0u, 0u, 0u, 0u);
    }
    for (uint32_t kb = 0; kb < len_a; kb += BIG_CUR) {
      // classify what is left of every B row of this batch of A entries
      const uint32_t idx = kb + tid;
      if (idx < len_a) {
        uint64_t p, e;
        if (first) {
          const uint32_t k = A.col[a0 + idx];
          p = B.ptr[k];
          e = B.ptr[k + 1];
          pos[idx] = p;
          end[idx] = e;
        } else {
          p = pos[idx];
          e = end[idx];
        }
        if (e - p) s_mid[atomicAdd(&s_n[0], 1u)] = idx;
      }
      SY_STAMP(0)
      __syncthreads();
      SY_STAMP(1)
      const uint32_t nmid = s_n[0];
      // first BIG_MID entries of every piece: a wavefront takes four pieces and all of their (up to
      // 4 x BIG_MID) entries at once.  (Fetching the next four pieces while these are ORed in needs
      // more than the 128 registers two workgroups per CU leave, and runs at half the speed.)
      for (uint32_t e0 = wv * 4u; e0 < nmid; e0 += BIG_WAVES * 4u) {
        uint32_t id[4], c[4][BIG_MID_LOADS];
        uint64_t p[4], full[4];
        uint32_t rem[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool on = e0 + u < nmid;
          id[u] = s_mid[on ? e0 + u : e0];
          p[u] = pos[id[u]];
          full[u] = on ? end[id[u]] - p[u] : 0ull;
          rem[u] = (uint32_t)min(full[u], (uint64_t)BIG_MID);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int j = 0; j < BIG_MID_LOADS; ++j) c[u][j] = (64u * j + lane < rem[u]) ? B.col[p[u] + 64u * j + lane] : EMPTY_KEY;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          uint32_t took = 0;
#pragma unroll
          for (int j = 0; j < BIG_MID_LOADS; ++j) {
            const bool in = 64u * j + lane < rem[u] && (last || (uint64_t)c[u][j] < chi);
            if (in) atomicOr(&bits[(c[u][j] >> 5) - wb], 1u << (c[u][j] & 31u));
            took += (uint32_t)__popcll(__ballot(in));
          }
          if (lane == 0 && rem[u]) {
            pos[id[u]] = p[u] + took;
            if (took == BIG_MID && full[u] > BIG_MID) { // all of it lay in the window and the B row goes on
              const uint32_t slot = atomicAdd(&s_n[1], 1u);
              s_long[slot] = id[u];
              s_cnt[slot] = 0u;
            }
          }
        }
      }
      __syncthreads(); // the list of pieces that go on, and their positions
      const uint32_t nlong = s_n[1];
      SY_STAMP(2)
      // long pieces: all wavefronts, 1024 entries of a piece per step, four pieces at a time
      for (uint32_t e0 = 0; e0 < nlong; e0 += 4u) {
        uint32_t c[4];
        uint64_t p[4], rem[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool on = e0 + u < nlong;
          const uint32_t id = s_long[on ? e0 + u : e0];
          p[u] = pos[id];
          rem[u] = on ? end[id] - p[u] : 0ull;
        }
#ifdef SG_SYM_STAMPS
        if (p[0] + rem[1] + p[2] + rem[3] == 0x123456789ull) carry++; // the loads of p and rem have to land
        { const unsigned long long tn = __builtin_amdgcn_s_memtime(); tph[8] += tn - tlast; tlast = tn; }
        tph[10] += 1;
#endif
#pragma unroll
        for (int u = 0; u < 4; ++u) c[u] = tid < rem[u] ? B.col[p[u] + tid] : EMPTY_KEY;
#ifdef SG_SYM_STAMPS
        if (c[0] + c[1] + c[2] + c[3] == 0x12345678u) carry++; // the loads have to land
        { const unsigned long long tn = __builtin_amdgcn_s_memtime(); tph[9] += tn - tlast; tlast = tn; }
#endif
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool in = tid < rem[u] && (last || (uint64_t)c[u] < chi);
          if (in) atomicOr(&bits[(c[u] >> 5) - wb], 1u << (c[u] & 31u));
          uint32_t n = (uint32_t)__popcll(__ballot(in));
          uint32_t took = n;
          for (uint64_t base = BIG_THREADS; n == 64u && base + wv * 64u < rem[u]; base += BIG_THREADS) {
            const uint64_t j = base + tid;
            const bool have = j < rem[u];
            const uint32_t c2 = have ? B.col[p[u] + j] : EMPTY_KEY;
            const bool in2 = have && (last || (uint64_t)c2 < chi);
            if (in2) atomicOr(&bits[(c2 >> 5) - wb], 1u << (c2 & 31u));
            n = (uint32_t)__popcll(__ballot(in2));
            took += n;
#ifdef SG_SYM_STAMPS
            tph[11] += 1;
#endif
          }
          if (lane == 0 && took) atomicAdd(&s_cnt[e0 + u], took);
        }
      }
      SY_STAMP(3)
      __syncthreads();
      SY_STAMP(4)
      if (tid < nlong) pos[s_long[tid]] += s_cnt[tid];
      if (tid < 2) s_n[tid] = 0u;
      if (kb + BIG_CUR < len_a) __syncthreads(); // the lists are filled again by the next batch
    }
    // set bits per block of 256 words (four per lane); the words move to registers for the
    // write-out and the bitmap is cleared for the next window
    uint4 wd[BIG_BPW];
    uint32_t ex[BIG_BPW];
#pragma unroll
    for (uint32_t i = 0; i < BIG_BPW; ++i) {
      const uint32_t b = wv + BIG_WAVES * i;
      wd[i] = make_uint4(0u, 0u, 0u, 0u);
      ex[i] = 0u;
      if (b < nblk) { // wave-uniform
        uint4 *src = reinterpret_cast<uint4 *>(&bits[b * 256u + lane * 4u]);
        wd[i] = *src;
        const uint32_t pc = __popc(wd[i].x) + __popc(wd[i].y) + __popc(wd[i].z) + __popc(wd[i].w);
        uint32_t incl = 0;
        if (__ballot(pc != 0u)) {
          incl = group_incl_scan<64>(pc);
          *src = make_uint4(0u, 0u, 0u, 0u);
        }
        ex[i] = incl - pc;
        if (lane == 63) s_blk[b] = incl;
      }
    }
    SY_STAMP(5)
    __syncthreads();
    // every wavefront scans the block totals for itself (nblk <= 128: two per lane)
    const uint32_t t0 = 2u * lane < nblk ? s_blk[2u * lane] : 0u;
    const uint32_t t1 = 2u * lane + 1u < nblk ? s_blk[2u * lane + 1u] : 0u;
    const uint32_t pair_incl = group_incl_scan<64>(t0 + t1);
    const uint32_t tot = group_last<64>(pair_incl);
    const uint32_t pair_excl = pair_incl - (t0 + t1); // offset of block 2 * lane
#pragma unroll
    for (uint32_t i = 0; i < BIG_BPW; ++i) {
      const uint32_t b = wv + BIG_WAVES * i;
      if (b >= nblk) break;
      // block b is held by lane b / 2 (b is wave-uniform)
      const uint32_t boff = (uint32_t)__builtin_amdgcn_readlane((int)pair_excl, (int)(b >> 1)) +
                            ((b & 1u) ? (uint32_t)__builtin_amdgcn_readlane((int)t0, (int)(b >> 1)) : 0u);
      uint32_t at = carry + boff + ex[i];
      const uint32_t cbase = (wb + b * 256u + lane * 4u) << 5;
      const uint32_t w4[4] = {wd[i].x, wd[i].y, wd[i].z, wd[i].w};
#pragma unroll
      for (uint32_t j = 0; j < 4; ++j) {
        uint32_t word = w4[j];
        while (word) {
          out[at++] = cbase + 32u * j + (uint32_t)__ffs(word) - 1u;
          word &= word - 1u;
        }
      }
    }
    carry += tot;
    SY_STAMP(6)
    __syncthreads(); // s_blk and the lists are reused by the next window
    SY_STAMP(7)
  }
#ifdef SG_SYM_STAMPS
  if (lane == 0)
    for (int i = 0; i < 16; ++i) atomicAdd(&stamps[i], tph[i]);
#endif
  if (tid == 0) {
    m.nnz[row] = carry;
    m.num_key[row] = (uint8_t)num_key_of(m.sym_key[row], carry, m.hash_num_max);
  }
}

// ---- symbolic, big rows, with the window bounds of B's rows known ----------------------------------------
// k_sym_big finds the piece of a B row that falls into a bitmap window by walking the row from a
// cursor, one dependent load after the other; the rows of B that matter here are long, and one step
// of 512 entries per workgroup in flight leaves the kernel waiting for memory most of the time.
// For a B with at most BIG2_MAX_WIN windows of 2^ws columns, the position of the first entry of
// every row of B in every window is computed once per B (k_b_winbounds: wb[k * (nwin + 1) + w],
// relative to the row's start; wb[k][nwin] = the row's length).  The pieces of a window are then
// known before a single column is loaded: a batch of up to BIG_THREADS entries of the A row gives
// their lengths, a prefix sum lays the batch's products end to end, and thread x takes the products
// x, x + BIG_THREADS, ... - FOUR of them in flight, none of their loads depending on another.
// The write-out of the window's bitmap is k_sym_big's.
constexpr uint32_t BIG2_MAX_WIN = 16;
__global__ void __launch_bounds__(256)
    k_b_winbounds(uint64_t rows, const uint64_t *__restrict__ b_ptr, const uint32_t *__restrict__ b_col, uint32_t nwin,
                  uint32_t ws, uint32_t *__restrict__ wb) {
  // one thread per (row, window boundary 1..nwin-1); boundary 0 is 0 and boundary nwin the row's length
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t row = i / (nwin + 1u);
  const uint32_t w = (uint32_t)(i % (nwin + 1u));
  if (row >= rows) return;
  const uint64_t b0 = b_ptr[row], b1 = b_ptr[row + 1];
  uint64_t lo = b0, hi = b1;
  if (w == 0) hi = b0;
  else if (w == nwin) lo = b1;
  const uint64_t key = (uint64_t)w << ws;
  while (lo < hi) { // first entry whose column is >= key (the row is sorted)
    const uint64_t mid = lo + ((hi - lo) >> 1);
    if ((uint64_t)b_col[mid] < key) lo = mid + 1; else hi = mid;
  }
  wb[i] = (uint32_t)(lo - b0);
}

__host__ __device__ inline size_t sym_big2_lds(uint32_t win_words) {
  return (size_t)win_words * 4 + (size_t)BIG_THREADS * (8 + 4) + 16 + (size_t)(win_words / 64) * 4;
}
#ifdef SG_SYM2_STAMPS
// diagnostic build: cycles per phase of k_sym_big2, summed over the wavefronts
__device__ unsigned long long g_big2_stamps[8];
#define BIG2_STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tph[i] += t_ - tlast; tlast = t_; } while (0)
#else
#define BIG2_STAMP(i) do { } while (0)
#endif
__global__ void __launch_bounds__(BIG_THREADS) __attribute__((amdgpu_waves_per_eu(2 * BIG_WAVES / 4, 2 * BIG_WAVES / 4)))
    k_sym_big2(const uint32_t *__restrict__ rows, uint32_t nrows, Csr A, Csr B, RowMeta m, const uint32_t *__restrict__ wb,
               uint32_t nwin, uint32_t ws) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint32_t win_words = 1u << (ws - 5u);
  uint32_t *bits = reinterpret_cast<uint32_t *>(smem);
  uint64_t *s_base = reinterpret_cast<uint64_t *>(smem + (size_t)win_words * 4);
  uint32_t *s_pre = reinterpret_cast<uint32_t *>(s_base + BIG_THREADS); // exclusive prefix of the batch's piece lengths (+ total)
  uint32_t *s_blk = s_pre + BIG_THREADS + 4;
  __shared__ uint64_t s_wave[BIG_WAVES];
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
  const uint32_t row = scalar_view(rows)[blockIdx.x];
  const uint64_t a0 = scalar_view(A.ptr)[row], a1 = scalar_view(A.ptr)[row + 1];
  const uint32_t len_a = (uint32_t)min(a1 - a0, (uint64_t)0xFFFFFFFFu);
  const uint32_t cmin = m.cmin[row], cmax = m.cmax[row];
  uint32_t *out = m.temp_cols + m.t_off[row];
  uint32_t carry = 0;
#ifdef SG_SYM2_STAMPS
  unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = __builtin_amdgcn_s_memtime();
#endif
  for (uint32_t w = tid * 4u; w < win_words; w += BIG_THREADS * 4u)
    *reinterpret_cast<uint4 *>(&bits[w]) = make_uint4(0u, 0u, 0u, 0u);
  // A row of up to BIG_THREADS entries (one per thread) keeps its entry's B row in registers and
  // fetches the bounds of the NEXT window while this one is walked: nothing of the row's own
  // description is waited for inside a window.
  const bool one_batch = len_a <= BIG_THREADS;
  const uint32_t win0 = cmin >> ws, win1 = cmax >> ws;
  uint64_t my_b0 = 0;
  const uint32_t *my_wb = wb; // (a valid address for the threads without an entry)
  bool have = false;
  uint32_t o_lo = 0, o_hi = 0; // bounds of my B row in the current window
  if (one_batch && tid < len_a) {
    const uint32_t k = A.col[a0 + tid];
    my_b0 = B.ptr[k];
    my_wb = wb + (size_t)k * (nwin + 1u);
    have = true;
    o_lo = my_wb[win0];
    o_hi = my_wb[win0 + 1u];
  }
  for (uint32_t win = win0; win <= win1; ++win) {
    // words of this window that can hold an output of the row
    const uint32_t wlo = max(win << (ws - 5u), cmin >> 5), whi = min(((win + 1u) << (ws - 5u)) - 1u, cmax >> 5);
    const uint32_t nw = whi - wlo + 1u;
    const uint32_t nblk = (nw + 255u) >> 8;
    uint32_t o_next = o_hi;
    if (one_batch && have && win < win1) o_next = my_wb[win + 2u]; // end of my piece in the next window: lands during this one
    for (uint32_t kb = 0; kb < len_a; kb += BIG_THREADS) {
      // the pieces of this batch of A entries in this window
      const uint32_t t = kb + tid;
      uint32_t plen = 0;
      uint64_t base = 0;
      if (one_batch) {
        plen = o_hi - o_lo;
        base = my_b0 + o_lo;
      } else if (t < len_a) {
        const uint32_t k = A.col[a0 + t];
        const uint32_t *wk = wb + (size_t)k * (nwin + 1u) + win;
        const uint32_t o0 = wk[0], o1 = wk[1];
        plen = o1 - o0;
        base = B.ptr[k] + o0;
      }
      uint64_t total64;
      const uint32_t excl = (uint32_t)block_excl_scan64(plen, s_wave, &total64); // (two barriers inside: the batch before is done with s_pre / s_base)
      s_pre[tid] = excl;
      s_base[tid] = base - excl; // (product q of the batch sits at s_base[entry] + q: one read and no subtraction per product)
      if (tid == 0) s_pre[BIG_THREADS] = (uint32_t)total64;
      __syncthreads();
      BIG2_STAMP(0);
      const uint32_t total = (uint32_t)total64; // (a batch holds fewer than 2^32 products: 512 pieces of a window of 2^19 columns)
      // entry whose piece holds product q: the last one whose prefix is <= q (pieces of length 0 share a prefix with their successor)
      auto entry_of = [&](uint32_t q) {
        uint32_t e = 0;
#pragma unroll
        for (uint32_t sft = BIG_THREADS / 2; sft > 0; sft >>= 1)
          if (s_pre[e + sft] <= q) e += sft;
        return e;
      };
      // The 64 products of a wavefront's step lie in a few neighbouring pieces.  Lane l keeps the prefix of
      // entry 8 l; two ballots (prefixes <= the step's first and last product) bound the entries of the
      // step, and where they span 8 or 16 the search takes 3 or 4 steps instead of 9.
      static_assert(BIG_THREADS == 512, "one coarse prefix per lane: every 8th of 512 entries");
      const uint32_t coarse = s_pre[lane * 8u];
      constexpr int NF = 8; // products of a thread whose loads are in flight together
      for (uint32_t q0 = 0; q0 < total; q0 += BIG_THREADS * NF) {
        uint32_t c[NF];
#pragma unroll
        for (int u = 0; u < NF; ++u) {
          const uint32_t qw = q0 + (uint32_t)u * BIG_THREADS + wv * 64u; // (wave-uniform) the step's first product
          const uint32_t q = qw + lane;
          c[u] = EMPTY_KEY;
          if (qw < total) { // wave-uniform
            const uint32_t n_lo = (uint32_t)__popcll(__ballot(coarse <= qw));
            const uint32_t n_hi = (uint32_t)__popcll(__ballot(coarse <= min(qw + 63u, total - 1u)));
            uint32_t e;
            if (n_hi - n_lo <= 1u) { // the step's entries lie in [8 (n_lo - 1), 8 n_hi): 8 or 16 of them
              e = (n_lo - 1u) * 8u;   // (the prefix of entry 0 is 0: n_lo >= 1)
              if (n_hi != n_lo && s_pre[e + 8u] <= q) e += 8u;
#pragma unroll
              for (uint32_t sft = 4; sft > 0; sft >>= 1)
                if (s_pre[e + sft] <= q) e += sft;
            } else {
              e = entry_of(q);
            }
            if (q < total) c[u] = B.col[s_base[e] + q];
          }
        }
#pragma unroll
        for (int u = 0; u < NF; ++u)
          if (c[u] != EMPTY_KEY) atomicOr(&bits[(c[u] >> 5) - wlo], 1u << (c[u] & 31u));
      }
      BIG2_STAMP(1);
      __syncthreads(); // every piece of the batch is in the bitmap; s_pre / s_base may be rewritten
      BIG2_STAMP(2);
    }
    o_lo = o_hi;
    o_hi = o_next;
    // set bits per block of 256 words (four per lane); the words move to registers for the
    // write-out and the bitmap is cleared for the next window
    uint4 wd[BIG_BPW];
    uint32_t ex[BIG_BPW];
#pragma unroll
    for (uint32_t i = 0; i < BIG_BPW; ++i) {
      const uint32_t b = wv + BIG_WAVES * i;
      wd[i] = make_uint4(0u, 0u, 0u, 0u);
      ex[i] = 0u;
      if (b < nblk) { // wave-uniform
        uint4 *src = reinterpret_cast<uint4 *>(&bits[b * 256u + lane * 4u]);
        wd[i] = *src;
        const uint32_t pc = __popc(wd[i].x) + __popc(wd[i].y) + __popc(wd[i].z) + __popc(wd[i].w);
        uint32_t incl = 0;
        if (__ballot(pc != 0u)) {
          incl = group_incl_scan<64>(pc);
          *src = make_uint4(0u, 0u, 0u, 0u);
        }
        ex[i] = incl - pc;
        if (lane == 63) s_blk[b] = incl;
      }
    }
    BIG2_STAMP(3);
    __syncthreads();
    BIG2_STAMP(4);
    // every wavefront scans the block totals for itself (nblk <= 128: two per lane)
    const uint32_t t0 = 2u * lane < nblk ? s_blk[2u * lane] : 0u;
    const uint32_t t1 = 2u * lane + 1u < nblk ? s_blk[2u * lane + 1u] : 0u;
    const uint32_t pair_incl = group_incl_scan<64>(t0 + t1);
    const uint32_t tot = group_last<64>(pair_incl);
    const uint32_t pair_excl = pair_incl - (t0 + t1); // offset of block 2 * lane
#pragma unroll
    for (uint32_t i = 0; i < BIG_BPW; ++i) {
      const uint32_t b = wv + BIG_WAVES * i;
      if (b >= nblk) break;
      const uint32_t boff = (uint32_t)__builtin_amdgcn_readlane((int)pair_excl, (int)(b >> 1)) +
                            ((b & 1u) ? (uint32_t)__builtin_amdgcn_readlane((int)t0, (int)(b >> 1)) : 0u);
      uint32_t *dst = out + carry + boff; // (wave-uniform)
      const uint32_t cbase = (wlo + b * 256u + lane * 4u) << 5;
      const uint32_t w4[4] = {wd[i].x, wd[i].y, wd[i].z, wd[i].w};
      // every lane stores the columns of its own four words
      uint32_t at = ex[i];
#pragma unroll
      for (uint32_t j = 0; j < 4; ++j) {
        uint32_t word = w4[j];
        while (word) {
          dst[at++] = cbase + 32u * j + (uint32_t)__ffs(word) - 1u;
          word &= word - 1u;
        }
      }
    }
    BIG2_STAMP(5);
    carry += tot;
    __syncthreads(); // s_blk is reused by the next window
    BIG2_STAMP(6);
  }
  if (tid == 0) {
    m.nnz[row] = carry;
    m.num_key[row] = (uint8_t)num_key_of(m.sym_key[row], carry, m.hash_num_max);
  }
#ifdef SG_SYM2_STAMPS
  if (lane == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(&g_big2_stamps[i], tph[i]);
#endif
}

// ---- numeric, big rows: slices of the output row, one wavefront per slice ---------------------------
// Needs B rows with strictly increasing columns.  The sorted structural columns of a big row are
// cut into chunks of SLICE_W outputs; a work item is a run of consecutive chunks of one row and is
// carried out by ONE wavefront, chunk after chunk.  For a chunk the wavefront keeps the chunk's
// columns and accumulators in LDS and walks the A row in storage order, taking from every B row
// only the entries whose columns fall into the chunk (a contiguous piece, since the B row is
// sorted): every output still receives its additions in the order of the CPU loop, and different
// chunks of a row run on different wavefronts at the same time.
//
// An item first builds the list of A entries whose B rows reach into its column range at all
// ({position in B, end position, A value, next column}, in A order; global scratch, `list_cap`
// entries per workgroup).  A chunk then looks only at the list's `next column` to find the entries
// it has to touch, and advances their positions for the chunk that follows.
#ifndef SG_SLICE_UG
#define SG_SLICE_UG 4
#endif
#ifndef SG_SLICE_W
#define SG_SLICE_W 1024
#endif
constexpr uint32_t SLICE_W = SG_SLICE_W; // 4^k, 2 * 4^k or 3 * 4^k, a multiple of 64
// largest power of four that divides SLICE_W into at most three parts: the position search splits
// into those parts first and four ways from there on
constexpr uint32_t SLICE_P4 = SLICE_W % 1024u == 0 && SLICE_W / 1024u <= 3u ? 1024u : SLICE_W % 256u == 0 && SLICE_W / 256u <= 3u ? 256u : 64u;
static_assert(SLICE_W % 64u == 0 && SLICE_W % SLICE_P4 == 0 && SLICE_W / SLICE_P4 <= 3u, "unsupported SLICE_W");
// LDS of one workgroup: cols u32 | vals f64 | owner u32 (claim rounds only) | flat-walk scratch
__host__ __device__ constexpr uint32_t slice_lds(bool hw_order) { return SLICE_W * (hw_order ? 12u : 16u) + FLAT_BYTES; }
#ifndef SG_SLICE_LONG_MIN
#define SG_SLICE_LONG_MIN 16
#endif
constexpr uint32_t SLICE_LONG_MIN = SG_SLICE_LONG_MIN; // mean expected piece length from which a block takes its pieces one B row at a time
#ifndef SG_SLICE_PF
#define SG_SLICE_PF 3
#endif
constexpr int SLICE_PF = SG_SLICE_PF; // steps of 64 products whose loads fly ahead of the step that is added

struct SliceItem {
  uint32_t row, c0, nc; // chunks [c0, c0 + nc) of the row
};

// chunks per item: long A rows get longer runs, so that building the list pays off
// run_min >= 4; run_div: entries of the A row per chunk of a run
__host__ __device__ inline uint32_t slice_run(uint32_t len_a, uint32_t run_min, uint32_t run_div) {
  uint32_t g = len_a / run_div;
  const uint32_t most = run_min > 64u ? run_min : 64u;
  return g < run_min ? run_min : g > most ? most : g;
}

__global__ void __launch_bounds__(256)
    k_slice_count(const uint32_t *__restrict__ rows, uint32_t nrows, const uint64_t *__restrict__ a_ptr,
                  const uint32_t *__restrict__ nnz, uint32_t *__restrict__ count, uint32_t run_min, uint32_t run_div) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nrows) return;
  const uint32_t row = rows[i];
  const uint32_t chunks = (nnz[row] + SLICE_W - 1u) / SLICE_W;
  const uint32_t g = slice_run((uint32_t)(a_ptr[row + 1] - a_ptr[row]), run_min, run_div);
  count[i] = (chunks + g - 1u) / g;
}

__global__ void __launch_bounds__(256)
    k_slice_fill(const uint32_t *__restrict__ rows, uint32_t nrows, const uint64_t *__restrict__ a_ptr,
                 const uint32_t *__restrict__ nnz, const uint64_t *__restrict__ offset,
                 SliceItem *__restrict__ items, uint32_t *__restrict__ keys, RowMeta m, uint32_t run_min,
                 uint32_t run_div, uint32_t by_cost) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nrows) return;
  const uint32_t row = rows[i];
  const uint32_t chunks = (nnz[row] + SLICE_W - 1u) / SLICE_W;
  const uint32_t g = slice_run((uint32_t)(a_ptr[row + 1] - a_ptr[row]), run_min, run_div);
  const uint32_t *tin = keys ? m.temp_cols + m.t_off[m.lead[row]] : nullptr;
  const uint64_t len_a = a_ptr[row + 1] - a_ptr[row];
  uint64_t o = offset[i];
  for (uint32_t c = 0; c < chunks; c += g) {
    const uint32_t nc = min(g, chunks - c);
    // the sort key: the item's first output column, or (by_cost, the weight of an A entry in quarters of a
    // product) what the item costs, the dearest first: every chunk walks the item's list of A entries and
    // adds a chunk's worth of products
    if (keys) {
      const uint64_t cost = (uint64_t)nc * ((len_a * (by_cost & 0xFFFFu) >> 2) + SLICE_W);
      if (by_cost >> 16) { // by the cost's order of magnitude (a power of two), the dearest first; inside one, by first column
        const uint32_t mag = 63u - (uint32_t)__clzll((long long)(cost | 1ull)); // 0..63
        keys[o] = ((63u - mag) << 26) | (tin[(size_t)c * SLICE_W] >> 6);
      } else
        keys[o] = by_cost ? 0xFFFFFFFEu - (uint32_t)min((uint64_t)0xFFFFFFF0u, cost) : tin[(size_t)c * SLICE_W];
    }
    items[o++] = SliceItem{row, c, nc};
  }
}

// Rows of the big-row symbolic bin in the order of their cost: sort keys (dearest first) and records
// {row, 0, 0} for the 12-byte record sort, and the row ids back out of the sorted records.
__global__ void __launch_bounds__(256)
    k_big_order_fill(const uint32_t *__restrict__ rows, uint32_t nrows, const uint32_t *__restrict__ ub,
                     uint32_t *__restrict__ keys, SliceItem *__restrict__ recs) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nrows) return;
  const uint32_t row = rows[i];
  keys[i] = ~ub[row];
  recs[i] = SliceItem{row, 0u, 0u};
}
__global__ void __launch_bounds__(256)
    k_big_order_rows(const SliceItem *__restrict__ recs, uint32_t nrows, uint32_t *__restrict__ rows_out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nrows) rows_out[i] = recs[i].row;
}

// longest A row among the rows of the numeric phase's big-row bin (atomic max into out[0], zero on entry)
__global__ void __launch_bounds__(256)
    k_slice_maxlen(uint32_t nrows, const uint64_t *__restrict__ a_ptr, const uint8_t *__restrict__ num_key,
                   uint32_t *__restrict__ out) {
  uint32_t mx = 0;
  for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += gridDim.x * blockDim.x)
    if (num_key[r] == (uint8_t)NUM_GLOBAL) mx = max(mx, (uint32_t)min((uint64_t)0xFFFFFFFFu, a_ptr[r + 1] - a_ptr[r]));
  mx = group_max<64>(mx);
  if ((threadIdx.x & 63u) == 0 && mx) atomicMax(out, mx);
}

// first position in [lo, hi) of B.col whose column is >= key (per lane)
// STRIDE: distance of consecutive columns in `col`, in uint32 (3 for the packed {column, value} records)
template <int STRIDE = 1>
__device__ __forceinline__ uint64_t lower_bound_col(const uint32_t *__restrict__ col, uint64_t lo, uint64_t hi,
                                                    uint64_t key) {
  while (lo < hi) {
    const uint64_t mid = lo + ((hi - lo) >> 1);
    if ((uint64_t)col[mid * STRIDE] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}
// the same for N independent ranges per lane, their loads in flight together
template <int N, int STRIDE = 1>
__device__ __forceinline__ void lower_bound_col_n(const uint32_t *__restrict__ col, uint64_t (&lo)[N], uint64_t (&hi)[N],
                                                  uint64_t key) {
  for (;;) {
    bool open = false;
#pragma unroll
    for (int u = 0; u < N; ++u) open |= lo[u] < hi[u];
    if (!__any(open)) break;
    uint32_t c[N];
    uint64_t mid[N];
#pragma unroll
    for (int u = 0; u < N; ++u) {
      mid[u] = lo[u] + ((hi[u] - lo[u]) >> 1);
      c[u] = lo[u] < hi[u] ? col[mid[u] * STRIDE] : 0u;
    }
#pragma unroll
    for (int u = 0; u < N; ++u)
      if (lo[u] < hi[u]) {
        if ((uint64_t)c[u] < key) lo[u] = mid[u] + 1; else hi[u] = mid[u];
      }
  }
}

// ---- column sample of B: the last column of every block of 32 entries (block j = entries [32 j, 32 j + 32)
// of B.col, 128 bytes = one cache line; the last block is padded with the largest column) ------------
// A search for a column inside a row of B first narrows its range to one block over the sample -
// 1/32 of B.col, 2 MB for 16 M entries, resident in L2 - and finishes inside that block's line.
// The plain binary search pulled a whole line for every one of its far probes, and the kernel that
// searches most is bound by the lines it pulls in.
constexpr uint32_t SAMPLE_LOG2 = 5;
__global__ void __launch_bounds__(256) k_b_colsample(uint64_t nnz, const uint32_t *__restrict__ b_col, uint32_t *__restrict__ samp) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t nb = (nnz + 31u) >> SAMPLE_LOG2;
  if (j >= nb) return;
  const uint64_t last = min(nnz - 1u, (j << SAMPLE_LOG2) + 31u);
  samp[j] = b_col[last];
}
// first position in [lo, hi) of B.col (one row's range, strictly increasing) whose column is >= key
__device__ __forceinline__ uint64_t lower_bound_sampled(const uint32_t *__restrict__ col, const uint32_t *__restrict__ samp,
                                                        uint64_t lo, uint64_t hi, uint64_t key) {
  if (lo >= hi) return lo;
  // blocks that lie inside the range as a whole: their last columns belong to this row and increase
  uint64_t j0 = (lo + 31u) >> SAMPLE_LOG2, j1 = hi >> SAMPLE_LOG2;
  if (j0 < j1) {
    uint64_t a = j0, b = j1;
    while (a < b) { // first whole block whose last column is >= key
      const uint64_t mid = a + ((b - a) >> 1);
      if ((uint64_t)samp[mid] < key) a = mid + 1; else b = mid;
    }
    // every entry of the whole blocks before block a is < key; the head before block j0 only when a > j0
    if (a > j0) lo = a << SAMPLE_LOG2;
    if (a < j1) hi = (a << SAMPLE_LOG2) + 32u;
  }
  while (lo < hi) { // at most two lines
    const uint64_t mid = lo + ((hi - lo) >> 1);
    if ((uint64_t)col[mid] < key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// `work`: SLICE_QUEUE_STRIDE * q = items handed out from queue q (zero on entry); queue q holds the
// items [total * q / queues, total * (q + 1) / queues), total = n_items[0].  Workgroups are dealt to
// the eight XCDs round robin, so workgroup b starts on queue b % queues and moves on to the next
// one when a queue runs dry.  With eight queues the wavefronts of ONE XCD work on consecutive items,
// that is on the chunks of a few rows of A at a time: the B rows those chunks share come from HBM
// once and from that XCD's L2 afterwards, instead of once per XCD.  o.zero_count[3] is set
// when a list entry read back from scratch is not a position inside B (never seen; it would mean
// that the list was damaged) - the entry is skipped and the host reports the multiply as failed.
// list_scratch: per workgroup, list_cap entries of {cur u64 | end u64 | av f64 | next u32 | queue u32}.
constexpr uint32_t SLICE_ENTRY_BYTES = 32;
constexpr uint32_t SLICE_MAX_QUEUES = 8;
constexpr uint32_t SLICE_QUEUE_STRIDE = 16; // counters, in uint64: one 128-byte line each
__global__ void __launch_bounds__(64)
    k_num_slice(const SliceItem *__restrict__ items, const uint64_t *__restrict__ n_items, Csr A, Csr B, RowMeta m,
                NumOut o, unsigned long long *__restrict__ work, unsigned char *__restrict__ list_scratch,
                uint64_t list_cap, uint64_t nnz_b, const Rec12 *__restrict__ packed, uint32_t queues, uint32_t hw_order,
                const uint32_t *__restrict__ samp) {
  // the entries of B as 12-byte records {column, value} (optional): a short piece of a row of B (a few
  // entries, the common case inside a chunk) then lies in one 128-byte line, or two, instead of one
  // line of columns and one of values, and this kernel is bound by the lines it pulls in
  const uint32_t *__restrict__ pcol = reinterpret_cast<const uint32_t *>(packed);
  auto bcol = [&](uint64_t i) -> uint32_t { return packed ? pcol[i * 3u] : B.col[i]; };
  auto bpair = [&](uint64_t i, uint32_t &c, double &v) {
    if (packed) {
      const Rec12 t = packed[i];
      c = t.col;
      v = __longlong_as_double((long long)(((unsigned long long)t.val_hi << 32) | t.val_lo));
    } else {
#if defined(SG_ABL_COL)  // diagnostic builds: the loads of one kind go to a cache-resident kilobyte (wrong results)
      c = B.col[i & 1023u];
#else
      c = B.col[i];
#endif
#if defined(SG_ABL_VAL)
      v = B.val[i & 1023u];
#else
      v = B.val[i];
#endif
    }
  };
  // (samp: the column sample of B, see k_b_colsample; null: plain binary search, as with the packed records)
  auto lbound = [&](uint64_t lo, uint64_t hi, uint64_t key) {
    if (packed) return lower_bound_col<3>(pcol, lo, hi, key);
    return samp ? lower_bound_sampled(B.col, samp, lo, hi, key) : lower_bound_col<1>(B.col, lo, hi, key);
  };
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t *scol = reinterpret_cast<uint32_t *>(smem);
  double *vals = reinterpret_cast<double *>(smem + SLICE_W * 4u);
  uint32_t *owner = reinterpret_cast<uint32_t *>(smem + SLICE_W * 12u);
  unsigned char *flat = smem + SLICE_W * (hw_order ? 12u : 16u);
  uint32_t *st = reinterpret_cast<uint32_t *>(flat);
  uint64_t *p0s = reinterpret_cast<uint64_t *>(flat + 256);
  double *avs = reinterpret_cast<double *>(flat + 768);
  const uint32_t lane = threadIdx.x;
  uint64_t *l_cur = reinterpret_cast<uint64_t *>(list_scratch + (size_t)blockIdx.x * list_cap * SLICE_ENTRY_BYTES);
  uint64_t *l_end = l_cur + list_cap;
  double *l_av = reinterpret_cast<double *>(l_end + list_cap);
  uint32_t *l_next = reinterpret_cast<uint32_t *>(l_av + list_cap);
  uint32_t *l_q = l_next + list_cap;
  const uint64_t total_items = n_items[0];
  if (!hw_order)
    for (uint32_t i = lane; i < SLICE_W; i += 64) owner[i] = 0u;
  uint32_t stamp = 0;
  uint32_t nz = 0;
  constexpr int U = 4;  // blocks of list entries whose loads and searches are in flight together
  constexpr int UG = SG_SLICE_UG; // B row pieces whose loads and column searches are in flight together
#ifdef SG_SLICE_STAMPS
  // diagnostic build: cycles per phase, summed over the wavefronts into o.zero_count[8..]
  unsigned long long tph[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tlast = __builtin_amdgcn_s_memtime();
#define SG_STAMP(i) { const unsigned long long tn = __builtin_amdgcn_s_memtime(); tph[i] += tn - tlast; tlast = tn; }
#else
#define SG_STAMP(i)
#endif
  const uint32_t home = blockIdx.x % queues;
  uint32_t tried = 0; // queues found dry so far
  for (;;) {
    unsigned long long t = 0;
    while (tried < queues) { // wave-uniform
      const uint32_t q = (home + tried) % queues;
      const unsigned long long q_lo = total_items * q / queues, q_hi = total_items * (q + 1ull) / queues;
      if (lane == 0) t = q_lo + atomicAdd(work + (size_t)SLICE_QUEUE_STRIDE * q, 1ull);
      t = readlane64(t, 0);
      if (t < q_hi) break;
      ++tried;
    }
    if (tried >= queues) break;
    SG_STAMP(0)
    const SliceItem it = items[t];
    const uint32_t row = it.row;
    const uint32_t n = m.nnz[row];
    const uint64_t off = o.c_ptr[row];
    const uint32_t *tin = m.temp_cols + m.t_off[m.lead[row]];
    const uint64_t a0 = A.ptr[row], a1 = A.ptr[row + 1];
    const uint32_t chunks = (n + SLICE_W - 1u) / SLICE_W;
    const bool item_first = it.c0 == 0, item_last = it.c0 + it.nc == chunks;
    const uint64_t clo_item = item_first ? 0ull : (uint64_t)tin[(size_t)it.c0 * SLICE_W];
    const uint64_t chi_item = item_last ? (1ull << 32) : (uint64_t)tin[(size_t)(it.c0 + it.nc) * SLICE_W];
    // -- the list: A entries whose B rows have entries in [clo_item, chi_item), in A order.  A row of
    //    up to 64 entries keeps it in registers (one entry per lane).
    const bool in_regs = a1 - a0 <= 64;
    uint64_t cur = 0, end = 0;
    double av = 0.0;
    uint32_t nextc = EMPTY_KEY;
    uint32_t nlist = 0;
    if (in_regs) {
      const uint64_t ka = a0 + lane;
      uint64_t lo = 0, b1 = 0;
      if (ka < a1) {
        const uint32_t k = A.col[ka];
        av = A.val[ka];
        const uint64_t b0 = B.ptr[k];
        b1 = B.ptr[k + 1];
        lo = item_first ? b0 : lbound(b0, b1, clo_item);
      }
      nextc = lo < b1 ? bcol(lo) : EMPTY_KEY;
      const bool ne = lo < b1 && (uint64_t)nextc < chi_item;
      cur = ne ? lo : 0;
      end = ne ? b1 : 0;
      if (!ne) nextc = EMPTY_KEY;
    } else {
      for (uint64_t ab = a0; ab < a1; ab += 64 * U) { // U blocks of 64 entries, their searches together
        uint64_t lo[U], hi[U], b1[U];
        double a[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint64_t ka = ab + 64u * u + lane;
          lo[u] = hi[u] = b1[u] = 0;
          a[u] = 0.0;
          if (ka < a1) {
            const uint32_t k = A.col[ka];
            a[u] = A.val[ka];
            lo[u] = B.ptr[k];
            b1[u] = B.ptr[k + 1];
            hi[u] = item_first ? lo[u] : b1[u]; // empty range: nothing to search
          }
        }
        if (packed) lower_bound_col_n<U, 3>(pcol, lo, hi, clo_item);
        else if (samp) {
#pragma unroll
          for (int u = 0; u < U; ++u) lo[u] = lower_bound_sampled(B.col, samp, lo[u], hi[u], clo_item); // (probes of the sample: L2 hits)
        } else lower_bound_col_n<U, 1>(B.col, lo, hi, clo_item);
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t nc0 = lo[u] < b1[u] ? bcol(lo[u]) : EMPTY_KEY;
          const bool ne = lo[u] < b1[u] && (uint64_t)nc0 < chi_item;
          const uint64_t mask = __ballot(ne);
          if (ne) {
            const uint32_t e = nlist + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
            l_cur[e] = lo[u]; l_end[e] = b1[u]; l_av[e] = a[u]; l_next[e] = nc0;
          }
          nlist += (uint32_t)__popcll(mask);
        }
      }
      // the list is read back by other lanes: all its stores have to be done
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    SG_STAMP(1)
    // -- the chunks
    for (uint32_t c = it.c0; c < it.c0 + it.nc; ++c) {
      const uint32_t pbase = c * SLICE_W;
      const uint32_t ncols = min(SLICE_W, n - pbase);
      const bool chunk_last = c + 1 == it.c0 + it.nc; // of the item
      const uint64_t chi = chunk_last ? chi_item : (uint64_t)tin[(size_t)pbase + SLICE_W];
      const uint32_t chunks_left = chunks - c;
      lds_fence();
      {
        uint32_t tc[SLICE_W / 64]; // the chunk's columns: all loads in flight before the first LDS store
#pragma unroll
        for (uint32_t i = 0; i < SLICE_W / 64; ++i) {
          const uint32_t p = i * 64u + lane;
          tc[i] = p < ncols ? tin[(size_t)pbase + p] : EMPTY_KEY;
        }
#pragma unroll
        for (uint32_t i = 0; i < SLICE_W / 64; ++i) {
          scol[i * 64u + lane] = tc[i];
          vals[i * 64u + lane] = 0.0;
        }
      }
      if (stamp > CLAIM_STAMP_LIMIT) { // claims compare tags: start over before the stamp field overflows
        for (uint32_t i = lane; i < SLICE_W; i += 64) owner[i] = 0u;
        stamp = 0;
      }
      lds_fence();
      SG_STAMP(2)
      // position of a column that is in the chunk: four-way search, three probes per level in
      // flight together (five dependent LDS round trips instead of ten)
      auto find = [&](uint32_t col) {
        uint32_t pos = 0;
        if constexpr (SLICE_W > SLICE_P4) {
          uint32_t part = 0;
#pragma unroll
          for (uint32_t j = 1; j < SLICE_W / SLICE_P4; ++j) part += scol[j * SLICE_P4] <= col ? 1u : 0u;
          pos = part * SLICE_P4;
        }
#pragma unroll
        for (uint32_t len = SLICE_P4 / 4; len > 0; len >>= 2) {
          const uint32_t a = scol[pos + len], b = scol[pos + 2u * len], c = scol[pos + 3u * len];
          pos += len * ((a <= col ? 1u : 0u) + (b <= col ? 1u : 0u) + (c <= col ? 1u : 0u));
        }
        return pos;
      };
      // One block of up to 64 list entries (one per lane; `act` = has entries in this chunk): adds
      // their pieces in A order, moves cur / nextc on.
      auto block = [&](bool act, uint64_t &cur, uint64_t end, double av, uint32_t &nextc) {
        const uint64_t amask = __ballot(act);
        const uint32_t nact = (uint32_t)__popcll(amask);
        // expected length of a piece: what is left of the B row, spread over the chunks that follow
        const uint32_t est = act ? (uint32_t)min(end - cur, (uint64_t)SLICE_W) / chunks_left : 0u;
        SG_STAMP(3)
        if (group_sum<64>(est) >= SLICE_LONG_MIN * nact) {
          // long pieces: one B row at a time, its entries over the lanes (distinct columns: no
          // conflicts); U pieces are loaded and searched together, then added in A order
          // group = UG pieces; the loads of the next group fly while this one is searched and added
          struct Group {
            uint32_t l[UG], rem[UG], col[UG];
            uint64_t p0[UG], pend[UG];
            double a[UG], v[UG];
            bool on[UG];
          };
          uint64_t mask = amask;
          auto fetch = [&](Group &g) {
#pragma unroll
            for (int u = 0; u < UG; ++u) {
              g.on[u] = mask != 0;
              g.l[u] = g.on[u] ? (uint32_t)__builtin_ctzll(mask) : 0u;
              if (g.on[u]) mask &= mask - 1ull;
              g.p0[u] = readlane64(cur, g.l[u]);
              g.pend[u] = readlane64(end, g.l[u]);
              g.a[u] = readlane_f64(av, g.l[u]);
              g.rem[u] = g.on[u] ? (uint32_t)min(g.pend[u] - g.p0[u], (uint64_t)64) : 0u;
            }
#pragma unroll
            for (int u = 0; u < UG; ++u) {
              const bool have = lane < g.rem[u];
              g.col[u] = EMPTY_KEY;
              g.v[u] = 0.0;
              if (have) bpair(g.p0[u] + lane, g.col[u], g.v[u]);
            }
          };
          Group g, gn;
          fetch(g);
          while (g.on[0]) {
            gn.on[0] = false;
            if (mask) fetch(gn); // other lanes' pieces: not touched by what follows
            uint32_t pos[UG];
            bool in[UG];
#pragma unroll
            for (int u = 0; u < UG; ++u) {
              in[u] = lane < g.rem[u] && (uint64_t)g.col[u] < chi;
              pos[u] = find(in[u] ? g.col[u] : 0u);
            }
#pragma unroll
            for (int u = 0; u < UG; ++u) {
              if (!g.on[u]) continue; // wave-uniform
              if (in[u]) lds_add_f64(&vals[pos[u]], __dmul_rn(g.a[u], g.v[u]));
              uint32_t took = (uint32_t)__popcll(__ballot(in[u]));
              uint64_t ncur = g.p0[u] + took;
              uint64_t notin = __ballot(lane < g.rem[u] && !in[u]);
              uint32_t nxt = EMPTY_KEY;
              if (notin) {
                nxt = (uint32_t)__builtin_amdgcn_readlane((int)g.col[u], (int)__builtin_ctzll(notin));
              } else {
                while (ncur < g.pend[u]) { // the whole block belonged to the chunk and the B row goes on
                  const uint32_t r2 = (uint32_t)min(g.pend[u] - ncur, (uint64_t)64);
                  const bool have = lane < r2;
                  uint32_t c2 = EMPTY_KEY;
                  double v2 = 0.0;
                  if (have) bpair(ncur + lane, c2, v2);
                  const bool in2 = have && (uint64_t)c2 < chi;
                  if (in2) lds_add_f64(&vals[find(c2)], __dmul_rn(g.a[u], v2));
                  took = (uint32_t)__popcll(__ballot(in2));
                  ncur += took;
                  notin = __ballot(have && !in2);
                  if (notin) {
                    nxt = (uint32_t)__builtin_amdgcn_readlane((int)c2, (int)__builtin_ctzll(notin));
                    break;
                  }
                }
              }
              if (lane == g.l[u]) { cur = ncur; nextc = nxt; }
            }
            g = gn;
          }
          SG_STAMP(4)
#ifdef SG_SLICE_STAMPS
          tph[18] += 1;    // long-piece blocks
          tph[19] += nact; // long pieces
#endif
        } else {
          // short pieces: products laid end to end over the lanes, conflicts resolved in lane order.
          // A piece's entries are distinct columns of this chunk: at most ncols, at most chi - nextc;
          // the three entries after the first are looked at directly, a longer piece is searched.
          uint32_t cnt = 0;
          uint64_t slo = 0, shi = 0, lim = 0;
          uint32_t c1 = EMPTY_KEY, c2 = EMPTY_KEY, c3 = EMPTY_KEY;
          if (act) {
            lim = min(end, cur + min((uint64_t)ncols, chi - nextc));
            c1 = cur + 1 < end ? bcol(cur + 1) : EMPTY_KEY;
            c2 = cur + 2 < end ? bcol(cur + 2) : EMPTY_KEY;
            c3 = cur + 3 < end ? bcol(cur + 3) : EMPTY_KEY;
            const bool i1 = cur + 1 < lim && (uint64_t)c1 < chi;
            const bool i2 = i1 && cur + 2 < lim && (uint64_t)c2 < chi;
            const bool i3 = i2 && cur + 3 < lim && (uint64_t)c3 < chi;
            cnt = 1u + (i1 ? 1u : 0u) + (i2 ? 1u : 0u) + (i3 ? 1u : 0u);
            if (i3 && cur + 4 < lim) { slo = cur + 4; shi = lim; }
          }
          SG_STAMP(7)
          const bool searched = slo < shi;
          const uint64_t found = lbound(slo, shi, chi);
          if (searched) cnt = (uint32_t)(found - cur);
          SG_STAMP(8)
#ifdef SG_SLICE_STAMPS
          tph[10] += 1;                                  // flat blocks
          tph[11] += __any(searched) ? 1 : 0;            // ... with a piece longer than four
          tph[12] += nact;                               // pieces
          tph[13] += group_sum<64>(cnt);                 // products
          tph[14] += in_regs ? 1 : 0;                    // blocks of rows that keep the list in registers
          tph[15] += (uint32_t)__popcll(__ballot(cnt == 1u));
          tph[16] += (uint32_t)__popcll(__ballot(cnt > 4u));
          tph[17] += group_sum<64>(cnt > 4u ? cnt : 0u); // products of pieces longer than four
#endif
          const uint32_t incl = group_incl_scan<64>(cnt);
          const uint32_t total = group_last<64>(incl);
          lds_fence();
          st[lane] = incl - cnt;
          p0s[lane] = cur;
          avs[lane] = av;
          lds_fence();
          auto fetchq = [&](uint32_t q, uint32_t &tt, uint32_t &col, double &bv) {
            const bool valid = q < total;
            const uint32_t qq = valid ? q : 0u;
            tt = 0;
#pragma unroll
            for (uint32_t sft = 32; sft > 0; sft >>= 1)
              if (st[tt + sft] <= qq) tt += sft;
            const uint64_t p = p0s[tt] + (qq - st[tt]);
            col = EMPTY_KEY;
            bv = 0.0;
            if (valid) bpair(p, col, bv);
          };
          uint32_t t_q[SLICE_PF], col_q[SLICE_PF];
          double bv_q[SLICE_PF];
#pragma unroll
          for (int d = 0; d < SLICE_PF; ++d) {
            t_q[d] = 0; col_q[d] = EMPTY_KEY; bv_q[d] = 0.0;
            if (64u * d < total) fetchq(64u * d + lane, t_q[d], col_q[d], bv_q[d]);
          }
#ifdef SG_SLICE_STAMPS
          if (col_q[0] == 12345u && bv_q[0] == 0.123) nz++; // the loads have to land before the stamp
#endif
          SG_STAMP(9)
          // two steps per round: their position searches (five dependent LDS round trips each) are
          // independent and run side by side; the additions follow in step order
          static_assert(SLICE_PF >= 2, "two steps are taken from the queue per round");
          for (uint32_t q0 = 0; q0 < total; q0 += 128) {
            const uint32_t t_c = t_q[0], col_c = col_q[0], t_d = t_q[1], col_d = col_q[1];
            const double bv_c = bv_q[0], bv_d = bv_q[1];
#pragma unroll
            for (int d = 0; d + 2 < SLICE_PF; ++d) { t_q[d] = t_q[d + 2]; col_q[d] = col_q[d + 2]; bv_q[d] = bv_q[d + 2]; }
#pragma unroll
            for (int d = SLICE_PF - 2; d < SLICE_PF; ++d) {
              t_q[d] = 0; col_q[d] = EMPTY_KEY; bv_q[d] = 0.0;
              if (q0 + 64u * (uint32_t)(d + 2) < total) // flies during the next rounds
                fetchq(q0 + 64u * (uint32_t)(d + 2) + lane, t_q[d], col_q[d], bv_q[d]);
            }
            const bool valid_c = col_c != EMPTY_KEY, valid_d = col_d != EMPTY_KEY;
            const uint32_t pos_c = find(valid_c ? col_c : 0u), pos_d = find(valid_d ? col_d : 0u);
            claim_in_lane_order(vals, owner, stamp, lane, valid_c, pos_c, __dmul_rn(avs[t_c], bv_c), hw_order != 0);
            if (q0 + 64u < total) // wave-uniform
              claim_in_lane_order(vals, owner, stamp, lane, valid_d, pos_d, __dmul_rn(avs[t_d], bv_d), hw_order != 0);
          }
          cur += cnt;
          if (act) { // the next column is one of the three looked at, unless the piece was longer
            const uint32_t seen = cnt == 1u ? c1 : cnt == 2u ? c2 : c3;
            nextc = cur >= end ? EMPTY_KEY : cnt <= 3u ? seen : bcol(cur);
          }
          SG_STAMP(5)
        }
      };
      if (in_regs) {
        const bool act = cur < end && (uint64_t)nextc < chi;
        if (__any(act)) block(act, cur, end, av, nextc);
      } else {
        // The list entries that have something in this chunk are gathered, in list order, in a
        // queue (global scratch, written here and read back by other lanes after a fence), and
        // taken from it 64 at a time.
        uint32_t nq = 0;
        for (uint32_t lb = 0; lb < nlist; lb += 64 * U) {
          uint32_t nx[U];
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const uint32_t e = lb + 64u * u + lane;
            nx[u] = e < nlist ? l_next[e] : EMPTY_KEY;
          }
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const bool a = nx[u] != EMPTY_KEY && (uint64_t)nx[u] < chi; // chi is 2^32 in a row's last chunk
            const uint64_t mask = __ballot(a);
            if (a) l_q[nq + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = lb + 64u * u + lane;
            nq += (uint32_t)__popcll(mask);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        for (uint32_t qb = 0; qb < nq; qb += 64) {
          bool have = qb + lane < nq;
          uint32_t e = have ? l_q[qb + lane] : 0u;
          if (e >= nlist) { e = 0u; if (have) o.zero_count[3] = 1ull; have = false; }
          uint64_t ecur = have ? l_cur[e] : 0ull;
          const uint64_t eend = have ? l_end[e] : 0ull;
          const double eav = have ? l_av[e] : 0.0;
          uint32_t enext = have ? l_next[e] : EMPTY_KEY;
          if (have && !(ecur < eend && eend <= nnz_b && (uint64_t)enext < chi)) {
            o.zero_count[3] = 1ull; // an entry that cannot be: reported by the host, never followed
            have = false;
          }
          block(have, ecur, eend, eav, enext);
          if (have) { l_cur[e] = ecur; l_next[e] = enext; }
        }
        // the list is read back by other lanes in the next chunk: all its stores have to be done
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      }
      lds_fence();
      SG_STAMP(3)
      for (uint32_t p = lane; p < ncols; p += 64) {
        const double v = vals[p];
        o.c_col[off + pbase + p] = scol[p];
        o.c_val[off + pbase + p] = v;
        nz += (v == 0.0) ? 1u : 0u;
      }
      SG_STAMP(6)
    }
  }
#ifdef SG_SLICE_STAMPS
  if (lane == 0)
    for (int i = 0; i < 20; ++i) atomicAdd(&o.zero_count[8 + i], tph[i]);
#endif
  nz = group_sum<64>(nz);
  if (lane == 0 && nz) atomicAdd(o.zero_count, (unsigned long long)nz);
}
#undef SG_STAMP

} // namespace sg
