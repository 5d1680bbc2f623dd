// Device-side Russian checkers rules shared by the board slab (russian.hip) and the tree search
// (mcts_games.hip).  See russian.hip for the formulation notes and the reference citations.
#pragma once
#include "common.h"

namespace rudev {

#ifdef RU_PROF
// timing-only instrumentation (never in the product build): cycles by section, per lane; the playout kernel adds the
// per-wave maximum of each section to g_ru_prof
__device__ unsigned long long g_ru_prof[16];
static __shared__ unsigned long long s_ru_prof[16][256];
#define RPROF_T0 unsigned long long _t0 = clock64()
#define RPROF_ADD(k) do { unsigned long long _t1 = clock64(); s_ru_prof[k][threadIdx.x] += _t1 - _t0; _t0 = clock64(); } while (0)
#define RPROF_INC(k) do { s_ru_prof[k][threadIdx.x] += 1; } while (0)
#else
#define RPROF_T0
#define RPROF_ADD(k)
#define RPROF_INC(k)
#endif

constexpr int RU_ACTIONS = 280, RU_MAX_PLY = 250, RU_M_INVALID = 281;
constexpr uint64_t RU_DARK = 0x55AA55AA55AA55AAull;  // (y + x) odd, bit = y*8 + x
constexpr uint64_t RU_X0 = 0x0101010101010101ull;

struct Ru { uint64_t wp, wk, bp, bk, w4; };  // w4: see include/elfhip.h (state layout of ELFHIP_RUSSIAN)

__host__ __device__ constexpr bool ru_white(uint64_t w4) { return w4 & 1; }
__host__ __device__ constexpr int ru_ply(uint64_t w4) { return (int)((w4 >> 8) & 0xFFFF); }

// direction k in the order of the reference's action table (HashAllMoves.h):
//   0 = (dy-1,dx+1)  1 = (-1,-1)  2 = (+1,-1)  3 = (+1,+1);   inverse direction = k ^ 2
__host__ __device__ constexpr uint64_t ru_wrap(int k) {  // squares a one-step shift may legally land on
  return (k == 0 || k == 3) ? ~RU_X0 : ~(RU_X0 << 7);
}
__host__ __device__ constexpr uint64_t ru_shn(uint64_t b, int k, int steps) {  // raw shift, no wrap mask
  return k == 0 ? b >> (7 * steps) : k == 1 ? b >> (9 * steps) : k == 2 ? b << (7 * steps) : b << (9 * steps);
}
__host__ __device__ constexpr uint64_t ru_sh(uint64_t b, int k) { return ru_shn(b, k, 1) & ru_wrap(k); }

// from-squares whose destination d steps along k stays on the board (closed form)
__host__ __device__ constexpr uint64_t ru_valid(int k, int d) {
  uint64_t rows = (k < 2) ? (~0ull << (8 * d)) : (~0ull >> (8 * d));
  uint64_t cols = (k == 0 || k == 3) ? RU_X0 * (uint64_t)((1u << (8 - d)) - 1u) : RU_X0 * (uint64_t)((0xFFu << d) & 0xFFu);
  return RU_DARK & rows & cols;
}
__host__ __device__ constexpr int ru_popc(uint64_t v) { int c = 0; for (; v; v &= v - 1) ++c; return c; }
// first action index of group (k, d): groups are ordered k-major, d = 1..7
__host__ __device__ constexpr int ru_base(int k, int d) {
  int b = 0;
  for (int kk = 0; kk < 4; ++kk)
    for (int dd = 1; dd < 8; ++dd) {
      if (kk == k && dd == d) return b;
      b += ru_popc(ru_valid(kk, dd));
    }
  return b;
}
static_assert(ru_base(3, 7) + ru_popc(ru_valid(3, 7)) == RU_ACTIONS, "russian action table size");

// ---- the 32 dark squares: index i = y*4 + (x >> 1) = (square y*8 + x) >> 1 ------------------------------------
// All rule code below works on 32-bit sets of dark squares (half the instructions of the 64-bit board for every
// and / or / popcount / lowest-bit step).  Rows alternate: in even rows the dark squares are x = 1,3,5,7, in odd rows
// x = 0,2,4,6, so one diagonal step is i +- 4 for one row parity and i +- 3 / +- 5 for the other; two and four steps
// along a diagonal are the same shift for every row.
__host__ __device__ constexpr uint32_t ru_c32(uint64_t b) {   // compress (constant expressions / host)
  uint32_t r = 0;
  for (int i = 0; i < 32; ++i) { const int y = i >> 2, x = 2 * (i & 3) + ((y & 1) ? 0 : 1); r |= (uint32_t)((b >> (y * 8 + x)) & 1) << i; }
  return r;
}
__host__ __device__ constexpr uint64_t ru_e64(uint32_t c) {   // expand
  uint64_t r = 0;
  for (int i = 0; i < 32; ++i) { const int y = i >> 2, x = 2 * (i & 3) + ((y & 1) ? 0 : 1); r |= (uint64_t)((c >> i) & 1) << (y * 8 + x); }
  return r;
}
// the same on the device: bit tricks instead of 32 steps (a board is converted when it is loaded and stored only)
__device__ inline uint32_t ru_c16(uint32_t h) {   // four rows (one 32-bit half) -> 16 bits
  uint32_t t = ((h >> 1) & 0x00550055u) | (h & 0x55005500u);   // every row's dark squares at bits 0,2,4,6 of its byte
  t = (t | (t >> 1)) & 0x33333333u;
  t = (t | (t >> 2)) & 0x0F0F0F0Fu;
  t = (t | (t >> 4)) & 0x00FF00FFu;
  return (t | (t >> 8)) & 0xFFFFu;
}
__device__ inline uint32_t ru_compress(uint64_t b) { return ru_c16((uint32_t)b) | (ru_c16((uint32_t)(b >> 32)) << 16); }
__device__ inline uint32_t ru_e32(uint32_t c) {   // 16 bits -> four rows
  uint32_t t = (c | (c << 8)) & 0x00FF00FFu;
  t = (t | (t << 4)) & 0x0F0F0F0Fu;
  t = (t | (t << 2)) & 0x33333333u;
  t = (t | (t << 1)) & 0x55555555u;
  return ((t & 0x00550055u) << 1) | (t & 0x55005500u);
}
__device__ inline uint64_t ru_expand(uint32_t c) { return (uint64_t)ru_e32(c & 0xFFFFu) | ((uint64_t)ru_e32(c >> 16) << 32); }

// one, two and four steps along k (bits that would leave the board are dropped)
__host__ __device__ constexpr uint32_t r_sh1(uint32_t b, int k) {
  return k == 0 ? ((b & 0x07070707u) >> 3) | ((b & 0xF0F0F0F0u) >> 4)
       : k == 1 ? ((b & 0x0F0F0F0Fu) >> 4) | ((b & 0xE0E0E0E0u) >> 5)
       : k == 2 ? ((b & 0x0F0F0F0Fu) << 4) | ((b & 0xE0E0E0E0u) << 3)
                : ((b & 0x07070707u) << 5) | ((b & 0xF0F0F0F0u) << 4);
}
__host__ __device__ constexpr uint32_t r_sh2(uint32_t b, int k) {
  return k == 0 ? (b & 0x77777777u) >> 7 : k == 1 ? (b & 0xEEEEEEEEu) >> 9 : k == 2 ? (b & 0xEEEEEEEEu) << 7 : (b & 0x77777777u) << 9;
}
__host__ __device__ constexpr uint32_t r_sh4(uint32_t b, int k) {
  return k == 0 ? (b & 0x33333333u) >> 14 : k == 1 ? (b & 0xCCCCCCCCu) >> 18 : k == 2 ? (b & 0xCCCCCCCCu) << 14 : (b & 0x33333333u) << 18;
}
// the shifts agree with the 64-bit board for every square and direction
constexpr bool ru_check_shifts() {
  for (int k = 0; k < 4; ++k)
    for (int i = 0; i < 32; ++i) {
      const uint32_t b = 1u << i;
      if (r_sh1(b, k) != ru_c32(ru_sh(ru_e64(b), k))) return false;
      if (r_sh2(b, k) != r_sh1(r_sh1(b, k), k)) return false;
      if (r_sh4(b, k) != r_sh2(r_sh2(b, k), k)) return false;
    }
  return true;
}
static_assert(ru_check_shifts(), "russian 32-bit diagonal shifts");
static_assert(ru_c32(RU_DARK) == 0xFFFFFFFFu && ru_e64(0xFFFFFFFFu) == RU_DARK, "russian dark-square packing");
__host__ __device__ constexpr uint32_t r_valid(int k, int d) { return ru_c32(ru_valid(k, d)); }

// action -> from | to << 6 (64-bit squares) and the dark squares strictly between the two; generated by the same rule
// as the reference's table (descending from-square inside a group)
struct RuActTab { uint2 e[RU_ACTIONS]; };
constexpr RuActTab ru_make_acttab() {
  RuActTab t{};
  int n = 0;
  for (int k = 0; k < 4; ++k)
    for (int d = 1; d < 8; ++d) {
      uint64_t v = ru_valid(k, d);
      for (int s = 63; s >= 0; --s)
        if ((v >> s) & 1) {
          const int step = (k < 2) ? -(k == 0 ? 7 : 9) : (k == 2 ? 7 : 9);
          uint64_t between = 0;
          for (int i = 1; i < d; ++i) between |= 1ull << (s + step * i);
          t.e[n].x = (uint32_t)(s | ((s + step * d) << 6));
          t.e[n].y = ru_c32(between);
          ++n;
        }
    }
  return t;
}
static __constant__ RuActTab c_ru_act = ru_make_acttab();

struct R32 { uint32_t wp, wk, bp, bk; uint64_t w4; };
__device__ inline R32 ru_to32(const Ru& s) { return R32{ru_compress(s.wp), ru_compress(s.wk), ru_compress(s.bp), ru_compress(s.bk), s.w4}; }
__device__ inline Ru ru_to64(const R32& s) { return Ru{ru_expand(s.wp), ru_expand(s.wk), ru_expand(s.bp), ru_expand(s.bk), s.w4}; }

// Kogge-Stone occluded fill along k: everything reachable from `gen` through empty squares (gen included).  The
// propagators (empty squares whose predecessor(s) along k are empty too) depend on the position only: made once.
struct RPro { uint32_t e, p2[4], p4[4]; };
__device__ inline RPro r_pro(uint32_t empty) {
  RPro P; P.e = empty;
#pragma unroll
  for (int k = 0; k < 4; ++k) { P.p2[k] = empty & r_sh1(empty, k); P.p4[k] = P.p2[k] & r_sh2(P.p2[k], k); }
  return P;
}
__device__ inline uint32_t r_fill(uint32_t gen, const RPro& P, int k) {
  gen |= P.e & r_sh1(gen, k);
  gen |= P.p2[k] & r_sh2(gen, k);
  gen |= P.p4[k] & r_sh4(gen, k);
  return gen;
}

// The bits of one action group (k, d) - at most 28 consecutive actions from a compile-time base - are collected in a
// 32-bit word (bit = rank inside the group) and placed into the 280-bit mask with constant shifts: one instruction
// per emitted action instead of a five-way word select.
__device__ inline void ru_place_group(uint64_t m[5], uint32_t g, int k, int d) {
  const int base = ru_base(k, d), size = ru_popc(ru_valid(k, d));
  const int w0 = base >> 6, sh = base & 63;
  m[w0] |= (uint64_t)g << sh;
  if (sh + size > 64) m[w0 + 1] |= (uint64_t)g >> (64 - sh);
}
// all from-squares of `from` move d steps along k
__device__ inline void ru_emit_group(uint64_t m[5], uint32_t from, int k, int d) {
  const uint32_t valid = r_valid(k, d);
  uint32_t g = 0;
  while (from) {
    const uint32_t o = from & (0u - from);
    from ^= o;
    g |= 1u << __popc(valid & ~(o | (o - 1u)));
  }
  ru_place_group(m, g, k, d);
}
// one origin, a set of destinations along k
__device__ inline void ru_emit_ray(uint64_t m[5], uint32_t o, uint32_t dests, int k) {
  const uint32_t above = ~(o | (o - 1u));
  uint32_t t = o;
#pragma unroll
  for (int d = 1; d < 8; ++d) {
    t = r_sh1(t, k);
    ru_place_group(m, (dests & t) ? 1u << __popc(r_valid(k, d) & above) : 0u, k, d);
  }
}

__device__ inline void ru_clear(Ru& s) {  // ClearBoard CheckersBoard.cc:10-33
  s.wp = RU_DARK & 0xFFFFFFull; s.wk = 0; s.bp = RU_DARK & (0xFFFFFFull << 40); s.bk = 0;
  s.w4 = (1ull << 8) | ((uint64_t)RU_M_INVALID << 24);
}
__device__ inline void ru_clear(R32& s) {
  s.wp = 0x00000FFFu; s.wk = 0; s.bp = 0xFFF00000u; s.bk = 0;
  s.w4 = (1ull << 8) | ((uint64_t)RU_M_INVALID << 24);
}
static_assert(ru_c32(RU_DARK & 0xFFFFFFull) == 0x00000FFFu && ru_c32(RU_DARK & (0xFFFFFFull << 40)) == 0xFFF00000u, "russian start position");

// getAllMoves CheckersBoard.cc:259-273 -> 280-bit legal mask
__device__ inline void ru_moves(const R32& s, uint64_t m[5]) {
#pragma unroll
  for (int i = 0; i < 5; ++i) m[i] = 0;
  const bool white = ru_white(s.w4);
  uint32_t P = white ? s.wp : s.bp, K = white ? s.wk : s.bk;
  const uint32_t enemy = white ? (s.bp | s.bk) : (s.wp | s.wk);
  const uint32_t empty = ~(s.wp | s.wk | s.bp | s.bk);
  const bool has_next = (s.w4 >> 1) & 1;
  if (has_next) {  // only the piece that just captured may move, and only by capturing (:262-263)
    const uint32_t nb = 1u << (((((uint32_t)(s.w4 >> 2) & 7u) << 3) | ((uint32_t)(s.w4 >> 5) & 7u)) >> 1);  // next_y<<2 | next_x<<5
    K = nb & (s.wk | s.bk);
    P = nb & ~(s.wk | s.bk);
  }
  bool any = false;
  RPROF_T0;
  // _pawnJumpInDirection :593-628 - all pawns at once
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint32_t land = r_sh1(r_sh1(P, k) & enemy, k) & empty;
    const uint32_t from = r_sh1(r_sh1(land, k ^ 2), k ^ 2);
    any |= from != 0;
    ru_emit_group(m, from, k, 2);
  }
  RPROF_ADD(0);
  // _kingJumpInDirection :457-530, by captured piece instead of by king.  A ray stops at any piece, own kings
  // included, so the occluded fill from the whole king set is the union of the kings' rays: "the first piece on some
  // king's ray along k is an enemy with an empty square behind it" is one fill per direction for all kings, and every
  // such piece belongs to exactly one king (found by walking back from it).
  if (__any(K != 0)) {
    const RPro E = r_pro(empty);
    uint32_t capl[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t cap = r_sh1(r_fill(K, E, k), k) & enemy;
      capl[k] = cap & r_sh1(r_sh1(cap, k) & empty, k ^ 2);
    }
    RPROF_ADD(1);
    if (__any((capl[0] | capl[1] | capl[2] | capl[3]) != 0)) {
      // F0[d] = squares from which a king sees, along d, empty squares, an enemy, an empty square (_kingJumpCheck
      // :412-453) on the board as it stands.  The reference tests the landing squares with the captured piece lifted
      // and the king still on its origin (:488): for a landing square behind the captured piece that changes nothing
      // along k and along the two crossing diagonals (neither contains the lifted square), and looking back along
      // k ^ 2 the first piece behind the lifted one is the capturing king itself - never a capture.  So the test is
      // the three other directions on the unchanged board, shared by every capture of the position.
      uint32_t F0[4];
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const uint32_t T = enemy & r_sh1(empty, d ^ 2);
        F0[d] = r_fill(r_sh1(T, d ^ 2) & empty, E, d ^ 2);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t C = F0[k] | F0[k ^ 1] | F0[k ^ 3];
        for (uint32_t cc = capl[k]; cc;) {
          const uint32_t c = cc & (0u - cc);
          cc ^= c;
          RPROF_INC(9);
          uint32_t L = r_fill(r_sh1(c, k), E, k);                // landing squares behind the captured piece
          const uint32_t Lc = L & C;
          if (Lc) L = (k >= 2) ? (Lc & (0u - Lc)) : (0x80000000u >> __clz((int)Lc));  // nearest one only (:505-513)
          const uint32_t t = r_sh1(c, k ^ 2);                    // the square in front of it: the king, or empty
          const uint32_t o = K & (t | r_sh1(r_fill(t & empty, E, k ^ 2), k ^ 2));
          ru_emit_ray(m, o, L, k);
          any = true;
        }
      }
    }
  }
  RPROF_ADD(2);
  if (any || has_next) return;
  // simple moves: pawns forward only (:317-346), kings slide (:348-404).  Forward is k = 2, 3 for white and k = 0, 1
  // for black; a wave holds boards of both colours to move (forced continuations shift the parity), so the two
  // forward directions are walked once with per-lane direction masks instead of all four directions in turn.
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const uint32_t from = P & (white ? r_sh1(empty, j) : r_sh1(empty, 2 + j));   // next square along k = (2 + j | j) empty
    const uint32_t valid = white ? r_valid(2 + j, 1) : r_valid(j, 1);
    uint32_t g = 0;
    for (uint32_t f = from; f;) {
      const uint32_t o = f & (0u - f);
      f ^= o;
      g |= 1u << __popc(valid & ~(o | (o - 1u)));
    }
    ru_place_group(m, white ? g : 0u, 2 + j, 1);
    ru_place_group(m, white ? 0u : g, j, 1);
  }
  RPROF_ADD(3);
  // King slides by action group instead of by king: R = squares whose next d squares along k are all empty, so
  // K & R are exactly the from-squares of group (k, d).  A wave runs 28 short group steps (fewer: a direction ends
  // at the first distance no lane's king can go) instead of four rays of seven tests for each king of its
  // king-richest board.
  if (__any(K != 0)) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      uint32_t R = 0xFFFFFFFFu;
#pragma unroll
      for (int d = 1; d < 8; ++d) {
        R = r_sh1(empty & R, k ^ 2);
        const uint32_t from = K & R;
        if (!__any(from != 0)) break;
        RPROF_INC(10);
        ru_emit_group(m, from, k, d);
      }
    }
  }
  RPROF_ADD(4);
}
__device__ inline void ru_moves(const Ru& s, uint64_t m[5]) { ru_moves(ru_to32(s), m); }

__device__ inline int ru_count(const uint64_t m[5]) {
  return __popcll(m[0]) + __popcll(m[1]) + __popcll(m[2]) + __popcll(m[3]) + __popcll(m[4]);
}

// CheckersPlay :36-82 (action must be legal)
// tab: the action table (c_ru_act.e); a kernel that plays many moves keeps a copy in LDS
template <class TAB>
__device__ inline void ru_play(R32& s, int action, TAB tab) {
  const uint2 e = tab[action];
  const int fs = e.x & 63, ts = e.x >> 6;
  const uint32_t fb = 1u << (fs >> 1), tb = 1u << (ts >> 1);
  const bool is_king = (fb & (s.wk | s.bk)) != 0;
  const bool white_piece = (fb & (s.wp | s.wk)) != 0;
  const uint32_t cap = e.y & (s.wp | s.wk | s.bp | s.bk);   // the piece between origin and destination, if any
  uint64_t w4 = s.w4;
  if (cap) {
    s.wp &= ~cap; s.wk &= ~cap; s.bp &= ~cap; s.bk &= ~cap;
    // continuation is evaluated from the destination while the piece still stands on its origin (:58-66)
    const bool white = ru_white(w4);
    const uint32_t enemy = white ? (s.bp | s.bk) : (s.wp | s.wk);
    const uint32_t empty = ~(s.wp | s.wk | s.bp | s.bk);
    bool more = false;
    if (is_king) {
      const RPro E = r_pro(empty);
#pragma unroll
      for (int k = 0; k < 4; ++k) more |= (r_sh1(r_sh1(r_fill(tb, E, k), k) & enemy, k) & empty) != 0;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) more |= (r_sh1(r_sh1(tb, k) & enemy, k) & empty) != 0;
    }
    w4 &= ~0xFEull;
    if (more) w4 |= 2ull | ((uint64_t)(ts >> 3) << 2) | ((uint64_t)(ts & 7) << 5);
  }
  // move the piece, promotion (_tryConvertIntoKing :275-280)
  if (white_piece) {
    if (is_king) s.wk = (s.wk & ~fb) | tb; else { s.wp &= ~fb; if (ts >= 56) s.wk |= tb; else s.wp |= tb; }
  } else {
    if (is_king) s.bk = (s.bk & ~fb) | tb; else { s.bp &= ~fb; if (ts < 8) s.bk |= tb; else s.bp |= tb; }
  }
  if (!((w4 >> 1) & 1)) w4 ^= 1;
  const uint64_t ply = (uint64_t)((ru_ply(w4) + 1) & 0xFFFF);
  w4 = (w4 & 0xFFull) | (ply << 8) | ((uint64_t)action << 24);
  s.w4 = w4;
}
__device__ inline void ru_play(Ru& s, int action) { R32 r = ru_to32(s); ru_play(r, action, c_ru_act.e); s = ru_to64(r); }

__device__ inline void ru_load(const uint64_t* __restrict__ w, int n, int b, Ru& s) {
  s.wp = w[0 * (size_t)n + b]; s.wk = w[1 * (size_t)n + b]; s.bp = w[2 * (size_t)n + b];
  s.bk = w[3 * (size_t)n + b]; s.w4 = w[4 * (size_t)n + b];
}
__device__ inline void ru_store(uint64_t* __restrict__ w, int n, int b, const Ru& s) {
  w[0 * (size_t)n + b] = s.wp; w[1 * (size_t)n + b] = s.wk; w[2 * (size_t)n + b] = s.bp;
  w[3 * (size_t)n + b] = s.bk; w[4 * (size_t)n + b] = s.w4;
}

// k-th set bit (0-based, ascending) of a 5-word mask
__device__ inline int ru_kth(const uint64_t m[5], int kk) {
  int w = 0; uint64_t sel = 0; bool found = false;
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    int c = __popcll(m[q]);
    bool here = !found && kk < c;
    if (here) { w = q; sel = m[q]; found = true; }
    if (!found) kk -= c;
  }
  return w * 64 + (int)elf_kth_bit64(sel, kk);
}

}  // namespace rudev
