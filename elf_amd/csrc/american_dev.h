// Device-side rules of American checkers shared by the board-slab kernels (american.hip) and the
// tree-search kernels (mcts.hip).  See american.hip for the formulation notes.
#pragma once
#include "common.h"

namespace amdev {


constexpr int AM_ACTIONS = 170;
constexpr int AM_MAX_PLY = 250;       // GameBoard.h:46
constexpr int AM_REPEAT = 4;          // GameBoard.h:48
constexpr uint64_t AM_UNUSED = (1ull << 8) | (1ull << 17) | (1ull << 26) | (1ull << 35);  // GameBoard.h:60
constexpr uint64_t AM_PLAYABLE = ((1ull << 36) - 1) ^ AM_UNUSED;

// move families in action-index order (HashAllMoves.h:193-363):
//  0: 0x11<<i dir1 (right forward)   1: 0x21<<i dir1 (left forward)
//  2: 0x11<<i dir0 (right backward)  3: 0x21<<i dir0 (left backward)
//  4: -(0x101<<i) dir1  5: -(0x401<<i) dir1  6: -(0x101<<i) dir0  7: -(0x401<<i) dir0
constexpr uint64_t valid_shifts(int span, bool jump) {
  uint64_t v = 0;
  for (int i = 0; i < 36; ++i) {
    bool ok = ((AM_PLAYABLE >> i) & 1) && (i + span < 36) && ((AM_PLAYABLE >> (i + span)) & 1);
    if (jump) ok = ok && ((AM_PLAYABLE >> (i + span / 2)) & 1);
    if (ok) v |= 1ull << i;
  }
  return v;
}
constexpr uint64_t V4 = valid_shifts(4, false), V5 = valid_shifts(5, false);
constexpr uint64_t V8 = valid_shifts(8, true), V10 = valid_shifts(10, true);
constexpr int popc_c(uint64_t v) { int c = 0; for (; v; v &= v - 1) ++c; return c; }
constexpr int N4 = popc_c(V4), N5 = popc_c(V5), N8 = popc_c(V8), N10 = popc_c(V10);
static_assert(N4 == 25 && N5 == 24 && N8 == 18 && N10 == 18, "action table layout");
static_assert(2 * (N4 + N5 + N8 + N10) == AM_ACTIONS, "170 actions");

struct FamInfo { uint64_t valid; int span; int base; int dir; };
__device__ __host__ constexpr FamInfo fam_info(int t) {
  switch (t) {
    case 0: return {V4, 4, 0, 1};
    case 1: return {V5, 5, N4, 1};
    case 2: return {V4, 4, N4 + N5, 0};
    case 3: return {V5, 5, 2 * N4 + N5, 0};
    case 4: return {V8, 8, 2 * (N4 + N5), 1};
    case 5: return {V10, 10, 2 * (N4 + N5) + N8, 1};
    case 6: return {V8, 8, 2 * (N4 + N5) + N8 + N10, 0};
    default: return {V10, 10, 2 * (N4 + N5) + 2 * N8 + N10, 0};
  }
}

// action -> (family, shift) table, built at compile time, resident in constant memory
struct ActTab { uint8_t fam[AM_ACTIONS + 2]; uint8_t shift[AM_ACTIONS + 2]; };
constexpr ActTab make_acttab() {
  ActTab t{};
  for (int f = 0; f < 8; ++f) {
    FamInfo fi = fam_info(f);
    int r = 0;
    for (int i = 0; i < 36; ++i)
      if ((fi.valid >> i) & 1) { t.fam[fi.base + r] = (uint8_t)f; t.shift[fi.base + r] = (uint8_t)i; ++r; }
  }
  return t;
}
static __constant__ ActTab c_acttab = make_acttab();

struct Am {
  uint64_t f[2], k[2];  // forward / backward per colour
  uint64_t w4, w5;      // packed scalars, layout in oracle/american.h
};

__device__ inline int am_active(const Am& s) { return (int)(s.w4 & 1); }
__device__ inline int am_ply(const Am& s) { return (int)((s.w4 >> 8) & 0xFFFF); }

__device__ inline void am_clear(Am& s) {  // GameBoard.cc:10-35
  s.f[0] = 0x1eff; s.k[0] = 0; s.f[1] = 0; s.k[1] = 0x7fbc00000ull;
  s.w4 = (1ull << 8) | (171ull << 24);
  s.w5 = 0xFFFFFFFFull;
}

// The eight family bitboards, normalised so that bit i of fam[t] <=> move word W_t << i.
// GameBoard.cc:321-351 (shift rules), :359-463 (jumps are mandatory).
__device__ inline void am_families(const Am& s, uint64_t fam[8]) {
  int a = am_active(s);
  uint64_t pa = s.f[a] | s.k[a], pp = s.f[a ^ 1] | s.k[a ^ 1];
  uint64_t e = AM_PLAYABLE ^ (pa | pp);
  uint64_t f = s.f[a], k = s.k[a];
  uint64_t rfj = (e >> 8) & (pp >> 4) & f, lfj = (e >> 10) & (pp >> 5) & f;
  uint64_t rbj = (e << 8) & (pp << 4) & k, lbj = (e << 10) & (pp << 5) & k;
  bool anyj = (rfj | lfj | rbj | lbj) != 0;
  fam[4] = rfj; fam[5] = lfj; fam[6] = rbj >> 8; fam[7] = lbj >> 10;
  fam[0] = anyj ? 0 : ((e >> 4) & f);
  fam[1] = anyj ? 0 : ((e >> 5) & f);
  fam[2] = anyj ? 0 : (((e << 4) & k) >> 4);
  fam[3] = anyj ? 0 : (((e << 5) & k) >> 5);
}

__device__ inline int am_count(const uint64_t fam[8]) {
  int c = 0;
#pragma unroll
  for (int t = 0; t < 8; ++t) c += __popcll(fam[t]);
  return c;
}

// repeat rule GameBoard.cc:151-159: with more than one legal move, a side that repeated >= 4
// times may not play the move it played two turns ago.
__device__ inline int am_apply_repeat_rule(const Am& s, uint64_t fam[8], int total) {
  int a = am_active(s);
  int rep = (int)((s.w4 >> (32 + 8 * a)) & 0xFF);
  if (total > 1 && rep >= AM_REPEAT) {
    int banned = (int)((s.w5 >> (16 * a + 8)) & 0xFF);  // last_move_<side>[1]
    if (banned < AM_ACTIONS) {
      int bt = c_acttab.fam[banned], bi = c_acttab.shift[banned];
      int removed = 0;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        uint64_t bit = (t == bt) ? (1ull << bi) : 0;
        removed += (fam[t] & bit) != 0;
        fam[t] &= ~bit;
      }
      total -= removed;
    }
  }
  return total;
}

// compress the bits of x selected by the compile-time mask V to the low end (software pext over
// the mask's contiguous runs; V has 7-8 runs)
template <uint64_t V>
__device__ inline uint64_t compress_bits(uint64_t x) {
  uint64_t out = 0;
  int pos = 0;
#pragma unroll
  for (int s = 0; s < 36; ++s) {
    bool start = ((V >> s) & 1) && (s == 0 || !((V >> (s - 1)) & 1));
    if (start) {
      int len = 0;
      while (s + len < 36 && ((V >> (s + len)) & 1)) ++len;
      out |= ((x >> s) & ((1ull << len) - 1)) << pos;
      pos += len;
    }
  }
  return out;
}

// 170-bit action mask from the family bitboards
__device__ inline void am_mask_words(const uint64_t fam[8], uint64_t m[3]) {
  uint64_t c[8];
  c[0] = compress_bits<V4>(fam[0]); c[1] = compress_bits<V5>(fam[1]);
  c[2] = compress_bits<V4>(fam[2]); c[3] = compress_bits<V5>(fam[3]);
  c[4] = compress_bits<V8>(fam[4]); c[5] = compress_bits<V10>(fam[5]);
  c[6] = compress_bits<V8>(fam[6]); c[7] = compress_bits<V10>(fam[7]);
  m[0] = m[1] = m[2] = 0;
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    int base = fam_info(t).base;
    int w = base >> 6, o = base & 63;
    m[w] |= c[t] << o;
    if (o != 0 && w + 1 < 3) m[w + 1] |= c[t] >> (64 - o);
  }
}

__device__ inline bool am_has_jump_from(const Am& s, int active, uint64_t piece) {  // GameBoard.cc:465-528
  uint64_t pa = s.f[active] | s.k[active], pp = s.f[active ^ 1] | s.k[active ^ 1];
  uint64_t e = AM_PLAYABLE ^ (pa | pp);
  uint64_t fj = ((e >> 8) & (pp >> 4) & piece) | ((e >> 10) & (pp >> 5) & piece);
  uint64_t bj = ((e << 8) & (pp << 4) & piece) | ((e << 10) & (pp << 5) & piece);
  // black always jumps forward, backward only as a king (piece in backward[BLACK]); white mirrored
  bool king = active == 0 ? (piece & s.k[0]) != 0 : (piece & s.f[1]) != 0;
  uint64_t own = active == 0 ? fj : bj, other = active == 0 ? bj : fj;
  return (own | (king ? other : 0)) != 0;
}

// CheckersPlay GameBoard.cc:37-129 for the move (family t, shift i) whose action index is `action`
__device__ inline void am_play(Am& s, int action, int t, int i) {
  const int a = am_active(s), p = a ^ 1;
  const bool jump_move = t >= 4;
  const int span = (t & 1) ? (jump_move ? 10 : 5) : (jump_move ? 8 : 4);
  const uint64_t wbits = (1ull | (1ull << span)) << i;  // the two-bit move word
  const int dir = (t & 2) ? 0 : 1;
  // repeat bookkeeping :59-79
  int lm0 = (int)((s.w5 >> (16 * a)) & 0xFF), lm1 = (int)((s.w5 >> (16 * a + 8)) & 0xFF);
  int rep = (int)((s.w4 >> (32 + 8 * a)) & 0xFF);
  rep = (lm1 == action) ? rep + 1 : 0;
  lm1 = lm0; lm0 = action;
  s.w5 = (s.w5 & ~(0xFFFFull << (16 * a))) | ((uint64_t)lm0 << (16 * a)) | ((uint64_t)lm1 << (16 * a + 8));
  int ply = am_ply(s) + 1;
  int jumpflag = (int)((s.w4 >> 1) & 1) | (jump_move ? 1 : 0);
  if (jump_move) {  // :85-100
    uint64_t taken = 1ull << (i + span / 2);
    s.f[p] &= ~taken;
    s.k[p] &= ~taken;
  }
  if (s.f[a] & wbits) s.f[a] ^= wbits;  // :102-106
  if (s.k[a] & wbits) s.k[a] ^= wbits;
  const uint64_t dest = dir ? (1ull << (i + span)) : (1ull << i);
  uint64_t repw = (s.w4 & ~(0xFFull << (32 + 8 * a))) | ((uint64_t)(rep & 0xFF) << (32 + 8 * a));
  repw &= 0x0000FFFF00000000ull;
  bool cont = jumpflag && am_has_jump_from(s, a, dest);  // :111-116 (Q1, Q3)
  int nactive = a;
  if (!cont) {
    if (a == 0 && (dest & 0x780000000ull)) s.k[0] |= dest;  // crown :118-121
    else if (a == 1 && (dest & 0xfull)) s.f[1] |= dest;
    jumpflag = 0;
    nactive = p;
  }
  s.w4 = (uint64_t)nactive | ((uint64_t)jumpflag << 1) | ((uint64_t)ply << 8) | ((uint64_t)action << 24) | repw;
}

// Feature planes of ONE history slot for the square owned by `lane` (lane = y*8+x).
// BoardFeature::extract BoardFeature.cc:58-76 + GetObservation GameBoard.cc:210-277.
// pl[0..3] = mover pawns / mover kings / opponent pawns / opponent kings, each in that player's
// own orientation (Q7); pl[4], pl[5] = black / white to move.
__device__ inline void am_slot_planes(uint64_t f0, uint64_t f1, uint64_t k0, uint64_t k1, int act, int lane, float pl[6]) {
  const int y = lane >> 3, x = lane & 7;
  // square -> padded bit index in black's orientation (inverse of GameBoard.cc:230-232)
  auto sq2bit = [](int yy, int xx) -> int {
    int rr = 7 - yy;
    int t = 6 + (rr & 1) - xx;
    if (t < 0 || t > 6 || (t & 1)) return -1;
    int buff = rr * 4 + (t >> 1);
    return buff + (buff >> 3);
  };
  const int bit_b = sq2bit(y, x), bit_w = sq2bit(7 - y, 7 - x);  // white's view = 180 degree rotation
  // a king sits in both words of its colour; GetObservation tests kings first
  uint64_t pawn[2] = {f0 & ~k0, k1 & ~f1}, king[2] = {k0, f1};
  const int ba = act == 0 ? bit_b : bit_w, bp = act == 0 ? bit_w : bit_b;
  pl[0] = pl[1] = pl[2] = pl[3] = 0.f;
  if (ba >= 0) { pl[0] = (float)((pawn[act] >> ba) & 1); pl[1] = (float)((king[act] >> ba) & 1); }
  if (bp >= 0) { pl[2] = (float)((pawn[act ^ 1] >> bp) & 1); pl[3] = (float)((king[act ^ 1] >> bp) & 1); }
  pl[4] = act == 0 ? 1.f : 0.f;
  pl[5] = act == 1 ? 1.f : 0.f;
}

}  // namespace amdev
