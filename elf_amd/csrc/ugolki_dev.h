// Device-side Ugolki rules shared by the board slab (ugolki.hip) and the tree search (mcts_games.hip).
// See ugolki.hip for the formulation notes and the reference citations.
#pragma once
#include "common.h"

namespace ugdev {

constexpr int UG_PASS = 416, UG_MAX_PLY = 500, UG_PLANES = 16;
constexpr uint64_t UG_WHITE_BASE = 0xE0E0E00000000000ull, UG_BLACK_BASE = 0x0000000000070707ull;
constexpr uint64_t UG_X0 = 0x0101010101010101ull;
constexpr uint64_t UG_NOT_X7 = ~(UG_X0 << 7), UG_NOT_X67 = ~((UG_X0 << 7) | (UG_X0 << 6));
constexpr uint64_t UG_NOT_X0 = ~UG_X0, UG_NOT_X01 = ~(UG_X0 | (UG_X0 << 1));

struct Ug { uint64_t p[2], ja, meta; };  // meta = active | black_win<<1 | white_win<<4 | ply<<8

__device__ inline int ug_active(const Ug& s) { return (int)(s.meta & 1); }
__device__ inline int ug_ply(const Ug& s) { return (int)((s.meta >> 8) & 0xFFFF); }
__device__ inline int ug_bwin(const Ug& s) { return (int)((s.meta >> 1) & 7); }
__device__ inline int ug_wwin(const Ug& s) { return (int)((s.meta >> 4) & 7); }
__device__ inline bool ug_over(const Ug& s) {  // IsOver :215-222
  int bw = ug_bwin(s), ww = ug_wwin(s);
  return (bw > 0 && ww > 0) || bw == 2 || ww == 2;
}

__device__ inline void ug_clear(Ug& s) {  // ClearBoard :27-48 (black starts on WHITE_BASE and vice versa)
  s.p[0] = UG_WHITE_BASE; s.p[1] = UG_BLACK_BASE; s.ja = 0; s.meta = 1ull << 8;
}

// family base indices in the action table: right, left, forward, backward steps (56 each),
// right, left, forward, backward jumps (48 each), pass
__host__ __device__ constexpr int ug_base(int g) { return g < 4 ? 56 * g : 224 + 48 * (g - 4); }

// the four jump families of `pieces` over `all2` (occupied squares incl. the square just vacated), :331-352
__device__ inline void ug_jumps(uint64_t pieces, uint64_t all2, uint64_t fam[8]) {
  const uint64_t emp2 = ~all2;
  fam[4] = pieces & (emp2 >> 2) & (all2 >> 1) & UG_NOT_X67;                 // i -> i+2, low = i
  fam[5] = ((pieces & UG_NOT_X01) >> 2) & emp2 & (all2 >> 1);                // i -> i-2, low = i-2
  fam[6] = pieces & (emp2 >> 16) & (all2 >> 8);                              // i -> i+16
  fam[7] = (pieces >> 16) & emp2 & (all2 >> 8);                              // i -> i-16, low = i-16
}

// get_legal_moves :367-384 as eight bitboards (bit = low square of the pattern) + pass flag
__device__ inline bool ug_families(const Ug& s, uint64_t fam[8]) {
  const int a = ug_active(s);
  const uint64_t all = s.p[0] | s.p[1];
  if (s.ja != 0) {  // a jump chain is running: only the jumping man, only jumps, or pass (_jumps_from :499-567)
    const uint64_t pawn = s.p[a] & s.ja;
    fam[0] = fam[1] = fam[2] = fam[3] = 0;
    ug_jumps(pawn, all | (s.ja ^ pawn), fam);
    return (fam[4] | fam[5] | fam[6] | fam[7]) != 0;
  }
  // a side must first clear its own starting corner once the opponent has left theirs (:374-383)
  uint64_t pieces = s.p[a];
  if (a == 0 && !(s.p[1] & UG_BLACK_BASE) && (s.p[0] & UG_WHITE_BASE)) pieces = s.p[0] & UG_WHITE_BASE;
  if (a == 1 && !(s.p[0] & UG_WHITE_BASE) && (s.p[1] & UG_BLACK_BASE)) pieces = s.p[1] & UG_BLACK_BASE;
  const uint64_t empty = ~all;
  fam[0] = pieces & (empty >> 1) & UG_NOT_X7;                // :313-317
  fam[1] = ((pieces & UG_NOT_X0) >> 1) & empty;              // :318-322
  fam[2] = pieces & (empty >> 8);                            // :323-326
  fam[3] = (pieces >> 8) & empty;                            // :327-330
  ug_jumps(pieces, all, fam);
  return false;
}

__device__ inline uint64_t ug_compress7(uint64_t b) {  // 8 squares/row -> 7 (x < 7)
  uint64_t r = 0;
#pragma unroll
  for (int y = 0; y < 8; ++y) r |= ((b >> (8 * y)) & 0x7Full) << (7 * y);
  return r;
}
__device__ inline uint64_t ug_compress6(uint64_t b) {  // 8 squares/row -> 6 (x < 6)
  uint64_t r = 0;
#pragma unroll
  for (int y = 0; y < 8; ++y) r |= ((b >> (8 * y)) & 0x3Full) << (6 * y);
  return r;
}

template <int OFF>
__device__ inline void ug_put(uint64_t m[7], uint64_t field) {
  m[OFF >> 6] |= field << (OFF & 63);
  if ((OFF & 63) != 0 && (OFF >> 6) + 1 < 7) m[(OFF >> 6) + 1] |= field >> (64 - (OFF & 63));
}

__device__ inline void ug_mask_words(const uint64_t fam[8], bool pass, uint64_t m[7]) {
#pragma unroll
  for (int i = 0; i < 7; ++i) m[i] = 0;
  ug_put<0>(m, ug_compress7(fam[0]));
  ug_put<56>(m, ug_compress7(fam[1]));
  ug_put<112>(m, fam[2]);
  ug_put<168>(m, fam[3]);
  ug_put<224>(m, ug_compress6(fam[4]));
  ug_put<272>(m, ug_compress6(fam[5]));
  ug_put<320>(m, fam[6]);
  ug_put<368>(m, fam[7]);
  if (pass) m[6] |= 1ull << (UG_PASS & 63);
}

__device__ inline int ug_count(const uint64_t fam[8], bool pass) {
  int c = pass ? 1 : 0;
#pragma unroll
  for (int g = 0; g < 8; ++g) c += __popcll(fam[g]);
  return c;
}

__device__ inline int ug_action_of(int g, int i) {  // family + low square -> action index
  int rank = (g < 2) ? i - (i >> 3) : (g == 4 || g == 5) ? i - 2 * (i >> 3) : i;
  return ug_base(g) + rank;
}
__device__ inline void ug_decode(int action, int& g, int& i) {  // action (< 416) -> family + low square
  g = action < 224 ? action / 56 : 4 + (action - 224) / 48;
  int r = action - ug_base(g);
  i = (g < 2) ? r + r / 7 : (g == 4 || g == 5) ? r + 2 * (r / 6) : r;
}

// Play :51-109.  pass: g = -1
__device__ inline void ug_play(Ug& s, int g, int i) {
  int a = ug_active(s), bw = ug_bwin(s), ww = ug_wwin(s);
  const int ply = ug_ply(s) + 1;
  uint64_t action = 0;
  const bool jump = g >= 4;
  if (g >= 0) {
    const uint64_t pat = (g < 2) ? 0x3ull : (g < 4) ? 0x101ull : (g < 6) ? 0x5ull : 0x10001ull;
    action = pat << i;
  }
  if (jump) s.ja = action;
  s.p[0] ^= (a == 0) ? action : 0ull;
  s.p[1] ^= (a == 1) ? action : 0ull;
  if ((s.p[0] & UG_BLACK_BASE) == UG_BLACK_BASE) ++bw;           // :85-92 (counted on every Play, pass included)
  else if ((s.p[1] & UG_WHITE_BASE) == UG_WHITE_BASE) ++ww;
  else { bw = 0; ww = 0; }
  bool stay = false;
  if (jump) {  // the man may go on jumping; the square it came from stays blocked (:95-101)
    const uint64_t mine = (a == 0) ? s.p[0] : s.p[1];
    const uint64_t pawn = mine & s.ja;
    uint64_t fam[8];
    ug_jumps(pawn, s.p[0] | s.p[1] | (s.ja ^ pawn), fam);
    stay = (fam[4] | fam[5] | fam[6] | fam[7]) != 0;
  }
  if (!stay) { s.ja = 0; a ^= 1; }
  s.meta = (uint64_t)a | ((uint64_t)(bw & 7) << 1) | ((uint64_t)(ww & 7) << 4) | ((uint64_t)(ply & 0xFFFF) << 8);
}

__device__ inline void ug_load(const uint64_t* __restrict__ w, int n, int b, Ug& s) {
  s.p[0] = w[0 * (size_t)n + b]; s.p[1] = w[1 * (size_t)n + b]; s.ja = w[2 * (size_t)n + b]; s.meta = w[3 * (size_t)n + b];
}
__device__ inline void ug_store(uint64_t* __restrict__ w, int n, int b, const Ug& s) {
  w[0 * (size_t)n + b] = s.p[0]; w[1 * (size_t)n + b] = s.p[1]; w[2 * (size_t)n + b] = s.ja; w[3 * (size_t)n + b] = s.meta;
}

// ---- kernels ------------------------------------------------------------------------------------
}  // namespace ugdev
