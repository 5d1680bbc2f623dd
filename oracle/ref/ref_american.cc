// TEST INFRASTRUCTURE ONLY (oracle). Never linked or loaded by the product path.
//
// Thin extern "C" harness around the UNMODIFIED reference American-checkers board
// (/root/reference/src_cpp/elfgames/american_checkers/game/GameBoard.{h,cc}).  It is compiled by
// oracle/Makefile together with the reference .cc file where it lies; nothing of the reference is
// copied into this repository.  The harness only converts between the reference's GameBoard struct
// and the flat "am_state" words our restatement and HIP kernels use, so tests can compare them.
//
// The reference headers rely on transitive includes, hence the std headers first.
#include <array>
#include <string>
#include <sstream>
#include <cstring>
#include <cstdint>
#include "elfgames/american_checkers/game/GameBoard.h"

extern "C" {

// Flat state: 6 x u64, identical to oracle/american.h am_state and the device SoA words.
//  w[0..1] forward[2], w[2..3] backward[2],
//  w[4] = active | jump<<1 | ply<<8 | last_move<<24 | rep_b<<32 | rep_w<<40
//  w[5] = lm_b0 | lm_b1<<8 | lm_w0<<16 | lm_w1<<24      (0xFF = -1)
static void pack(const GameBoard& b, uint64_t* w) {
  w[0] = (uint64_t)b.forward[0];
  w[1] = (uint64_t)b.forward[1];
  w[2] = (uint64_t)b.backward[0];
  w[3] = (uint64_t)b.backward[1];
  w[4] = (uint64_t)(b.active & 1) | ((uint64_t)(b.jump & 1) << 1) |
         ((uint64_t)(b._ply & 0xFFFF) << 8) | ((uint64_t)(b._last_move & 0xFF) << 24) |
         ((uint64_t)(b._black_repeats_step & 0xFF) << 32) |
         ((uint64_t)(b._white_repeats_step & 0xFF) << 40);
  w[5] = (uint64_t)(b._last_move_black[0] & 0xFF) | ((uint64_t)(b._last_move_black[1] & 0xFF) << 8) |
         ((uint64_t)(b._last_move_white[0] & 0xFF) << 16) |
         ((uint64_t)(b._last_move_white[1] & 0xFF) << 24);
}

static int64_t sx8(uint64_t v) { return (v & 0xFF) == 0xFF ? -1 : (int64_t)(v & 0xFF); }

static void unpack(const uint64_t* w, GameBoard* b) {
  memset(b, 0, sizeof(GameBoard));  // Q4: _remove_step_* are never initialised by the reference
  b->forward[0] = (int64_t)w[0];
  b->forward[1] = (int64_t)w[1];
  b->backward[0] = (int64_t)w[2];
  b->backward[1] = (int64_t)w[3];
  b->pieces[0] = b->forward[0] | b->backward[0];
  b->pieces[1] = b->forward[1] | b->backward[1];
  b->empty = UNUSED_BITS ^ MASK ^ (b->pieces[0] | b->pieces[1]);
  b->active = (int)(w[4] & 1);
  b->passive = 1 - b->active;
  b->jump = (int)((w[4] >> 1) & 1);
  b->_ply = (int)((w[4] >> 8) & 0xFFFF);
  b->_last_move = (int)((w[4] >> 24) & 0xFF);
  b->_black_repeats_step = (int)((w[4] >> 32) & 0xFF);
  b->_white_repeats_step = (int)((w[4] >> 40) & 0xFF);
  b->_last_move_black[0] = sx8(w[5]);
  b->_last_move_black[1] = sx8(w[5] >> 8);
  b->_last_move_white[0] = sx8(w[5] >> 16);
  b->_last_move_white[1] = sx8(w[5] >> 24);
}

int ref_am_sizeof_board() { return (int)sizeof(GameBoard); }

void ref_am_clear(uint64_t* w) {
  GameBoard b;
  memset(&b, 0, sizeof(b));
  ClearBoard(&b);
  pack(b, w);
}

// GetValidMovesBinary -> 170 ints
void ref_am_valid_moves(const uint64_t* w, int* out170) {
  GameBoard b;
  unpack(w, &b);
  auto r = GetValidMovesBinary(b);
  for (int i = 0; i < 170; ++i) out170[i] = r[i];
}

// CheckersPlay; returns the reference's bool (true = same side continues a multi-jump)
int ref_am_play(uint64_t* w, int action) {
  GameBoard b;
  unpack(w, &b);
  bool cont = CheckersPlay(&b, action);
  pack(b, w);
  return cont ? 1 : 0;
}

int ref_am_try_play(const uint64_t* w, int action) {
  GameBoard b;
  unpack(w, &b);
  return CheckersTryPlay(b, (Coord)action) ? 1 : 0;
}

int ref_am_is_over(const uint64_t* w) {
  GameBoard b;
  unpack(w, &b);
  return CheckersIsOver(b) ? 1 : 0;
}

// GetObservation(board, player) -> 64 ints, row-major [y][x]
void ref_am_observation(const uint64_t* w, int player, int* out64) {
  GameBoard b;
  unpack(w, &b);
  auto o = GetObservation(b, player);
  for (int y = 0; y < 8; ++y)
    for (int x = 0; x < 8; ++x) out64[y * 8 + x] = o[y][x];
}

// i_to_m table row: move word and direction flag of an action index
void ref_am_action(int action, int64_t* move, int64_t* dir) {
  auto it = moves::i_to_m.find(action);
  *move = it->second[0];
  *dir = it->second[1];
}


// Random-legal playouts driven through the REFERENCE's own GetValidMovesBinary / CheckersPlay (same
// counter-based RNG and reset rule as oracle/american.c am_playout and the GPU kernel): the reference
// CPU path timed for benchmarks/bench_boards.py ("kind": "reference"), and a third cross-check of the
// playout harness.  boards: n x 6 exported words, updated in place.  Returns moves played.
static inline uint64_t ref_splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
uint64_t ref_am_playout(uint64_t* boards, int n, uint64_t board_id0, uint64_t step_base, int nply, uint64_t seed) {
  uint64_t moves = 0;
  for (int bi = 0; bi < n; ++bi) {
    GameBoard b;
    unpack(boards + (size_t)bi * 6, &b);
    for (int t = 0; t < nply; ++t) {
      auto m = GetValidMovesBinary(b);
      int cnt = 0;
      for (int i = 0; i < 170; ++i) cnt += m[i] != 0;
      if (cnt == 0 || b._ply >= 250) { memset(&b, 0, sizeof(b)); ClearBoard(&b); continue; }
      uint64_t r = ref_splitmix64(seed + (board_id0 + (uint64_t)bi) * 0x9E3779B97F4A7C15ull + step_base + (uint64_t)t);
      int k = (int)(r % (uint64_t)cnt), a = 0;
      for (a = 0; a < 170; ++a) if (m[a]) { if (k == 0) break; --k; }
      CheckersPlay(&b, a);
      ++moves;
    }
    pack(b, boards + (size_t)bi * 6);
  }
  return moves;
}
}  // extern "C"
