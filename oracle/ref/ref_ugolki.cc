// TEST INFRASTRUCTURE ONLY (oracle).  extern "C" harness around the UNMODIFIED reference Ugolki
// board (/root/reference/src_cpp/elfgames/ugolki/game/GameBoard.{h,cc}), compiled where it lies by
// oracle/Makefile into oracle/_ref/libref_ugolki.so (its own library: the symbol names clash with
// the American-checkers GameBoard.cc).
#include <array>
#include <string>
#include <sstream>
#include <cstring>
#include <cstdint>
#include "elfgames/ugolki/game/GameBoard.h"

extern "C" {

// Flat state (same words as oracle/ugolki.h): w[0] = pieces[BLACK_PLAYER], w[1] = pieces[WHITE_PLAYER],
// w[2] = jump_action, w[3] = active | black_win<<1 | white_win<<4 | ply<<8
static void pack(const GameBoard& b, uint64_t* w) {
  w[0] = b.pieces[BLACK_PLAYER]; w[1] = b.pieces[WHITE_PLAYER]; w[2] = b.jump_action;
  w[3] = (uint64_t)(b.active & 1) | ((uint64_t)(b.black_win & 7) << 1) | ((uint64_t)(b.white_win & 7) << 4) |
         ((uint64_t)(b._ply & 0xFFFF) << 8);
}
static void unpack(const uint64_t* w, GameBoard* b) {
  memset(b, 0, sizeof(*b));
  b->pieces[BLACK_PLAYER] = w[0]; b->pieces[WHITE_PLAYER] = w[1]; b->jump_action = w[2];
  b->active = (int)(w[3] & 1); b->passive = 1 - b->active;
  b->black_win = (int)((w[3] >> 1) & 7); b->white_win = (int)((w[3] >> 4) & 7);
  b->_ply = (int)((w[3] >> 8) & 0xFFFF);
  b->_last_move = M_INVALID;
}

void ref_ug_clear(uint64_t* w) { GameBoard b; memset(&b, 0, sizeof(b)); ClearBoard(&b); pack(b, w); }
void ref_ug_valid_moves(const uint64_t* w, int* out418) {
  GameBoard b; unpack(w, &b);
  auto r = GetValidMovesBinary(b);
  for (int i = 0; i < 418; ++i) out418[i] = r[i];
}
void ref_ug_play(uint64_t* w, int action) { GameBoard b; unpack(w, &b); Play(&b, action); pack(b, w); }
int ref_ug_tryplay(const uint64_t* w, int action) { GameBoard b; unpack(w, &b); return TryPlay(b, (Coord)action) ? 1 : 0; }
int ref_ug_is_over(const uint64_t* w) { GameBoard b; unpack(w, &b); return IsOver(b) ? 1 : 0; }
void ref_ug_observation(const uint64_t* w, int player, int* out64) {
  GameBoard b; unpack(w, &b);
  auto o = GetObservation(b, player);
  for (int y = 0; y < 8; ++y) for (int x = 0; x < 8; ++x) out64[y * 8 + x] = o[y][x];
}
int ref_ug_action(int action, uint64_t* bits, uint64_t* flags) {
  auto it = moves::i_to_m.find(action);
  if (it == moves::i_to_m.end()) return 0;
  *bits = it->second[0]; *flags = it->second[1];
  return 1;
}
int ref_ug_last_move(void) { GameBoard b; memset(&b, 0, sizeof(b)); ClearBoard(&b); Play(&b, 112); return (int)b._last_move; }

}  // extern "C"
