// TEST INFRASTRUCTURE ONLY (oracle).  extern "C" harness around the UNMODIFIED reference Russian
// checkers board (/root/reference/src_cpp/elfgames/russian_checkers/game/CheckersBoard.{h,cc}),
// compiled where it lies by oracle/Makefile into oracle/_ref/libref_russian.so.
#include <array>
#include <string>
#include <sstream>
#include <cstring>
#include <cstdint>
#include "elfgames/russian_checkers/game/CheckersBoard.h"

extern "C" {

// Flat state (same words as oracle/russian.h): w[0..3] = bitboards (bit y*8+x) of white pawns,
// white kings, black pawns, black kings; w[4] = player(1 = white) | has_next<<1 | next_y<<2 |
// next_x<<5 | ply<<8 | last_move<<24
static void pack(const CheckersBoard& b, uint64_t* w) {
  w[0] = w[1] = w[2] = w[3] = 0;
  for (int y = 0; y < 8; ++y)
    for (int x = 0; x < 8; ++x) {
      int v = b.board[y][x];
      uint64_t bit = 1ull << (y * 8 + x);
      if (v == WHITE_PAWN) w[0] |= bit; else if (v == WHITE_KING) w[1] |= bit;
      else if (v == BLACK_PAWN) w[2] |= bit; else if (v == BLACK_KING) w[3] |= bit;
    }
  uint64_t has = b.next_bit_y != -1;
  w[4] = (uint64_t)(b.current_player == WHITE_PLAYER) | (has << 1) | ((uint64_t)(has ? b.next_bit_y : 0) << 2) |
         ((uint64_t)(has ? b.next_bit_x : 0) << 5) | ((uint64_t)(b._ply & 0xFFFF) << 8) | ((uint64_t)(b._last_move & 0xFFFF) << 24);
}

static void unpack(const uint64_t* w, CheckersBoard* b) {
  memset(b, 0, sizeof(*b));
  for (int y = 0; y < 8; ++y)
    for (int x = 0; x < 8; ++x) {
      uint64_t bit = 1ull << (y * 8 + x);
      int v = 0;
      if (w[0] & bit) v = WHITE_PAWN; else if (w[1] & bit) v = WHITE_KING;
      else if (w[2] & bit) v = BLACK_PAWN; else if (w[3] & bit) v = BLACK_KING;
      b->board[y][x] = v;
    }
  b->current_player = (w[4] & 1) ? WHITE_PLAYER : BLACK_PLAYER;
  bool has = (w[4] >> 1) & 1;
  b->next_bit_y = has ? (int)((w[4] >> 2) & 7) : -1;
  b->next_bit_x = has ? (int)((w[4] >> 5) & 7) : -1;
  b->_ply = (int)((w[4] >> 8) & 0xFFFF);
  b->_last_move = (int)((w[4] >> 24) & 0xFFFF);
}

void ref_ru_clear(uint64_t* w) { CheckersBoard b; memset(&b, 0, sizeof(b)); ClearBoard(&b); pack(b, w); }

void ref_ru_valid_moves(const uint64_t* w, int* out280) {
  CheckersBoard b; unpack(w, &b);
  auto r = GetValidMovesBinary(b);
  for (int i = 0; i < 280; ++i) out280[i] = r[i];
}

void ref_ru_play(uint64_t* w, int action) { CheckersBoard b; unpack(w, &b); CheckersPlay(&b, (Coord)action); pack(b, w); }

int ref_ru_is_over(const uint64_t* w) { CheckersBoard b; unpack(w, &b); return CheckersIsOver(b) ? 1 : 0; }

void ref_ru_observation(const uint64_t* w, int player_is_white, int* out64) {
  CheckersBoard b; unpack(w, &b);
  auto o = GetObservation(b, player_is_white ? WHITE_PLAYER : BLACK_PLAYER);
  for (int y = 0; y < 8; ++y) for (int x = 0; x < 8; ++x) out64[y * 8 + x] = o[y][x];
}

void ref_ru_action(int action, int* from, int* to) {
  auto it = moves::i_to_m.find(action);
  *from = (int)it->second[0]; *to = (int)it->second[1];
}

}  // extern "C"
