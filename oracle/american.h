/* TEST INFRASTRUCTURE ONLY (oracle). Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this; the product path (elf_amd/csrc) never does.
 *
 * CPU restatement of the reference's American checkers rules
 * (/root/reference/src_cpp/elfgames/american_checkers/game/GameBoard.{h,cc}, GameState.{h,cc},
 * BoardFeature.cc).  Pinned bit-exactly against the compiled reference (oracle/_ref) by
 * tests/test_oracle_american.py in the build container, and against committed fixtures
 * (tests/golden/american_*.npz) everywhere else.
 */
#ifndef ORACLE_AMERICAN_H
#define ORACLE_AMERICAN_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AM_NUM_ACTIONS 170
#define AM_MAX_PLY 250      /* TOTAL_MAX_MOVE  GameBoard.h:46 */
#define AM_REPEAT_MOVE 4    /* REPEAT_MOVE     GameBoard.h:48 */
#define AM_HISTORY 6        /* MAX_CHECKERS_HISTORY GameBoard.h:35 */
#define AM_M_INVALID 171    /* GameBoard.h:33 */

/* Flat board: the same 6 words the HIP kernels keep per board (struct-of-arrays on the device).
 *  w[0..1] forward[BLACK,WHITE]   w[2..3] backward[BLACK,WHITE]      (GameBoard.h:65-66)
 *  w[4] = active | jump<<1 | ply<<8 | last_move<<24 | black_repeats<<32 | white_repeats<<40
 *  w[5] = last_move_black[0] | [1]<<8 | last_move_white[0]<<16 | [1]<<24     (0xFF encodes -1)
 * pieces[] and empty (GameBoard.h:67-68) are derived: pieces = forward|backward. */
typedef struct { uint64_t w[6]; } am_state;

void am_clear(am_state* s);                              /* ClearBoard       GameBoard.cc:10-35  */
void am_legal_mask(const am_state* s, uint64_t m[3]);    /* GetValidMovesBinary :131-162          */
int  am_play(am_state* s, int action);                   /* CheckersPlay     :37-129 (returns 1 if the mover continues) */
int  am_is_over(const am_state* s);                      /* CheckersIsOver   :201-203             */
void am_observation(const am_state* s, int player, int8_t out[64]); /* GetObservation :210-277   */

/* action table (HashAllMoves.h:193-363), regenerated from the layout rule, not copied */
int64_t am_action_move(int action);   /* two-bit move word, negative for jumps */
int     am_action_dir(int action);    /* 1 = towards higher bit index */

/* GameState wrapper (GameState.h:7-100, GameState.cc:4-92) with its 6-deep board history */
typedef struct {
  am_state board;
  am_state hist[AM_HISTORY];
  int hist_len;
} am_game;

void  am_game_reset(am_game* g);                         /* GameState::reset  GameState.cc:24-30 */
int   am_game_terminated(const am_game* g);              /* GameState.h:51-53 */
int   am_game_forward(am_game* g, int action);           /* GameState::forward GameState.cc:4-18 */
float am_game_evaluate(const am_game* g);                /* evaluateGame :65-77 */
void  am_game_features(const am_game* g, float out[36 * 64]); /* BoardFeature::extract BoardFeature.cc:47-77 */

/* splitmix64 counter RNG of SURVEY.md §8(d) */
uint64_t elf_splitmix64(uint64_t x);

/* Random-legal playout driver (SURVEY.md §8(d) C2 inputs): advances `n` boards `nply` plies,
 * resetting in place on terminal.  Returns number of moves played. board_id0 = id of boards[0].
 * step_base = number of plies these boards were already driven (the RNG counter). */
uint64_t am_playout(am_state* boards, int n, uint64_t board_id0, uint64_t step_base, int nply, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
