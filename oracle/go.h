/* TEST INFRASTRUCTURE ONLY (oracle).  CPU restatement of the BUILD-DEFINED Go 19x19 rules.
 *
 * PARITY UNPINNED vs ELF: the reference fork contains no Go source, so there is
 * nothing in /root/reference to follow or to be bit-exact against.  The rules below are the
 * build's own definition (DESIGN.md "Go rules"), modelled on the API list that survives in
 * /root/reference/design_doc/go_game.md:6-122 (TryPlay / isSuicideMove / isSimpleKoViolation /
 * check_superko / getTrompTaylorScore / extractAGZ).  This file is pinned by rule-derived
 * known-answer tests (tests/test_oracle_go.py) and is what the HIP kernels are held bit-exact to.
 * It is written with the classic point-array + flood-fill formulation, deliberately unlike the
 * row-bitboard formulation of the HIP kernel, so the two check each other.
 */
#ifndef ORACLE_GO_H
#define ORACLE_GO_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define GO_N 19
#define GO_POINTS 361
#define GO_PASS 361
#define GO_ACTIONS 362
#define GO_MAX_PLY 722
#define GO_STATE_WORDS 22
#define GO_NO_POINT 0xFFFF
#define GO_ZSEED 0x60BA7D5EEDull
#define GO_PLANES 18
#define GO_HISTORY 8

typedef struct {
  int8_t pt[GO_POINTS];  /* 0 empty, 1 black, 2 white; index r*19+c */
  int to_move;           /* 0 black, 1 white */
  int ply, passes, superko;
  int ko;                /* point index or GO_NO_POINT */
  int last;              /* last action, GO_NO_POINT at start */
  int cap[2];            /* stones captured BY black / BY white */
  uint64_t hash;         /* positional Zobrist: xor of go_zobrist(colour, point) */
  int nhist;             /* hashes of all earlier positions of this game (start position first) */
  uint64_t hist[GO_MAX_PLY + 2];
  /* 8-deep stone history for the feature planes, latest first */
  int nstones_hist;
  int8_t stones_hist[GO_HISTORY][GO_POINTS];
} go_state;

uint64_t go_zobrist(int colour, int point);
void go_reset(go_state* s);
int  go_terminated(const go_state* s);                    /* two passes | ply>=722 | superko */
void go_legal_mask(const go_state* s, uint64_t m[6]);     /* bit a: point legal; bit 361: pass (when not terminated) */
int  go_forward(go_state* s, int action);                 /* 0 if terminated/illegal, state unchanged */
float go_evaluate(const go_state* s);                     /* Tromp-Taylor area, komi 7.5: +1/-1, 0 if not terminated */
float go_tt_score(const go_state* s);                     /* black area - white area - 7.5 */
void go_features(const go_state* s, float out[GO_PLANES * GO_POINTS]);

/* flat export (same words as the device export) */
void go_export(const go_state* s, uint64_t w[GO_STATE_WORDS]);
void go_import(go_state* s, const uint64_t w[GO_STATE_WORDS]); /* history := this position only */

/* random-legal playout driver, same contract as am_playout; pass only when no point is legal */
uint64_t go_playout(go_state* boards, int n, uint64_t board_id0, uint64_t step_base, int nply, uint64_t seed);
int go_sizeof_state(void);

#ifdef __cplusplus
}
#endif
#endif
