/* elfhip.h - C-ABI of the MI355X (gfx950) self-play hot path.
 *
 * Plain C, no torch / STL types.  Every call returns 0 on success or a negative ELFHIP_E* code;
 * elfhip_last_error() returns a thread-local message for the last failure.  No exceptions cross
 * this boundary.  Pointers named d_* are DEVICE pointers (hipMalloc / torch.cuda memory on the
 * handle's device); pointers named h_* are HOST pointers.  All work of a handle is enqueued on the
 * handle's HIP stream (default: the null stream; see elfhip_*_set_stream) and calls return without
 * synchronising unless documented.  A handle is not thread-safe; distinct handles are independent.
 *
 * What each entry point replaces in the reference (ARVILab/ELF, paths under src_cpp/):
 *   boards_*    the per-game board rules + GameState wrapper that the reference runs one C++
 *               thread per game for: elfgames/american_checkers/game/GameBoard.cc:10-528,
 *               GameState.cc:4-92, BoardFeature.cc:47-77 (and the russian_checkers / ugolki
 *               equivalents).  Go 19x19 has no counterpart in this fork: its rules
 *               are build-defined, see DESIGN.md.
 *   mcts_*      elf/ai/tree_search/tree_search.h:112-378,405-622, tree_search_node.h:98-609,
 *               tree_search_base.h:137-367 and elfgames/<game>/mcts/MCTSGameActor.h:79-335.
 *   net_*       src_py/elfgames/american_checkers/model_american_checkers.py:137-349.
 * The reference-side binding a maintainer would add is shown in INTEGRATION.md.
 */
#ifndef ELFHIP_H
#define ELFHIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ELFHIP_OK 0
#define ELFHIP_EINVAL (-1)   /* bad argument (null pointer, size out of range, unknown game) */
#define ELFHIP_EHIP (-2)     /* a HIP runtime call failed */
#define ELFHIP_ENOMEM (-3)   /* device allocation failed */
#define ELFHIP_ESTATE (-4)   /* handle in the wrong state for the call */

enum elfhip_game {
  ELFHIP_AMERICAN = 0, /* american_checkers: 170 actions, 6 state words, 36x8x8 planes */
  ELFHIP_RUSSIAN = 1,  /* russian_checkers : 280 actions */
  ELFHIP_UGOLKI = 2,   /* ugolki           : 418 actions */
  ELFHIP_GO19 = 3      /* Go 19x19         : 362 actions (361 points + pass), build-defined */
};

enum elfhip_dtype { ELFHIP_F32 = 0, ELFHIP_BF16 = 1, ELFHIP_BITS = 2 /* one u64/u32 word group per plane */ };

const char* elfhip_last_error(void);
/* Library / device sanity: returns number of visible HIP devices (>=0) or a negative error. */
int elfhip_device_count(void);
/* Version of this ABI (bumped on incompatible change). */
int elfhip_abi_version(void);

/* Static per-game geometry (no GPU needed). */
typedef struct {
  int num_actions;      /* size of the policy vector (GameFeature.h:225-232 "checkers_num_action") */
  int mask_words;       /* u64 words of a legal mask                                            */
  int state_words;      /* u64 words of one exported board state                                 */
  int num_planes;       /* feature planes ("checkers_num_planes")                                */
  int board_size;       /* 8 or 19                                                               */
  int max_ply;          /* game is terminated at ply >= max_ply                                  */
} elfhip_game_info;
int elfhip_game_get_info(int game, elfhip_game_info* out);

/* ------------------------------------------------------------------------------------------------
 * Board slabs: n independent boards of one game, struct-of-arrays in HBM.
 * Replaces GameBoard/GameState of the reference (one object per game thread).
 * ---------------------------------------------------------------------------------------------- */
typedef struct elfhip_boards elfhip_boards;

int elfhip_boards_create(int game, int n_boards, int device, elfhip_boards** out);
int elfhip_boards_destroy(elfhip_boards* b);
int elfhip_boards_set_stream(elfhip_boards* b, void* hip_stream);
int elfhip_boards_count(const elfhip_boards* b);
int elfhip_boards_game(const elfhip_boards* b);

/* GameState::reset (GameState.cc:24-30) on every board; clears history and counters. */
int elfhip_boards_reset(elfhip_boards* b);

/* Copy all board states to / from a HOST buffer of n*state_words u64 (array-of-structs, layout in
 * DESIGN.md; for American checkers the 6 words of oracle/american.h).  Synchronises the stream.
 * Import also resets each board's feature history to that single position. */
int elfhip_boards_export(elfhip_boards* b, uint64_t* h_states);
int elfhip_boards_import(elfhip_boards* b, const uint64_t* h_states);

/* GetValidMovesBinary (GameBoard.cc:131-162) for every board: d_mask[n][mask_words] u64, bit a set
 * iff action a is legal.  d_terminal (nullable) [n] u8 receives GameState::terminated()
 * (GameState.h:51-53). */
int elfhip_boards_legal_mask(elfhip_boards* b, uint64_t* d_mask, uint8_t* d_terminal);

/* GameState::forward (GameState.cc:4-18) on every board with its own action: d_actions[n] i16.
 * A negative action skips that board.  d_ok (nullable) [n] u8 receives forward()'s return value
 * (0: terminated or illegal, state unchanged).  Legal moves push onto the 6-deep history. */
int elfhip_boards_step(elfhip_boards* b, const int16_t* d_actions, uint8_t* d_ok);

/* BoardFeature::extract (BoardFeature.cc:47-77) for every board.
 *   ELFHIP_F32 : d_out float [n][planes][size][size]   (the reference's "s" batch key)
 *   ELFHIP_BF16: same shape, bf16
 *   ELFHIP_BITS: u64 [n][planes] for 8x8 games (bit y*8+x) */
int elfhip_boards_features(elfhip_boards* b, void* d_out, int dtype);

/* evaluateGame (GameState.cc:65-77): d_value float [n] (0 when not terminated). */
int elfhip_boards_evaluate(elfhip_boards* b, float* d_value);

/* Synthetic random-legal playouts: every board advances `nply`
 * steps inside ONE kernel launch, state held in registers.  Per step: terminated -> reset in place
 * (0 moves); else r = splitmix64(seed + board_id*0x9E3779B97F4A7C15 + step_counter), action =
 * (r mod n_legal)-th set bit of the legal mask in ascending action index.  board ids are
 * first_board_id + index; the per-board step counter persists in the handle (reset by _reset).
 * d_moves (nullable): u64 counter incremented by the number of moves actually played. */
int elfhip_boards_playout(elfhip_boards* b, int nply, uint64_t seed, uint64_t first_board_id,
                          unsigned long long* d_moves);

/* ------------------------------------------------------------------------------------------------
 * Host-only hooks for the explicit libstdc++-order emulations the search kernels use
 * (elf_amd/csrc/stdorder.h).  No reference counterpart: they exist so CPU tests can compare the
 * emulations with the real std::sort / std::unordered_map.
 *   sort : order[i] = index of the i-th element after std::sort(desc by key) of n (idx,key) pairs
 *   umap : iter_keys = keys in std::unordered_map<unsigned short,...> iteration order after
 *          inserting the m distinct keys in the given order (m <= 512)
 * ---------------------------------------------------------------------------------------------- */
int elfhip_test_sort_order(int n, const float* h_keys, uint16_t* h_order);
int elfhip_test_umap_order(int m, const uint16_t* h_keys, uint16_t* h_iter_keys);

#ifdef __cplusplus
}
#endif
#endif /* ELFHIP_H */
