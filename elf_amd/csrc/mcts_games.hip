// Tree search for the reference's other two games, Russian checkers and Ugolki: the search core of
// mcts_core.h (game independent, see mcts.hip for the design notes) with the game plugged in through
// a small static interface.  The reference instantiates the same TreeSearchT with
// CheckersMCTSActor (russian_checkers/mcts/CheckersMCTSActor.h:79-335) / MCTSGameActor
// (ugolki/mcts/MCTSGameActor.h:79-335), which differ from the American actor only in type names:
// q_flip = side to move is WHITE, terminal -> evaluateGame() > 0 ? +1 : -1 without edges, otherwise
// std::sort of all (action, pi) pairs, checkMove filter, normalise.  Neither game keeps a board
// history in its feature planes, so a leaf's planes depend on the leaf state only.
#include "mcts_core.h"
#include "mcts_quad.h"
#include "russian_dev.h"
#include "ugolki_dev.h"

namespace {
using namespace mcts;

// ---- game adaptors ------------------------------------------------------------------------------
struct RuGame {
  static constexpr int NA = 280, MAXLEGAL = 128, W64 = 5, MW = 5, PLANES = 6, EDGES_PER_NODE = 16;
  using S = rudev::Ru;
  __device__ static void load(const uint64_t* p, S& s) { s.wp = p[0]; s.wk = p[1]; s.bp = p[2]; s.bk = p[3]; s.w4 = p[4]; }
  __device__ static void store(uint64_t* p, const S& s) { p[0] = s.wp; p[1] = s.wk; p[2] = s.bp; p[3] = s.bk; p[4] = s.w4; }
  __device__ static void play(S& s, int action) { rudev::ru_play(s, action); }
  // legal mask + terminated() (CheckersState.h:51-53) + evaluateGame (:52-64) + q_flip
  __device__ static void eval(const S& s, uint64_t mk[MW], bool& terminal, bool& flip, float& tvalue) {
    rudev::ru_moves(s, mk);
    const bool stuck = rudev::ru_count(mk) == 0, cap = rudev::ru_ply(s.w4) >= rudev::RU_MAX_PLY;
    flip = rudev::ru_white(s.w4);
    terminal = stuck || cap;
    tvalue = (!cap && flip) ? 1.0f : -1.0f;  // -1 at the move cap, else -1 if black is stuck, +1 if white is
  }
  // CheckersFeature::extract (CheckersFeature.cc:57-82), lane = square
  __device__ static void planes(const S& s, int lane, float pl[PLANES]) {
    const bool white = rudev::ru_white(s.w4);
    const int rot = 63 - lane;
    pl[0] = (white ? (s.bp >> rot) & 1 : (s.wp >> lane) & 1) ? 1.f : 0.f;
    pl[1] = (white ? (s.bk >> rot) & 1 : (s.wk >> lane) & 1) ? 1.f : 0.f;
    pl[2] = (white ? (s.wp >> lane) & 1 : (s.bp >> rot) & 1) ? 1.f : 0.f;
    pl[3] = (white ? (s.wk >> lane) & 1 : (s.bk >> rot) & 1) ? 1.f : 0.f;
    pl[4] = white ? 0.f : 1.f;
    pl[5] = white ? 1.f : 0.f;
  }
};

struct UgGame {
  static constexpr int NA = 418, MAXLEGAL = 96, W64 = 4, MW = 7, PLANES = 16, EDGES_PER_NODE = 40;
  using S = ugdev::Ug;
  __device__ static void load(const uint64_t* p, S& s) { s.p[0] = p[0]; s.p[1] = p[1]; s.ja = p[2]; s.meta = p[3]; }
  __device__ static void store(uint64_t* p, const S& s) { p[0] = s.p[0]; p[1] = s.p[1]; p[2] = s.ja; p[3] = s.meta; }
  __device__ static void play(S& s, int action) {
    int g = -1, i = 0;
    if (action != ugdev::UG_PASS) ugdev::ug_decode(action, g, i);
    ugdev::ug_play(s, g, i);
  }
  __device__ static void eval(const S& s, uint64_t mk[MW], bool& terminal, bool& flip, float& tvalue) {
    uint64_t fam[8];
    const bool pass = ugdev::ug_families(s, fam);
    ugdev::ug_mask_words(fam, pass, mk);
    flip = ugdev::ug_active(s) == 1;
    const bool cap = ugdev::ug_ply(s) >= ugdev::UG_MAX_PLY;
    terminal = ugdev::ug_over(s) || cap;             // GameState.h:49-51
    // evaluateGame GameState.cc:56-70
    tvalue = cap ? -1.f : ((ugdev::ug_bwin(s) > 0 && ugdev::ug_wwin(s) > 0) ? -1.f : (ugdev::ug_bwin(s) == 2 ? 1.f : -1.f));
  }
  // BoardFeature::extract (ugolki/game/BoardFeature.cc:45-63): planes 0..3, the rest stay zero
  __device__ static void planes(const S& s, int lane, float pl[PLANES]) {
    const bool white = ugdev::ug_active(s) == 1;
    const bool blk = (s.p[0] >> lane) & 1, wht = (s.p[1] >> (63 - lane)) & 1;
#pragma unroll
    for (int q = 0; q < PLANES; ++q) pl[q] = 0.f;
    pl[0] = (white ? wht : blk) ? 1.f : 0.f;
    pl[1] = (white ? blk : wht) ? 1.f : 0.f;
    pl[2] = white ? 0.f : 1.f;
    pl[3] = white ? 1.f : 0.f;
  }
};

template <class G>
__device__ inline uint64_t* gnode(const TV& tv, int node) { return (uint64_t*)(tv.state + (size_t)node * (2 * G::W64)); }

template <class G>
struct Hook {  // allocateState tree_search.h:209-224: clone the parent's state and play the edge's action
  __device__ static inline void on_new_child(const TreeDev&, TV& tv, int node, int child, int action) {
    typename G::S s;
    G::load(gnode<G>(tv, node), s);
    G::play(s, action);
    G::store(gnode<G>(tv, child), s);
    tv.meta[child] = NOT_VISITED | META_STATE_SET;
  }
};

// TreeSearchT::setRootNodeState tree_search.h:570-585 (+ the clear of align_state when not persistent)
template <class G>
__global__ void __launch_bounds__(64)
games_set_roots_kernel(TreeDev d, int persistent, const uint64_t* __restrict__ w, int n, uint8_t* __restrict__ ok) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= d.n_trees) return;
  if (d.t_flags[t] & FLAG_INACTIVE) { if (ok) ok[t] = 1; return; }
  if (!persistent) { reset_tree(d, t); d.t_flags[t] &= ~FLAG_ROOT_MISMATCH; }
  TV tv = tree_view(d, t);
  uint64_t* rb = gnode<G>(tv, 0);
  bool good = true;
  if (tv.meta[0] & META_STATE_SET) {
    for (int k = 0; k < G::W64; ++k) good = good && rb[k] == w[(size_t)k * n + t];
    if (!good) d.t_flags[t] |= FLAG_ROOT_MISMATCH;
  } else {
    for (int k = 0; k < G::W64; ++k) rb[k] = w[(size_t)k * n + t];
    tv.meta[0] |= META_STATE_SET;
  }
  if (ok) ok[t] = good ? 1 : 0;
}

// MODE 0 fused uniform-policy search, 1 select one batch and emit leaves, 2 expand + backup (as mcts.hip)
template <class G, int MODE>
__global__ void __launch_bounds__(64)
games_search_kernel(TreeDev d, Opt o, uint64_t* __restrict__ leaf_states, float* __restrict__ leaf_feat,
                    uint8_t* __restrict__ need_eval, const float* __restrict__ pi, const float* __restrict__ vin) {
  __shared__ ExpandScratch<G::NA, G::MAXLEGAL> S;
  __shared__ int s_leaf[MAXB];
  __shared__ int s_locked[MAXB];
  const int t = blockIdx.x, lane = threadIdx.x;
  int flags = d.t_flags[t];
  if (flags) return;
  TV tv = tree_view(d, t);
  int nn = d.t_nnodes[t], ne = d.t_nedges[t];
  unsigned long long rollouts = 0, evals = 0, depth_sum = 0, edge_sum = 0;
  const int nbatch = MODE == 0 ? (o.R + o.B - 1) / o.B : 1;
  for (int batch = 0; batch < nbatch && !flags; ++batch) {
    if (MODE != 2) {
      for (int j = 0; j < o.B; ++j) {
        int depth = 0;
        const int leaf = select_one<Hook<G>>(d, tv, o, t, j, lane, nn, flags, depth);
        rollouts += 1; depth_sum += (unsigned long long)depth;
        const uint32_t meta = tv.meta[leaf];
        const int lock = (meta & 0xFF) == NOT_VISITED;  // requestEvaluation tree_search_node.h:169-179
        if (lock) tv.meta[leaf] = (meta & ~0xFFu) | EVAL_REQUESTED;
        __threadfence_block();
        if (lane == 0) { s_leaf[j] = leaf; s_locked[j] = lock; }
        if (MODE == 1) {
          const size_t slot = (size_t)t * o.B + j;
          if (lane == 0) { d.leaf[slot] = leaf; d.locked[slot] = (uint8_t)lock; }
          typename G::S s;
          G::load(gnode<G>(tv, leaf), s);
          if (leaf_states && lane < G::W64) leaf_states[slot * G::W64 + lane] = gnode<G>(tv, leaf)[lane];
          if (leaf_feat) {
            float pl[G::PLANES];
            G::planes(s, lane, pl);
#pragma unroll
            for (int q = 0; q < G::PLANES; ++q) leaf_feat[(slot * G::PLANES + q) * 64 + lane] = pl[q];
          }
          if (need_eval && lane == 0) {
            uint64_t mk[G::MW]; bool terminal, flip; float tv_;
            G::eval(s, mk, terminal, flip, tv_);
            need_eval[slot] = (lock && !terminal) ? 1 : 0;
          }
        }
      }
      __syncthreads();
    } else {
      if (lane < o.B) { s_leaf[lane] = d.leaf[(size_t)t * o.B + lane]; s_locked[lane] = d.locked[(size_t)t * o.B + lane]; }
      __syncthreads();
    }
    if (MODE != 1) {
      for (int j = 0; j < o.B; ++j) {
        if (!s_locked[j]) continue;
        const size_t slot = (size_t)t * o.B + j;
        typename G::S s;
        G::load(gnode<G>(tv, s_leaf[j]), s);
        uint64_t mk[G::MW]; bool terminal, flip; float value;
        G::eval(s, mk, terminal, flip, value);   // pre_evaluate (actor :198-218)
        if (!terminal) value = MODE == 2 ? vin[slot] : 0.0f;
        expand_edges<G::NA, G::MAXLEGAL>(d, tv, s_leaf[j], terminal, flip, value, mk, MODE == 2 ? pi + slot * G::NA : nullptr, lane,
                                         ne, flags, S, evals, edge_sum);
      }
      backup_batch(d, tv, o, t, s_leaf);
      __syncthreads();
    }
  }
  if (lane == 0) {
    d.t_nnodes[t] = nn; d.t_nedges[t] = ne; d.t_flags[t] = flags;
    if (rollouts) atomicAdd(&d.stats[0], rollouts);
    if (evals) atomicAdd(&d.stats[1], evals);
    if (depth_sum) atomicAdd(&d.stats[2], depth_sum);
    if (edge_sum) atomicAdd(&d.stats[3], edge_sum);
  }
}

template <class G>
struct GameMcts : MctsBase {
  int alloc(int n_trees, const elfhip_ts_options* opt) { return alloc_common(n_trees, opt, G::NA, 2 * G::W64, G::EDGES_PER_NODE); }
  int set_roots(BoardSlab* slab, uint8_t* d_ok) override {
    hipLaunchKernelGGL(games_set_roots_kernel<G>, dim3((d.n_trees + 63) / 64), dim3(64), 0, stream, d, o.persistent,
                       (const uint64_t*)slab->dev_ptr[0], slab->n, d_ok);
    HIPCHECK(hipGetLastError());
    return ELFHIP_OK;
  }
  int run_uniform() override {
    if (quad_enabled()) {   // four trees per wave (mcts_quad.h)
      hipLaunchKernelGGL(quad_search_kernel<G>, dim3((d.n_trees + 3) / 4), dim3(64), quad_dynamic_lds<G>(o.B), stream, d, o);
      HIPCHECK(hipGetLastError());
      return ELFHIP_OK;
    }
    hipLaunchKernelGGL((games_search_kernel<G, 0>), dim3(d.n_trees), dim3(64), 0, stream, d, o, (uint64_t*)nullptr,
                       (float*)nullptr, (uint8_t*)nullptr, (const float*)nullptr, (const float*)nullptr);
    HIPCHECK(hipGetLastError());
    return ELFHIP_OK;
  }
  int select(uint64_t* d_leaf_states, float* d_leaf_features, uint8_t* d_need_eval) override {
    { const int rc = zero_need(d_need_eval); if (rc != ELFHIP_OK) return rc; }
    hipLaunchKernelGGL((games_search_kernel<G, 1>), dim3(d.n_trees), dim3(64), 0, stream, d, o, d_leaf_states,
                       d_leaf_features, d_need_eval, (const float*)nullptr, (const float*)nullptr);
    HIPCHECK(hipGetLastError());
    return ELFHIP_OK;
  }
  int backup(const float* d_pi, const float* d_value) override {
    hipLaunchKernelGGL((games_search_kernel<G, 2>), dim3(d.n_trees), dim3(64), 0, stream, d, o, (uint64_t*)nullptr,
                       (float*)nullptr, (uint8_t*)nullptr, d_pi, d_value);
    HIPCHECK(hipGetLastError());
    return ELFHIP_OK;
  }
};

template <class G>
MctsBase* make(int n_trees, const elfhip_ts_options* opt, int* rc) {
  GameMcts<G>* m = new (std::nothrow) GameMcts<G>();
  if (!m) { elfhip_set_error("mcts_create: host allocation failed"); *rc = ELFHIP_ENOMEM; return nullptr; }
  *rc = m->alloc(n_trees, opt);
  if (*rc != ELFHIP_OK) { delete m; return nullptr; }
  return m;
}

}  // namespace

mcts::MctsBase* elfhip_make_mcts_russian(int n_trees, const elfhip_ts_options* opt, int* rc) { return make<RuGame>(n_trees, opt, rc); }
mcts::MctsBase* elfhip_make_mcts_ugolki(int n_trees, const elfhip_ts_options* opt, int* rc) { return make<UgGame>(n_trees, opt, rc); }
