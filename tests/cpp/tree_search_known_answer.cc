// Drives include/elfhip_tree_search.hpp the way reference code drives TreeSearchT (run -> best action ->
// forward -> treeAdvance) and prints "move: action total_visits | action:visits ..." per move for tree 0,
// so the Python test can compare with the compiled reference's known answers.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "elfhip_tree_search.hpp"

int main(int argc, char** argv) {
  const int batch = argc > 1 ? atoi(argv[1]) : 1;
  const int n = 3, moves = 12;
  using namespace elfhip::tree_search;
  TSOptions o;
  o.num_threads = 1; o.num_rollouts_per_thread = 256; o.num_rollouts_per_batch = batch; o.virtual_loss = 1;
  o.persistent_tree = true; o.pick_method = "most_visited"; o.alg_opt.use_prior = true; o.alg_opt.c_puct = 1.5f;
  elfhip_boards* b = nullptr;
  if (elfhip_boards_create(ELFHIP_AMERICAN, n, 0, &b) < 0) { fprintf(stderr, "%s\n", elfhip_last_error()); return 1; }
  try {
    TreeSearchBatch ts(ELFHIP_AMERICAN, n, o);
    int16_t* d_act = nullptr; uint8_t* d_ok = nullptr;
    if (hipMalloc((void**)&d_act, n * 2) != hipSuccess || hipMalloc((void**)&d_ok, n) != hipSuccess) return 1;
    for (int mv = 0; mv < moves; ++mv) {
      std::vector<TreeSearchBatch::MCTSResult> res = ts.runUniform(b);
      printf("%d: %d %d |", mv, (int)res[0].best_action, res[0].total_visits);
      for (auto& p : res[0].action_edge_pairs) printf(" %d:%d", (int)p.first, p.second.num_visits);
      printf("\n");
      std::vector<int16_t> a(n);
      for (int i = 0; i < n; ++i) a[i] = res[i].best_action;
      if (hipMemcpy(d_act, a.data(), n * 2, hipMemcpyHostToDevice) != hipSuccess) return 1;
      if (elfhip_boards_step(b, d_act, d_ok) < 0) return 1;
      ts.treeAdvance(a);
    }
    // a root that does not match the tree is the reference's std::range_error
    if (elfhip_boards_reset(b) < 0) return 1;
    bool threw = false;
    try { ts.runUniform(b); } catch (const std::range_error&) { threw = true; }
    printf("mismatch_throws: %d\n", threw ? 1 : 0);
  } catch (const std::exception& e) { fprintf(stderr, "exception: %s\n", e.what()); return 2; }
  elfhip_boards_destroy(b);
  return 0;
}
