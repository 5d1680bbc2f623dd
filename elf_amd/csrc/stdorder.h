// Explicit re-implementations of the two libstdc++ behaviours the reference's search results
// depend on (SURVEY.md 7 hard part 1, Appendix C.1), usable on host and device:
//   * std::sort (introsort: median-of-3 partition while > 16, depth limit 2*floor(log2 n) with a
//     heap-sort fallback, final insertion sort) on (action, pi) pairs, comparator pi_a > pi_b  -
//     MCTSGameActor.h:295-300 sorts with it and the sort is NOT stable, so ties resolve by the
//     algorithm's own data movement;
//   * std::unordered_map<unsigned short, ...> iteration order (identity hash, prime bucket counts
//     13 -> 29 -> 59 -> 127 -> 257 -> 541, insert-at-bucket-begin, rehash relinking in list
//     order) - NodeT::stateActions_ (tree_search_node.h:325) is walked in that order by UCT
//     (:394) and MCTSResultT::addActions (tree_search_base.h:321), first strict maximum wins.
// Checked against the real library by tests/test_stdorder.py through the elfhip_test_* hooks (the wave-parallel
// form of the second one by tests/test_gpu_stdorder.py).
#pragma once
#include <stdint.h>
#ifndef ELF_HD
#define ELF_HD __host__ __device__ inline
#endif

namespace stdorder {

struct Item { float key; uint16_t idx; };

ELF_HD bool cmp(const Item& a, const Item& b) { return a.key > b.key; }
ELF_HD void swp(Item& a, Item& b) { Item t = a; a = b; b = t; }

// std::__adjust_heap + std::__push_heap
ELF_HD void adjust_heap(Item* first, int hole, int len, Item value) {
  const int top = hole;
  int second = hole;
  while (second < (len - 1) / 2) {
    second = 2 * (second + 1);
    if (cmp(first[second], first[second - 1])) second--;
    first[hole] = first[second];
    hole = second;
  }
  if ((len & 1) == 0 && second == (len - 2) / 2) {
    second = 2 * (second + 1);
    first[hole] = first[second - 1];
    hole = second - 1;
  }
  int parent = (hole - 1) / 2;
  while (hole > top && cmp(first[parent], value)) {
    first[hole] = first[parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  first[hole] = value;
}

// std::__partial_sort(first, last, last) == make_heap + sort_heap
ELF_HD void heap_sort(Item* first, int len) {
  if (len < 2) return;
  for (int parent = (len - 2) / 2;; --parent) {
    Item v = first[parent];
    adjust_heap(first, parent, len, v);
    if (parent == 0) break;
  }
  for (int last = len; last > 1;) {
    --last;
    Item v = first[last];
    first[last] = first[0];
    adjust_heap(first, 0, last, v);
  }
}

ELF_HD void unguarded_linear_insert(Item* a, int last) {
  Item val = a[last];
  int next = last - 1;
  while (cmp(val, a[next])) { a[last] = a[next]; last = next; --next; }
  a[last] = val;
}

ELF_HD void insertion_sort(Item* a, int first, int last) {
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    if (cmp(a[i], a[first])) {
      Item val = a[i];
      for (int j = i; j > first; --j) a[j] = a[j - 1];
      a[first] = val;
    } else {
      unguarded_linear_insert(a, i);
    }
  }
}

// std::sort(a, a+n, cmp)
ELF_HD void sort_desc(Item* a, int n) {
  if (n < 2) return;
  int lg = 0;
  for (int t = n; t > 1; t >>= 1) ++lg;
  // explicit stack replaces the recursion on the right part; ranges are disjoint so order is free
  int sf[48], sl[48], sd[48], sp = 0;
  sf[0] = 0; sl[0] = n; sd[0] = 2 * lg; sp = 1;
  while (sp > 0) {
    --sp;
    int first = sf[sp], last = sl[sp], depth = sd[sp];
    while (last - first > 16) {
      if (depth == 0) { heap_sort(a + first, last - first); break; }
      --depth;
      // __unguarded_partition_pivot
      int mid = first + (last - first) / 2;
      {  // __move_median_to_first(first, first+1, mid, last-1)
        int x = first + 1, y = mid, z = last - 1;
        if (cmp(a[x], a[y])) {
          if (cmp(a[y], a[z])) swp(a[first], a[y]);
          else if (cmp(a[x], a[z])) swp(a[first], a[z]);
          else swp(a[first], a[x]);
        } else if (cmp(a[x], a[z])) swp(a[first], a[x]);
        else if (cmp(a[y], a[z])) swp(a[first], a[z]);
        else swp(a[first], a[y]);
      }
      int lo = first + 1, hi = last;
      for (;;) {  // __unguarded_partition(first+1, last, pivot = first)
        while (cmp(a[lo], a[first])) ++lo;
        --hi;
        while (cmp(a[first], a[hi])) --hi;
        if (!(lo < hi)) break;
        swp(a[lo], a[hi]);
        ++lo;
      }
      const int cut = lo;
      sf[sp] = cut; sl[sp] = last; sd[sp] = depth; ++sp;  // right part later
      last = cut;
    }
  }
  // __final_insertion_sort
  if (n > 16) {
    insertion_sort(a, 0, 16);
    for (int i = 16; i < n; ++i) unguarded_linear_insert(a, i);
  } else {
    insertion_sort(a, 0, n);
  }
}

// Iteration order of std::unordered_map<unsigned short, T> after inserting distinct keys[0..m)
// in that order.  next: scratch int16[m]; buckets: scratch int16[bucket capacity >= 541].
// out[0..m) receives POSITIONS (indices into keys) in iteration order.  m <= 512.
// key % bucket_count for the bucket counts libstdc++'s prime rehash policy goes through below 542: constant
// divisors compile to a multiply and a shift (the run-time modulo is ~40 instructions on the GPU)
ELF_HD int mod_buckets(int x, int nb) {
  switch (nb) {
    case 1: return 0;
    case 13: return x % 13;
    case 29: return x % 29;
    case 59: return x % 59;
    case 127: return x % 127;
    case 257: return x % 257;
    case 541: return x % 541;
    default: return x % nb;
  }
}

ELF_HD void umap_iteration_order(const uint16_t* keys, int m, int16_t* next, int16_t* buckets, uint16_t* out) {
  const int16_t NONE = -1, BB = -2;  // BB = the list's before-begin sentinel
  int nb = 1, count = 0;
  int16_t head = NONE;  // before_begin.next
  buckets[0] = NONE;
  for (int i = 0; i < m; ++i) {
    int grow = 0;
    if (count == 0) grow = 13; else if (count == 13) grow = 29; else if (count == 29) grow = 59;
    else if (count == 59) grow = 127; else if (count == 127) grow = 257; else if (count == 257) grow = 541;
    if (grow) {  // _M_rehash_aux(unique keys)
      for (int b = 0; b < grow; ++b) buckets[b] = NONE;
      int16_t p = head;
      head = NONE;
      int bbegin = 0;
      while (p != NONE) {
        int16_t nx = next[p];
        int b = mod_buckets(keys[p], grow);
        if (buckets[b] == NONE) {
          next[p] = head; head = p; buckets[b] = BB;
          if (next[p] != NONE) buckets[bbegin] = p;
          bbegin = b;
        } else {
          int16_t before = buckets[b];
          if (before == BB) { next[p] = head; head = p; }
          else { next[p] = next[before]; next[before] = p; }
        }
        p = nx;
      }
      nb = grow;
    }
    // _M_insert_bucket_begin
    int b = mod_buckets(keys[i], nb);
    if (buckets[b] != NONE) {
      int16_t before = buckets[b];
      if (before == BB) { next[i] = head; head = (int16_t)i; }
      else { next[i] = next[before]; next[before] = (int16_t)i; }
    } else {
      next[i] = head; head = (int16_t)i;
      if (next[i] != NONE) buckets[mod_buckets(keys[next[i]], nb)] = (int16_t)i;
      buckets[b] = BB;
    }
    ++count;
  }
  int k = 0;
  for (int16_t p = head; p != NONE; p = next[p]) out[k++] = (uint16_t)p;
}

#ifdef __HIPCC__
// The same iteration order computed by one 64-lane wave instead of a walk of the linked list.  Between two
// rehashes the container only ever (a) puts a node whose bucket is empty at the front of the whole list and
// (b) puts a node whose bucket is occupied at the front of that bucket's run, and _M_rehash_aux applies exactly
// those two rules to the old list in its order.  So with S = (list order at the last rehash) ++ (keys inserted
// since), the list is S sorted by (index in S of the first element of the same bucket, descending; own index in
// S, descending) - a rank every lane can count for its elements.  One such pass per bucket count 13 .. 541.
// keys[0..m), m <= 512, all arrays in LDS: seq uint16[m], rkey int[(m+3)&~3] (16-byte aligned), table int[>= bucket
// count].  out[0..m) receives positions as above.  Every lane of the wave must call it.
// Orders the calling wave's LDS / memory accesses before and after it (lanes of a wave run in lockstep, so no
// barrier instruction is needed; the fence keeps the compiler and the wait counters honest).  Used instead of
// __syncthreads() so that several waves of one workgroup can run these routines on different data.
__device__ inline void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
}

template <int NB>
__device__ inline void umap_epoch_wave(const uint16_t* keys, int cnt, int end, int lane, const uint16_t* src, uint16_t* dst,
                                       int* rkey, int* table) {
  for (int b = lane; b < NB; b += 64) table[b] = 0x7fffffff;
  wave_sync();
  for (int i = lane; i < end; i += 64) {
    const int p = i < cnt ? (int)src[i] : i;
    atomicMin(&table[(int)keys[p] % NB], i);
  }
  wave_sync();
  const int end4 = (end + 3) & ~3;
  for (int i = lane; i < end4; i += 64) {
    const int p = i < cnt ? (int)src[i] : i;
    rkey[i] = i < end ? table[(int)keys[p] % NB] * 1024 + i : (int)0x80000000;
  }
  wave_sync();
  for (int c = 0; c < end; c += 64) {
    const int i = c + lane;
    const bool in = i < end;
    const int mine = rkey[in ? i : 0];
    int before = 0;
    for (int e = 0; e < end4; e += 4) {
      const int4 k = *(const int4*)(rkey + e);   // the same address in every lane: an LDS broadcast
      before += (k.x > mine ? 1 : 0) + (k.y > mine ? 1 : 0) + (k.z > mine ? 1 : 0) + (k.w > mine ? 1 : 0);
    }
    if (in) dst[before] = (uint16_t)(i < cnt ? (int)src[i] : i);
  }
  wave_sync();
}

__device__ inline void umap_iteration_order_wave(const uint16_t* keys, int m, int lane, uint16_t* seq, int* rkey, int* table,
                                                 uint16_t* out) {
  if (m <= 0) return;
  const int passes = 1 + (m > 13) + (m > 29) + (m > 59) + (m > 127) + (m > 257);
  uint16_t* dst = (passes & 1) ? out : seq;   // ping-pong so that the last pass writes out
  uint16_t* src = (passes & 1) ? seq : out;
#define ELF_UMAP_PASS(CNT, NB, NEXT)                                                   \
  if (m > CNT) {                                                                       \
    umap_epoch_wave<NB>(keys, CNT, m < NEXT ? m : NEXT, lane, src, dst, rkey, table);  \
    uint16_t* t = src; src = dst; dst = t;                                             \
  }
  ELF_UMAP_PASS(0, 13, 13)
  ELF_UMAP_PASS(13, 29, 29)
  ELF_UMAP_PASS(29, 59, 59)
  ELF_UMAP_PASS(59, 127, 127)
  ELF_UMAP_PASS(127, 257, 257)
  ELF_UMAP_PASS(257, 541, 512)
#undef ELF_UMAP_PASS
}
#endif

}  // namespace stdorder
