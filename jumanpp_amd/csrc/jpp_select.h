// Step-for-step restatement of libstdc++'s std::nth_element (GCC 11
// bits/stl_algo.h __introselect / __unguarded_partition_pivot /
// __move_median_to_first / __insertion_sort / __heap_select, and
// bits/stl_heap.h __adjust_heap / __push_heap) on an index array.
//
// Why: the reference picks the right nodes that receive full global-beam
// scoring with std::nth_element over float prescores
// (ScoreProcessor::makeT0cutoffBeam, src/core/analysis/score_processor.cc:471-495).
// With tied scores (aliased dictionary entries, zero weights) the chosen set
// depends on the exact sequence of comparisons and swaps, so bit-identical
// lattices need the same algorithm, not just "a" selection.  It runs on one
// lane over LDS-resident arrays (R <= 256 elements).
#ifndef JPP_SELECT_H
#define JPP_SELECT_H

#include "jpp_rt.h"

namespace jpp {

struct ScoreGreater {
  const float* sc;
  __device__ __forceinline__ bool operator()(u16 a, u16 b) const { return sc[a] > sc[b]; }
};

template <typename T>
__device__ __forceinline__ void sel_swap(T* a, T* b) {
  T t = *a;
  *a = *b;
  *b = t;
}

template <typename T, typename C>
__device__ inline void sel_insertion_sort(T* first, T* last, C comp) {
  if (first == last) return;
  for (T* i = first + 1; i != last; ++i) {
    if (comp(*i, *first)) {
      T val = *i;
      for (T* p = i; p != first; --p) *p = *(p - 1);
      *first = val;
    } else {
      T val = *i;
      T* lastp = i;
      T* next = i - 1;
      while (comp(val, *next)) {
        *lastp = *next;
        lastp = next;
        --next;
      }
      *lastp = val;
    }
  }
}

template <typename T, typename C>
__device__ inline void sel_move_median_to_first(T* result, T* a, T* b, T* c, C comp) {
  if (comp(*a, *b)) {
    if (comp(*b, *c)) sel_swap(result, b);
    else if (comp(*a, *c)) sel_swap(result, c);
    else sel_swap(result, a);
  } else if (comp(*a, *c)) {
    sel_swap(result, a);
  } else if (comp(*b, *c)) {
    sel_swap(result, c);
  } else {
    sel_swap(result, b);
  }
}

template <typename T, typename C>
__device__ inline T* sel_unguarded_partition(T* first, T* last, T* pivot, C comp) {
  for (;;) {
    while (comp(*first, *pivot)) ++first;
    --last;
    while (comp(*pivot, *last)) --last;
    if (!(first < last)) return first;
    sel_swap(first, last);
    ++first;
  }
}

template <typename T, typename C>
__device__ inline void sel_push_heap(T* first, long hole, long top, T value, C comp) {
  long parent = (hole - 1) / 2;
  while (hole > top && comp(first[parent], value)) {
    first[hole] = first[parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  first[hole] = value;
}

template <typename T, typename C>
__device__ inline void sel_adjust_heap(T* first, long hole, long len, T value, C comp) {
  const long top = hole;
  long second = hole;
  while (second < (len - 1) / 2) {
    second = 2 * (second + 1);
    if (comp(first[second], first[second - 1])) second--;
    first[hole] = first[second];
    hole = second;
  }
  if ((len & 1) == 0 && second == (len - 2) / 2) {
    second = 2 * (second + 1);
    first[hole] = first[second - 1];
    hole = second - 1;
  }
  sel_push_heap(first, hole, top, value, comp);
}

template <typename T, typename C>
__device__ inline void sel_heap_select(T* first, T* middle, T* last, C comp) {
  long len = middle - first;
  if (len >= 2) {
    long parent = (len - 2) / 2;
    for (;;) {
      T value = first[parent];
      sel_adjust_heap(first, parent, len, value, comp);
      if (parent == 0) break;
      parent--;
    }
  }
  for (T* i = middle; i < last; ++i) {
    if (comp(*i, *first)) {
      T value = *i;
      *i = *first;
      sel_adjust_heap(first, 0, len, value, comp);
    }
  }
}

template <typename C>
__device__ inline void nth_element_u16(u16* first, u16* nth, u16* last, C comp) {
  if (first == last || nth == last) return;
  long n = last - first;
  long lg = 0;
  while ((n >> (lg + 1)) != 0) ++lg;
  long depth = lg * 2;
  while (last - first > 3) {
    if (depth == 0) {
      sel_heap_select(first, nth + 1, last, comp);
      sel_swap(first, nth);
      return;
    }
    --depth;
    u16* mid = first + (last - first) / 2;
    sel_move_median_to_first(first, first + 1, mid, last - 1, comp);
    u16* cut = sel_unguarded_partition(first + 1, last, first, comp);
    if (cut <= nth) first = cut;
    else last = cut;
  }
  sel_insertion_sort(first, last, comp);
}

// std::sort (GCC 11 bits/stl_algo.h: __introsort_loop with threshold 16, then
// __final_insertion_sort).  For n <= 16 this is a plain (stable) insertion sort.
template <typename T, typename C>
__device__ inline void std_sort(T* first, T* last, C comp) {
  if (first == last) return;
  long n = last - first;
  long lg = 0;
  while ((n >> (lg + 1)) != 0) ++lg;
  // explicit stack instead of the recursion on the right half (the two halves are disjoint,
  // so the processing order does not change the result)
  T* stF[24];
  T* stL[24];
  long stD[24];
  int sp = 0;
  stF[sp] = first;
  stL[sp] = last;
  stD[sp] = lg * 2;
  ++sp;
  while (sp > 0) {
    --sp;
    T* f = stF[sp];
    T* l = stL[sp];
    long depth = stD[sp];
    while (l - f > 16) {
      if (depth == 0) {
        // __partial_sort(f, l, l): make_heap + sort_heap
        sel_heap_select(f, l, l, comp);
        T* e = l;
        while (e - f > 1) {
          --e;
          T value = *e;
          *e = *f;
          sel_adjust_heap(f, 0, (long)(e - f), value, comp);
        }
        break;
      }
      --depth;
      T* mid = f + (l - f) / 2;
      sel_move_median_to_first(f, f + 1, mid, l - 1, comp);
      T* cut = sel_unguarded_partition(f + 1, l, f, comp);
      if (sp < 24) {
        stF[sp] = cut;
        stL[sp] = l;
        stD[sp] = depth;
        ++sp;
      }
      l = cut;
    }
  }
  if (last - first > 16) {
    sel_insertion_sort(first, first + 16, comp);
    for (T* i = first + 16; i != last; ++i) {
      T val = *i;
      T* lastp = i;
      T* next = i - 1;
      while (comp(val, *next)) {
        *lastp = *next;
        lastp = next;
        --next;
      }
      *lastp = val;
    }
  } else {
    sel_insertion_sort(first, last, comp);
  }
}


// The partitioning half of std_sort above, without the final insertion pass.  libstdc++'s
// __final_insertion_sort is a STABLE sort of whatever the introsort loop leaves (strict comparisons, elements
// only move past strictly smaller ones), so std::sort(first, last) == stable_sort(the array as partitioned):
// a caller that can rank in parallel runs only this serial part -- a handful of Hoare partitions for the
// 17..32 candidates of a beam -- and then takes "number of greater elements + number of equal elements at
// smaller positions" as the final position.  Returns false if the depth limit was hit (heap-sort fallback,
// which is not stable): the caller then has to replay std_sort in full.
template <typename T, typename C>
__device__ inline bool std_sort_partition_only(T* first, T* last, C comp) {
  if (first == last) return true;
  long n = last - first;
  long lg = 0;
  while ((n >> (lg + 1)) != 0) ++lg;
  T* stF[24];
  T* stL[24];
  long stD[24];
  int sp = 0;
  stF[sp] = first;
  stL[sp] = last;
  stD[sp] = lg * 2;
  ++sp;
  while (sp > 0) {
    --sp;
    T* f = stF[sp];
    T* l = stL[sp];
    long depth = stD[sp];
    while (l - f > 16) {
      if (depth == 0) return false;
      --depth;
      T* mid = f + (l - f) / 2;
      sel_move_median_to_first(f, f + 1, mid, l - 1, comp);
      T* cut = sel_unguarded_partition(f + 1, l, f, comp);
      if (sp < 24) {
        stF[sp] = cut;
        stL[sp] = l;
        stD[sp] = depth;
        ++sp;
      }
      l = cut;
    }
  }
  return true;
}

// The same for at most 32 elements, without the explicit stack (a private array would live in scratch memory
// on the device, one HBM round trip per push and pop).  With n <= 32 at most one side of a partition is longer
// than 16, and a side of 16 or fewer elements is not touched again before the final insertion pass, so the
// recursion of __introsort_loop degenerates to "partition the one long range that is left"; the depth budget
// shrinks by one per partition on either side, exactly as in the recursive form.
template <typename T, typename C>
__device__ inline bool std_sort_partition_only_le32(T* first, T* last, C comp) {
  long n = last - first;
  long lg = 0;
  while ((n >> (lg + 1)) != 0) ++lg;
  long depth = lg * 2;
  T* f = first;
  T* l = last;
  while (l - f > 16) {
    if (depth == 0) return false;
    --depth;
    T* mid = f + (l - f) / 2;
    sel_move_median_to_first(f, f + 1, mid, l - 1, comp);
    T* cut = sel_unguarded_partition(f + 1, l, f, comp);
    if (l - cut > 16) f = cut;   // the right part is the long one (the left one is then short: done)
    else l = cut;                // otherwise carry on with the left part
  }
  return true;
}

// util::part_step / util::partition (src/util/stl_util.h:51-135): the quickselect makeT0Beam runs when
// it has more than beam*4/3 candidates; returns the end of the part that is then std::sort-ed.
template <typename T, typename C>
__device__ inline T* jpp_part_step(T* start, T* end, C comp) {
  long sz = end - start;
  if (sz == 1) return end;
  if (sz == 2) {
    T* n = start + 1;
    if (comp(*n, *start)) sel_swap(n, start);
    return n;
  }
  if (sz == 3) {
    T* n0 = start;
    T* n1 = n0 + 1;
    T* n2 = n1 + 1;
    if (comp(*n1, *n0)) sel_swap(n1, n0);
    if (comp(*n2, *n1)) sel_swap(n2, n1);
    if (comp(*n1, *n0)) sel_swap(n1, n0);
    return n1;
  }
  T* pivot = start + sz / 2;
  --end;
  sel_swap(pivot, end);
  pivot = end;
  --end;
  while (start != end) {
    if (comp(*start, *pivot)) {
      ++start;
    } else {
      sel_swap(start, end);
      --end;
    }
  }
  if (comp(*pivot, *end)) {
    sel_swap(pivot, end);
  } else {
    ++end;
    sel_swap(pivot, end);
  }
  return end;
}

template <typename T, typename C>
__device__ inline T* jpp_partition(T* start, T* end, C comp, long minSize, long maxSize) {
  for (;;) {
    T* mid = jpp_part_step(start, end, comp);
    long sz = mid - start;
    if (minSize <= sz && sz <= maxSize) return mid;
    if (sz > maxSize) {
      end = mid;
      continue;
    }
    sz += 1;
    start = mid + 1;
    minSize -= sz;
    maxSize -= sz;
    if (minSize == 0) return start;
  }
}

}  // namespace jpp

#endif  // JPP_SELECT_H
