// heap_emul.hpp -- bit-exact emulation of the libstdc++ (GCC 11-13, bits/stl_heap.h, bits/stl_algo.h)
// binary-heap and selection mechanics that the reference's search runs on its result and candidate
// arrays: std::push_heap / pop_heap / make_heap (hnswalg_slim.h:331-332,353-354,408-423,438-440;
// std::priority_queue in hnswalg.h:337-344) and std::nth_element (hnswalg_slim.h:2126-2127).
//
// The comparators of the reference look at .first only (hnswalg_slim.h:169-183), so WHICH of several
// equal-distance entries is popped / evicted / selected is decided by these sift mechanics; emulating
// them step for step is what makes neighbour ids bit-exact on tie-heavy (integer-valued) data.
// Usable from one lane of a wavefront on LDS arrays, and from host code (unit-tested against the
// real std:: algorithms in tests/test_heap_emul.py via csrc/selftest.cpp).
#pragma once
#include "hd.hpp"

namespace hs {

struct Pair {
  float d;
  uint32_t id;
};

// comp(a,b) of compare_by_first (max-heap on distance): a.first < b.first      (hnswalg_slim.h:169-175)
struct LessD { HS_HD bool operator()(const Pair &a, const Pair &b) const { return a.d < b.d; } };
// comp(a,b) of compare_by_first_rev (min-heap on distance): a.first > b.first  (hnswalg_slim.h:177-183)
struct GreaterD { HS_HD bool operator()(const Pair &a, const Pair &b) const { return a.d > b.d; } };

// std::__push_heap
// `A a` is a pointer to Pair-like elements or any handle with a[i] convertible to Pair and assignable from Pair
// (slimq_search.hip keeps a small heap in registers, one element per lane).
template <class A, class C>
HS_HD void sift_up(A a, long hole, long top, Pair v, C comp) {
  long parent = (hole - 1) / 2;
  while (hole > top && comp(a[parent], v)) {
    a[hole] = a[parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  a[hole] = v;
}
// std::push_heap on a[0..n): a[n-1] is the new element.
template <class A, class C>
HS_HD void push_heap(A a, long n, C comp) {
  Pair v = a[n - 1];
  sift_up(a, n - 1, 0, v, comp);
}
// std::__adjust_heap
template <class A, class C>
HS_HD void adjust_heap(A a, long hole, long len, Pair v, C comp) {
  const long top = hole;
  long child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (comp(a[child], a[child - 1])) child--;
    a[hole] = a[child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    a[hole] = a[child - 1];
    hole = child - 1;
  }
  sift_up(a, hole, top, v, comp);
}
// std::pop_heap on a[0..n): afterwards a[n-1] holds the old root and a[0..n-1) is a heap.
template <class A, class C>
HS_HD void pop_heap(A a, long n, C comp) {
  if (n > 1) {
    Pair v = a[n - 1];
    a[n - 1] = a[0];
    adjust_heap(a, 0, n - 1, v, comp);
  }
}
// std::make_heap
template <class P, class C>
HS_HD void make_heap(P *a, long n, C comp) {
  if (n < 2) return;
  long parent = (n - 2) / 2;
  while (true) {
    Pair v = a[parent];
    adjust_heap(a, parent, n, v, comp);
    if (parent == 0) return;
    parent--;
  }
}

template <class P>
HS_HD void swap_el(P *a, long i, long j) {
  Pair t = a[i];
  a[i] = a[j];
  a[j] = t;
}

// std::__introselect as called by std::nth_element(a, a+nth, a+n, comp).
template <class P, class C>
HS_HD void nth_element(P *a, long nth, long n, C comp) {
  if (n == 0 || nth == n) return;
  long first = 0, last = n;
  long depth = 0;  // 2 * floor(log2(n))
  for (long t = n; t > 1; t >>= 1) depth++;
  depth *= 2;
  while (last - first > 3) {
    if (depth == 0) {
      // std::__heap_select(first, nth+1, last) ; iter_swap(first, nth)
      long middle = nth + 1;
      make_heap(a + first, middle - first, comp);
      for (long i = middle; i < last; i++)
        if (comp(a[i], a[first])) {
          Pair v = a[i];
          a[i] = a[first];
          adjust_heap(a + first, 0, middle - first, v, comp);
        }
      swap_el(a, first, nth);
      return;
    }
    depth--;
    // std::__unguarded_partition_pivot
    long mid = first + (last - first) / 2;
    {  // std::__move_median_to_first(result=first, a=first+1, b=mid, c=last-1)
      long ia = first + 1, ib = mid, ic = last - 1;
      if (comp(a[ia], a[ib])) {
        if (comp(a[ib], a[ic])) swap_el(a, first, ib);
        else if (comp(a[ia], a[ic])) swap_el(a, first, ic);
        else swap_el(a, first, ia);
      } else if (comp(a[ia], a[ic])) swap_el(a, first, ia);
      else if (comp(a[ib], a[ic])) swap_el(a, first, ic);
      else swap_el(a, first, ib);
    }
    long lo = first + 1, hi = last;
    while (true) {  // std::__unguarded_partition(first+1, last, pivot=first)
      while (comp(a[lo], a[first])) lo++;
      hi--;
      while (comp(a[first], a[hi])) hi--;
      if (!(lo < hi)) break;
      swap_el(a, lo, hi);
      lo++;
    }
    long cut = lo;
    if (cut <= nth) first = cut;
    else last = cut;
  }
  // std::__insertion_sort(first, last)
  if (first == last) return;
  for (long i = first + 1; i != last; i++) {
    Pair v = a[i];
    if (comp(v, a[first])) {
      for (long j = i; j > first; j--) a[j] = a[j - 1];
      a[first] = v;
    } else {  // std::__unguarded_linear_insert
      long pos = i, next = i - 1;
      while (comp(v, a[next])) {
        a[pos] = a[next];
        pos = next;
        next--;
      }
      a[pos] = v;
    }
  }
}

// std::sort(a, a+n, comp) (bits/stl_algo.h: __introsort_loop + __final_insertion_sort), step for step, so that WHICH of
// several equal-key entries ends up where is libstdc++'s choice -- convertFromHNSW sorts (distance, id) pairs by distance
// only (hnswalg_slim.h:963-966, 1049-1052) before PruneByHeuristic, and the kept set depends on that order.
// The recursion of __introsort_loop (right part recursed, left part looped) works on disjoint ranges, so an explicit stack
// visiting them in any order leaves the same array.  Returns false if the stack would overflow (never for n < 2^31).
template <class P, class C>
HS_HD void insertion_sort_range(P *a, long first, long last, C comp, bool guarded) {
  // std::__insertion_sort (guarded) / std::__unguarded_insertion_sort
  if (first == last) return;
  for (long i = guarded ? first + 1 : first; i != last; i++) {
    Pair v = a[i];
    if (guarded && comp(v, a[first])) {
      for (long j = i; j > first; j--) a[j] = a[j - 1];
      a[first] = v;
    } else {  // std::__unguarded_linear_insert
      long pos = i, next = i - 1;
      while (comp(v, a[next])) {
        a[pos] = a[next];
        pos = next;
        next--;
      }
      a[pos] = v;
    }
  }
}
template <class P, class C>
HS_HD bool std_sort(P *a, long n, C comp) {
  if (n < 2) return true;
  long depth0 = 0;
  for (long t = n; t > 1; t >>= 1) depth0++;
  depth0 *= 2;
  long stk_first[64], stk_last[64], stk_depth[64];
  int sp = 0;
  stk_first[0] = 0; stk_last[0] = n; stk_depth[0] = depth0; sp = 1;
  while (sp > 0) {
    sp--;
    long first = stk_first[sp], last = stk_last[sp], depth = stk_depth[sp];
    while (last - first > 16) {
      if (depth == 0) {
        // std::__partial_sort(first, last, last): __heap_select (= make_heap, nothing beyond middle) + __sort_heap
        make_heap(a + first, last - first, comp);
        for (long m = last - first; m > 1; m--) pop_heap(a + first, m, comp);
        break;
      }
      depth--;
      // std::__unguarded_partition_pivot
      const long mid = first + (last - first) / 2;
      {
        const long ia = first + 1, ib = mid, ic = last - 1;
        if (comp(a[ia], a[ib])) {
          if (comp(a[ib], a[ic])) swap_el(a, first, ib);
          else if (comp(a[ia], a[ic])) swap_el(a, first, ic);
          else swap_el(a, first, ia);
        } else if (comp(a[ia], a[ic])) swap_el(a, first, ia);
        else if (comp(a[ib], a[ic])) swap_el(a, first, ic);
        else swap_el(a, first, ib);
      }
      long lo = first + 1, hi = last;
      while (true) {
        while (comp(a[lo], a[first])) lo++;
        hi--;
        while (comp(a[first], a[hi])) hi--;
        if (!(lo < hi)) break;
        swap_el(a, lo, hi);
        lo++;
      }
      // __introsort_loop(cut, last, depth) ; last = cut
      if (sp >= 64) return false;
      stk_first[sp] = lo; stk_last[sp] = last; stk_depth[sp] = depth; sp++;
      last = lo;
    }
  }
  // std::__final_insertion_sort
  if (n > 16) {
    insertion_sort_range(a, 0, 16, comp, true);
    insertion_sort_range(a, 16, n, comp, false);
  } else {
    insertion_sort_range(a, 0, n, comp, true);
  }
  return true;
}

}  // namespace hs
