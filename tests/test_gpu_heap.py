"""The flat kernel's candidate-heap code (csrc/flat_search.hip: fh_push, fh_pop, fh_pop_wave) against a step-for-step Python
restatement of libstdc++'s std::push_heap / std::pop_heap with compare_by_first_rev (hnswalg_slim.h:177-183), on tie-heavy
operation sequences: the array layout after every sequence and every popped root must be identical."""
import random

import pytest

from hsutil import load_product

pytestmark = pytest.mark.gpu


def push_ref(a, x):   # std::__push_heap with comp(a, b) = a.first > b.first
    a.append(x)
    hole = len(a) - 1
    parent = (hole - 1) // 2
    while hole > 0 and a[parent][0] > x[0]:
        a[hole] = a[parent]
        hole = parent
        parent = (hole - 1) // 2
    a[hole] = x


def pop_ref(a):       # std::pop_heap + pop_back: __adjust_heap(first, 0, len, value)
    n = len(a)
    root = a[0]
    if n > 1:
        v = a[n - 1]
        length = n - 1
        hole = child = 0
        while child < (length - 1) // 2:
            child = 2 * (child + 1)
            if a[child][0] > a[child - 1][0]:
                child -= 1
            a[hole] = a[child]
            hole = child
        if (length & 1) == 0 and child == (length - 2) // 2:
            child = 2 * (child + 1)
            a[hole] = a[child - 1]
            hole = child - 1
        parent = (hole - 1) // 2
        while hole > 0 and a[parent][0] > v[0]:
            a[hole] = a[parent]
            hole = parent
            parent = (hole - 1) // 2
        a[hole] = v
    a.pop()
    return root


@pytest.mark.parametrize("wave_pop", [True, False])
@pytest.mark.parametrize("lds_slots", [1024, 64])
@pytest.mark.parametrize("key_range,p_pop", [(8, 0.3), (1000, 0.45), (3, 0.2)])
def test_heap_ops_match_libstdcxx(wave_pop, lds_slots, key_range, p_pop):
    hs = load_product()
    rng = random.Random(1234 + key_range)
    for trial in range(3):
        ops, ref, want_pops, idc = [], [], [], 0
        for _ in range(700):
            if ref and rng.random() < p_pop:
                ops.append((1, 0.0, 0))
                want_pops.append(pop_ref(ref))
            else:
                idc += 1
                x = (float(rng.randint(0, key_range)), idc)
                ops.append((0, x[0], x[1]))
                push_ref(ref, x)
        heap, pops = hs.debug_heap_ops(ops, wave_pop=wave_pop, lds_slots=lds_slots)
        assert pops == want_pops, f"trial {trial}: popped roots differ"
        assert heap == ref, f"trial {trial}: heap layout differs"
