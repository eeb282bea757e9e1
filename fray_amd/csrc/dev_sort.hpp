// std::sort as libstdc++ 11 performs it (bits/stl_algo.h:1812-1974, bits/stl_heap.h), on a permutation: CsgOp::intersect
// sorts its operands' intersections by distance with std::sort (geometry.cpp:171-174), which is not stable, so WHICH of two
// equally distant intersections comes first -- coincident faces of CSG operands are equally distant -- is decided by the
// library's algorithm: introsort (median-of-three quicksort on ranges longer than 16, heapsort when the depth budget of
// 2 * floor(log2 n) runs out) and a final insertion sort.  Here the same element moves are made on an index array
// `a[0..n)` ordered by `key[a[i]]`; comp(x, y) is key[x] < key[y], the reference's comparator.
//
// FRAY_SORT_FN lets a host test compile this header with g++ and compare it with std::sort itself
// (tests/native/sort_check.cpp).
#pragma once
#ifndef FRAY_SORT_FN
#define FRAY_SORT_FN __device__ __forceinline__
#endif

struct StdSort {
    const double* key;
    unsigned char* a;
    FRAY_SORT_FN bool lt(unsigned char x, unsigned char y) const { return key[x] < key[y]; }
    FRAY_SORT_FN void swp(int i, int j) { unsigned char t = a[i]; a[i] = a[j]; a[j] = t; }

    // ---- bits/stl_heap.h
    FRAY_SORT_FN void push_heap(int first, int hole, int top, unsigned char value)
    {
        int parent = (hole - 1) / 2;
        while (hole > top && lt(a[first + parent], value)) {
            a[first + hole] = a[first + parent];
            hole = parent;
            parent = (hole - 1) / 2;
        }
        a[first + hole] = value;
    }
    FRAY_SORT_FN void adjust_heap(int first, int hole, int len, unsigned char value)
    {
        const int top = hole;
        int second = hole;
        while (second < (len - 1) / 2) {
            second = 2 * (second + 1);
            if (lt(a[first + second], a[first + (second - 1)])) second--;
            a[first + hole] = a[first + second];
            hole = second;
        }
        if ((len & 1) == 0 && second == (len - 2) / 2) {
            second = 2 * (second + 1);
            a[first + hole] = a[first + (second - 1)];
            hole = second - 1;
        }
        push_heap(first, hole, top, value);
    }
    FRAY_SORT_FN void heap_sort(int first, int last)        // __partial_sort(first, last, last): make_heap + sort_heap
    {
        const int len = last - first;
        if (len >= 2)
            for (int parent = (len - 2) / 2;; parent--) {
                adjust_heap(first, parent, len, a[first + parent]);
                if (parent == 0) break;
            }
        while (last - first > 1) {
            --last;
            const unsigned char value = a[last];
            a[last] = a[first];
            adjust_heap(first, 0, last - first, value);
        }
    }
    // ---- bits/stl_algo.h
    FRAY_SORT_FN void move_median_to_first(int result, int x, int y, int z)
    {
        if (lt(a[x], a[y])) {
            if (lt(a[y], a[z])) swp(result, y);
            else if (lt(a[x], a[z])) swp(result, z);
            else swp(result, x);
        } else if (lt(a[x], a[z])) swp(result, x);
        else if (lt(a[y], a[z])) swp(result, z);
        else swp(result, y);
    }
    FRAY_SORT_FN int unguarded_partition(int first, int last, int pivot)
    {
        for (;;) {
            while (lt(a[first], a[pivot])) ++first;
            --last;
            while (lt(a[pivot], a[last])) --last;
            if (!(first < last)) return first;
            swp(first, last);
            ++first;
        }
    }
    FRAY_SORT_FN void unguarded_linear_insert(int last)
    {
        const unsigned char val = a[last];
        int next = last - 1;
        while (lt(val, a[next])) { a[last] = a[next]; last = next; --next; }
        a[last] = val;
    }
    FRAY_SORT_FN void insertion_sort(int first, int last)
    {
        if (first == last) return;
        for (int i = first + 1; i != last; ++i) {
            if (lt(a[i], a[first])) {
                const unsigned char val = a[i];
                for (int k = i; k > first; k--) a[k] = a[k - 1];      // move_backward(first, i, i + 1)
                a[first] = val;
            } else unguarded_linear_insert(i);
        }
    }
    FRAY_SORT_FN void sort(int n)
    {
        if (n <= 0) return;
        int lg = 0;
        for (int v = n; v > 1; v >>= 1) lg++;
        // __introsort_loop(0, n, 2 * lg): the recursion on the right part becomes an explicit stack (the two parts are disjoint, their order is immaterial)
        int stFirst[16], stLast[16], stDepth[16], sp = 0;
        int first = 0, last = n, depth = 2 * lg;
        for (;;) {
            while (last - first > 16) {
                if (depth == 0) {
#ifdef FRAY_SORT_COUNT_HEAP
                    FRAY_SORT_COUNT_HEAP++;
#endif
                    heap_sort(first, last);
                    break;
                }
                --depth;
                const int mid = first + (last - first) / 2;
                move_median_to_first(first, first + 1, mid, last - 1);
                const int cut = unguarded_partition(first + 1, last, first);
                stFirst[sp] = cut; stLast[sp] = last; stDepth[sp] = depth; sp++;
                last = cut;
            }
            if (sp == 0) break;
            sp--;
            first = stFirst[sp]; last = stLast[sp]; depth = stDepth[sp];
        }
        // __final_insertion_sort
        if (n > 16) {
            insertion_sort(0, 16);
            for (int i = 16; i != n; ++i) unguarded_linear_insert(i);
        } else insertion_sort(0, n);
    }
};
