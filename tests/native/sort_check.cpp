// Test harness (CPU): fray_amd/csrc/dev_sort.hpp must order every input exactly as std::sort of this toolchain does,
// ties included (the comparator only looks at the key, so equal keys expose the algorithm's element moves).
#define FRAY_SORT_FN inline
static long g_heap_fallbacks = 0;
#define FRAY_SORT_COUNT_HEAP g_heap_fallbacks
#include "dev_sort.hpp"

#include <algorithm>
#include <cstdio>
#include <random>
#include <vector>

struct E { double key; int id; };

int main()
{
    std::mt19937 g(12345);
    long cases = 0;
    for (int n = 0; n <= 60; n++)
        for (int rep = 0; rep < 4000; rep++) {
            std::vector<E> v(n);
            const int mode = rep % 8;
            const int distinct = 1 + (int)(g() % (mode < 3 ? 3 : mode < 6 ? 12 : 1000));
            for (int i = 0; i < n; i++) v[i] = E{(double)(g() % distinct), i};
            if (mode == 6) std::sort(v.begin(), v.end(), [](const E& x, const E& y) { return x.key < y.key; });          // already sorted
            if (mode == 7) {                                                                                           // two sorted runs, as CsgOp builds them
                const int h = n ? (int)(g() % (n + 1)) : 0;
                std::sort(v.begin(), v.begin() + h, [](const E& x, const E& y) { return x.key < y.key; });
                std::sort(v.begin() + h, v.end(), [](const E& x, const E& y) { return x.key < y.key; });
            }
            if (mode == 5 && n > 2) {                                                                                  // organ pipe
                for (int i = 0; i < n; i++) v[i].key = i < n / 2 ? i : n - i;
            }
            if (mode == 4 && n >= 4 && n % 2 == 0) {                                                                   // Musser's median-of-three killer: drives introsort into its heapsort fallback
                const int k = n / 2;
                for (int i = 1; i <= k; i++) {
                    v[i - 1].key = (i % 2) ? i : k + i - 1;
                    v[k + i - 1].key = 2 * i;
                }
            }
            for (int i = 0; i < n; i++) v[i].id = i;
            std::vector<double> key(n);
            std::vector<unsigned char> a(n);
            for (int i = 0; i < n; i++) { key[i] = v[i].key; a[i] = (unsigned char)i; }
            std::sort(v.begin(), v.end(), [](const E& x, const E& y) { return x.key < y.key; });
            StdSort s{key.data(), a.data()};
            s.sort(n);
            for (int i = 0; i < n; i++)
                if (a[i] != v[i].id) { printf("MISMATCH n=%d rep=%d at %d: %d vs %d\n", n, rep, i, (int)a[i], v[i].id); return 1; }
            cases++;
        }
    printf("ok %ld cases, %ld heapsort fallbacks exercised\n", cases, g_heap_fallbacks);
    return 0;
}
