// Test infrastructure: the oracle's restatement of libstdc++'s std::sort (oracle/bp_oracle.c: oracle_std_sort_desc) against
// the real std::sort of the host toolchain, with the comparator serial_relative uses (bp.hpp:472-482: descending key).
// Equal keys, all-equal keys, NaNs, sorted / reversed / organ-pipe inputs and inputs that exhaust the recursion budget are
// the point: on those the result depends on the exact sequence of swaps.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

extern "C" void oracle_std_sort_desc(int *v, long n, const double *key);

static long check(const std::vector<double> &key, std::vector<int> start) {
    std::vector<int> a = start, b = start;
    std::sort(a.begin(), a.end(), [&](int x, int y) { return key[x] > key[y]; });
    oracle_std_sort_desc(b.data(), (long)b.size(), key.data());
    return a == b ? 0 : 1;
}

extern "C" long std_sort_port_mismatches(long rounds, unsigned seed) {
    std::mt19937_64 g(seed);
    long bad = 0, cases = 0;
    const int sizes[] = {0, 1, 2, 3, 15, 16, 17, 18, 31, 32, 33, 64, 100, 144, 257, 441, 1000, 1600, 4097, 10000};
    for (long r = 0; r < rounds; ++r)
        for (int n : sizes) {
            for (int kind = 0; kind < 10; ++kind) {
                std::vector<double> key((size_t)n);
                std::vector<int> start((size_t)n);
                for (int i = 0; i < n; ++i) start[(size_t)i] = i;
                std::uniform_real_distribution<double> u(-5, 5);
                switch (kind) {
                    case 0: for (auto &k : key) k = u(g); break;                                   // distinct
                    case 1: for (auto &k : key) k = 2.19722457733622;  break;                      // all equal (uniform priors)
                    case 2: for (auto &k : key) k = (double)(g() % 3); break;                      // many ties
                    case 3: for (int i = 0; i < n; ++i) key[(size_t)i] = i; break;                  // ascending (worst for "descending")
                    case 4: for (int i = 0; i < n; ++i) key[(size_t)i] = -i; break;                 // already descending
                    case 5: for (int i = 0; i < n; ++i) key[(size_t)i] = std::min(i, n - 1 - i); break;  // organ pipe
                    case 6: for (auto &k : key) k = (g() % 11 == 0) ? NAN : u(g); break;           // NaNs: comparator not a weak order
                    case 7: for (auto &k : key) k = (g() % 2) ? INFINITY : -INFINITY; break;
                    case 8: for (int i = 0; i < n; ++i) key[(size_t)i] = (i % 2) ? i : -i; break;
                    default: for (auto &k : key) k = std::floor(u(g)); break;
                }
                if (kind >= 5 && n > 1) std::shuffle(start.begin(), start.end(), g);             // a carried-over order, not 0 .. n-1
                bad += check(key, start);
                ++cases;
            }
        }
    // median-of-three killer sequences drive the quicksort to its depth limit -> the heapsort fallback must agree too
    for (int n : {64, 1000, 4096, 10000}) {
        std::vector<double> key((size_t)n);
        std::vector<int> start((size_t)n);
        for (int i = 0; i < n; ++i) start[(size_t)i] = i;
        const int k = n / 2;
        for (int i = 1; i <= k; ++i) {
            if (i % 2) { key[(size_t)i - 1] = -(double)i; key[(size_t)i] = -(double)(k + i); }
            key[(size_t)(k + i - 1)] = -(double)(2 * i);
        }
        bad += check(key, start);
        ++cases;
    }
    std::printf("std::sort port: %ld cases, %ld mismatches\n", cases, bad);
    return bad;
}
