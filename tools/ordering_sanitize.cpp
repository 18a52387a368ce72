// Sanitizer run of the built-in nested dissection with the team BFS (CPU only; round 3):
//   g++ -O1 -g -std=c++17 -fsanitize=thread -pthread -Iinclude -Isparse-matrix-factorization-library_amd/csrc tools/ordering_sanitize.cpp \
//       sparse-matrix-factorization-library_amd/csrc/sf_symbolic.cpp -o /tmp/nd_tsan && SF_ANALYZE_THREADS=8 /tmp/nd_tsan 50
//   (and with -fsanitize=address,undefined; the printed hash of the permutation must not depend on SF_ANALYZE_THREADS).  Clean at 50^3 / 62^3.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "sf_symbolic.h"
using sf::Long;
int main(int argc, char** argv) {
    const int g = argc > 1 ? atoi(argv[1]) : 50;
    const Long n = (Long)g * g * g;
    std::vector<Long> Cp(n + 1, 0), Ci, perm(n);
    for (Long j = 0; j < n; ++j) {
        const Long x = j % g, y = (j / g) % g, z = j / (g * g);
        Ci.push_back(j);
        if (x + 1 < g) Ci.push_back(j + 1);
        if (y + 1 < g) Ci.push_back(j + g);
        if (z + 1 < g) Ci.push_back(j + g * g);
        Cp[j + 1] = (Long)Ci.size();
    }
    if (sf::graph_nd_perm(n, Cp.data(), Ci.data(), 64, perm.data())) return 2;
    unsigned long long h = 1469598103934665603ull;
    for (Long i = 0; i < n; ++i) { h ^= (unsigned long long)perm[i]; h *= 1099511628211ull; }
    printf("n %lld perm hash %llu\n", (long long)n, h);
    return 0;
}
