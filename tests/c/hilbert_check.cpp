// Host check of pybnesian_amd/csrc/kde_kernels.hpp: hilbert_key - the sort key of the pruned sweeps at three / four key dimensions.  Exhaustive over a
// small grid: the keys are a bijection onto 0 .. 2^(n bits) - 1 and consecutive keys are NEIGHBOURING cells (one axis, one step) - the property
// that makes 16 consecutive rows a compact tile.  Built and run by tests/test_hilbert_cpu.py (hipcc, host code only).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../pybnesian_amd/csrc/kde_kernels.hpp"

int main() {
    const int cases[][2] = {{2, 5}, {3, 4}, {4, 3}, {3, 5}, {4, 4}};
    for (const auto& c : cases) {
        const int n = c[0], bits = c[1];
        const uint64_t cells = 1ull << (n * bits);
        std::vector<int64_t> at(cells, -1);
        for (uint64_t id = 0; id < cells; ++id) {
            uint32_t X[4] = {0, 0, 0, 0};
            for (int i = 0; i < n; ++i) X[i] = (uint32_t)((id >> (i * bits)) & ((1u << bits) - 1u));
            const uint32_t key = pbn::hilbert_key(X, n, bits);
            if (key >= cells || at[key] >= 0) { std::printf("n=%d bits=%d: key %u of cell %llu out of range or taken\n", n, bits, key, (unsigned long long)id); return 1; }
            at[key] = (int64_t)id;
        }
        for (uint64_t k = 0; k + 1 < cells; ++k) {
            int dist = 0;
            for (int i = 0; i < n; ++i) {
                const int a = (int)((at[k] >> (i * bits)) & ((1 << bits) - 1)), b = (int)((at[k + 1] >> (i * bits)) & ((1 << bits) - 1));
                dist += std::abs(a - b);
            }
            if (dist != 1) { std::printf("n=%d bits=%d: keys %llu and %llu are %d steps apart\n", n, bits, (unsigned long long)k, (unsigned long long)(k + 1), dist); return 1; }
        }
        std::printf("n=%d bits=%d: %llu cells, bijective, every step is one cell\n", n, bits, (unsigned long long)cells);
    }
    return 0;
}
