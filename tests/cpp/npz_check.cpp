// npz_check.cpp — CPU-only check of include/ark/Npz.h: prints shape + sum + first/last element of every array.
#include <cstdio>
#include "ark/Npz.h"
int main(int argc, char** argv) {
    std::map<std::string, ark::npz::Array> z;
    try {
        z = ark::npz::load(argv[1]);
    } catch (const std::exception& e) {      // a damaged file must end here, not in a crash
        std::fprintf(stderr, "npz_check: %s\n", e.what());
        return 3;
    }
    for (auto& kv : z) {
        double s = 0;
        const size_t n = kv.second.size();
        for (size_t i = 0; i < n; ++i) s += kv.second.at(i);
        std::printf("%s %zu %.17g %.17g %.17g", kv.first.c_str(), n, s, n ? kv.second.at(0) : 0.0, n ? kv.second.at(n - 1) : 0.0);
        for (size_t d : kv.second.shape) std::printf(" %zu", d);
        std::printf("\n");
    }
    return 0;
}
