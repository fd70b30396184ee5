// avatar_methods_check.cpp — CPU-only check of the host-side members of ark::Avatar (include/ark/Avatar.h) that the trackers do not call but
// the reference's tools do (smplsynth.cpp:110-112,160): randomize (Avatar.cpp:77-126), smplParams (:128-137), pdf (:139, GaussianMixture.cpp:83-93),
// alignToJoints (:141-193).  Usage: avatar_methods_check <model_dir> <targets.txt: 24 x 3 joint positions, "nan nan nan" = missing>
// Prints named rows of numbers that tests/test_avatar_methods_cpu.py compares with numpy / the oracle.
#include <cmath>
#include <cstdio>
#include <fstream>

#include "ark/Avatar.h"

static void row(const char* name, const double* v, size_t n) {
    std::printf("%s", name);
    for (size_t i = 0; i < n; ++i) std::printf(" %.17g", v[i]);
    std::printf("\n");
}
static void rots(const char* name, const ark::Avatar& a) {
    std::printf("%s", name);
    for (const auto& R : a.r) for (int i = 0; i < 9; ++i) std::printf(" %.17g", R.data()[i]);      // column-major 3x3 blocks
    std::printf("\n");
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    const ark::AvatarModel model(argv[1]);
    ark::Avatar ava(model);
    // 1. randomize: seeded shape / root draws twice (must repeat), the pose from the prior with the library generators reseeded
    ark::random_util::reseed(7);
    ava.randomize(true, true, true, 1234);
    row("w1", ava.w.data(), ava.w.size()); row("p1", ava.p.data(), 3); rots("r1", ava);
    const ark::VectorXd sp = ava.smplParams();
    row("smpl1", sp.data(), sp.size());
    std::printf("pdf1 %.17g\n", ava.pdf());
    ark::Avatar b(model);
    ark::random_util::reseed(7);
    b.randomize(true, true, true, 1234);
    row("w2", b.w.data(), b.w.size()); row("p2", b.p.data(), 3); rots("r2", b);
    // only the root: pose and shape stay
    ark::Avatar c(model);
    c.randomize(false, false, true, 99);
    row("w3", c.w.data(), c.w.size()); row("p3", c.p.data(), 3); rots("r3", c);
    // 2. the mixture's factors, for the dense numpy check
    row("consts", model.posePrior.consts.data(), model.posePrior.consts.size());
    row("consts_log", model.posePrior.consts_log.data(), model.posePrior.consts_log.size());
    // 3. alignToJoints on the given targets
    ark::CloudType pos; pos.resize(3, 24);
    std::ifstream tf(argv[2]);
    for (int i = 0; i < 24; ++i) for (int k = 0; k < 3; ++k) { std::string s; tf >> s; pos(k, (size_t)i) = (s == "nan") ? std::nan("") : std::stod(s); }
    ark::Avatar d(model);
    d.alignToJoints(pos);
    row("w4", d.w.data(), d.w.size()); row("p4", d.p.data(), 3); rots("r4", d);
    return 0;
}
