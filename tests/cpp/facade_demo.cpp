// facade_demo.cpp — exercises the C++ facade exactly like the reference's tracker loop does
// (demo.cpp:137-143 construct + knobs, :252-268 reinit + optimize).  Inputs written by tests/test_gpu_facade.py:
//   argv[1] model dir (model.npz + pose_prior.txt), argv[2] frame.bin (int N; N*3 doubles xyz; N ints labels;
//   start state: 10 w, 3 p, 24*9 R col-major), argv[3] output.bin (3V cloud, 3 p, K w, J*9 R, final cost).
#include <cstdio>
#include <vector>

#include "ark/AvatarOptimizer.h"

int main(int argc, char** argv) {
    if (argc < 4) { std::fprintf(stderr, "usage: facade_demo model_dir frame.bin out.bin\n"); return 2; }
    const ark::AvatarModel model(argv[1]);
    ark::Avatar ava(model);
    FILE* f = std::fopen(argv[2], "rb");
    if (!f) { std::perror("frame"); return 2; }
    int N = 0;
    if (std::fread(&N, sizeof(int), 1, f) != 1) return 2;
    ark::CloudType dataCloud;
    dataCloud.resize(3, N);
    ark::VectorXi labels(N);
    const int J = model.numJoints(), K = model.numShapeKeys();
    bool okr = std::fread(dataCloud.data(), sizeof(double), 3 * (size_t)N, f) == 3 * (size_t)N &&
               std::fread(labels.data(), sizeof(int), N, f) == (size_t)N &&
               std::fread(ava.w.data(), sizeof(double), K, f) == (size_t)K && std::fread(ava.p.data(), sizeof(double), 3, f) == 3;
    for (int j = 0; okr && j < J; ++j) okr = std::fread(ava.r[j].data(), sizeof(double), 9, f) == 9;
    std::fclose(f);
    if (!okr) { std::fprintf(stderr, "short frame file\n"); return 2; }
    ava.update();

    ark::CameraIntrin intrin;
    std::vector<int> partMap(J);
    for (int j = 0; j < J; ++j) partMap[j] = j;
    ark::AvatarOptimizer avaOpt(ava, intrin, ark::Size(1280, 720), J, partMap);
    avaOpt.betaPose = 0.05;      // demo.cpp:139-143
    avaOpt.betaShape = 0.12;
    avaOpt.optimize(dataCloud, labels, 1, 4);

    FILE* o = std::fopen(argv[3], "wb");
    std::fwrite(ava.cloud.data(), sizeof(double), ava.cloud.size(), o);
    std::fwrite(ava.p.data(), sizeof(double), 3, o);
    std::fwrite(ava.w.data(), sizeof(double), K, o);
    for (int j = 0; j < J; ++j) std::fwrite(ava.r[j].data(), sizeof(double), 9, o);
    std::fwrite(&avaOpt.lastStats.final_cost, sizeof(double), 1, o);
    std::fclose(o);
    std::printf("facade_demo: N=%d correspondences=%d cost %.6f -> %.6f\n", N, avaOpt.lastStats.num_correspondences,
                avaOpt.lastStats.initial_cost, avaOpt.lastStats.final_cost);
    return 0;
}
