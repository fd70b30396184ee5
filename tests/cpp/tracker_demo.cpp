// tracker_demo.cpp — the reference's tracking loop (demo.cpp:137-151 setup, :215-290 per frame) over ark::FrameTracker.
// Inputs written by tests/test_gpu_tracker.py / bench.py:
//   argv[1] model dir (model.npz + pose_prior.txt)
//   argv[2] sequence.bin: int nframes, width, height, interval, frameICP, reinitICP, reinitCnz; then per frame:
//           int top, left, bottom, right; width*height*3 float xyz; width*height uint8 mask
//   argv[3] output.bin: per frame: int fitted; 3V doubles cloud, 3 p, K w   (cloud etc. as left by the last fit)
//   argv[4] (optional) repeat count for timing: the sequence minus its first frame is replayed that many times
//   argv[5] (optional) AvatarOptimizer::functionTolerance (default: the reference's 1e-4, AvatarOptimizer.cpp:1333; 0 = every fit runs its full budget)
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ark/FrameTracker.h"

int main(int argc, char** argv) {
    if (argc < 4) { std::fprintf(stderr, "usage: tracker_demo model_dir sequence.bin out.bin [timing repeats]\n"); return 2; }
    const ark::AvatarModel model(argv[1]);
    ark::Avatar ava(model);
    FILE* f = std::fopen(argv[2], "rb");
    if (!f) { std::perror("sequence"); return 2; }
    int hdr[7];
    if (std::fread(hdr, sizeof(int), 7, f) != 7) return 2;
    const int nframes = hdr[0], W = hdr[1], H = hdr[2];
    struct Frame { ark::FrameTracker::Rect box; std::vector<float> xyz; std::vector<std::uint8_t> mask; };
    std::vector<Frame> frames(nframes);
    for (auto& fr : frames) {
        int b[4];
        fr.xyz.resize((size_t)W * H * 3); fr.mask.resize((size_t)W * H);
        if (std::fread(b, sizeof(int), 4, f) != 4 || std::fread(fr.xyz.data(), sizeof(float), fr.xyz.size(), f) != fr.xyz.size() ||
            std::fread(fr.mask.data(), 1, fr.mask.size(), f) != fr.mask.size()) { std::fprintf(stderr, "short sequence file\n"); return 2; }
        fr.box.top = b[0]; fr.box.left = b[1]; fr.box.bottom = b[2]; fr.box.right = b[3];
    }
    std::fclose(f);
    const int J = model.numJoints(), K = model.numShapeKeys();
    ark::CameraIntrin intrin;
    std::vector<int> partMap(J);
    for (int j = 0; j < J; ++j) partMap[j] = j;
    ark::AvatarOptimizer avaOpt(ava, intrin, ark::Size(W, H), J, partMap);
    avaOpt.betaPose = 0.05;      // demo.cpp:139-143
    avaOpt.betaShape = 0.12;
    if (argc > 5) avaOpt.functionTolerance = std::atof(argv[5]);
    ark::FrameTracker tracker(avaOpt);
    tracker.interval = hdr[3]; tracker.frameICPIters = hdr[4]; tracker.reinitICPIters = tracker.initialICPIters = hdr[5]; tracker.reinitCnz = hdr[6];
    FILE* o = std::fopen(argv[3], "wb");
    for (auto& fr : frames) {
        const int fitted = tracker.process(fr.xyz.data(), fr.mask.data(), W, H, fr.box) ? 1 : 0;
        std::fwrite(&fitted, sizeof(int), 1, o);
        if (fitted) {
            std::fwrite(ava.cloud.data(), sizeof(double), ava.cloud.size(), o);
            std::fwrite(ava.p.data(), sizeof(double), 3, o);
            std::fwrite(ava.w.data(), sizeof(double), K, o);
        }
    }
    std::fclose(o);
    const int reps = argc > 4 ? std::atoi(argv[4]) : 0;
    if (reps > 0 && nframes > 1) {
        long n = 0, gn = 0;
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; ++r)
            for (int i = 1; i < nframes; ++i) {
                const bool fit = tracker.process(frames[i].xyz.data(), frames[i].mask.data(), W, H, frames[i].box);
                n += fit; gn += fit ? avaOpt.lastStats.gn_iterations : 0;
            }
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        std::printf("tracker_demo timing: %ld frames, %.4f ms per frame, %.2f GN iterations per frame\n", n, ms / (double)n, (double)gn / (double)n);
    }
    std::printf("tracker_demo: %d frames, %ld fitted, reinit=%d\n", nframes, tracker.framesFitted, (int)tracker.reinit);
    return 0;
}
