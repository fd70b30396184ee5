// ark/FrameTracker.h — the per-frame protocol of the reference's trackers (demo.cpp:215-290, live-demo.cpp:335-432) over
// ark::AvatarOptimizer: interval subsampling of the labelled XYZ map inside the foreground bounding box, the
// tracking-loss / reinitialisation policy, the per-frame ICP budgets and the temporal warm start (the avatar state simply
// carries over between frames).  SURVEY.md §8 row f3.  Header-only, no OpenCV: images are plain row-major buffers.
//
// Inputs per frame are what the reference's perception front-end produces (out of scope here): an XYZ map (height x width x 3
// float, camera coordinates, cv::Vec3f layout) and a per-pixel body-part mask (height x width uint8, 255 = background), plus
// the foreground bounding box (bgsub.topLeft / bgsub.botRight, inclusive).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "AvatarOptimizer.h"

namespace ark {

class FrameTracker {
   public:
    struct Rect { int top = 0, left = 0, bottom = 0, right = 0; };   // inclusive, like bgsub.topLeft / botRight

    explicit FrameTracker(AvatarOptimizer& ava_opt) : avaOpt(ava_opt), ava(ava_opt.ava) {}

    /** Every `interval`-th pixel of the bounding box that carries a body-part label (demo.cpp:216-250); y negated (:245).
     *  Returns the number of points; a label >= numParts is fatal exactly like demo.cpp:236-243. */
    size_t subsample(const float* xyz, const std::uint8_t* part_mask, int width, const Rect& box, CloudType& dataCloud,
                     VectorXi& dataPartLabels) const {
        size_t cnz = 0;
        for (int r = box.top; r <= box.bottom; r += interval) {
            const std::uint8_t* partptr = part_mask + (size_t)r * width;
            for (int c = box.left; c <= box.right; c += interval) cnz += partptr[c] != 255;
        }
        dataCloud.resize(3, cnz);
        dataPartLabels.assign(cnz, 0);
        size_t i = 0;
        for (int r = box.top; r <= box.bottom; r += interval) {
            const float* ptr = xyz + (size_t)r * width * 3;
            const std::uint8_t* partptr = part_mask + (size_t)r * width;
            for (int c = box.left; c <= box.right; c += interval) {
                if (partptr[c] == 255) continue;
                if (partptr[c] >= avaOpt.numParts) {
                    std::fprintf(stderr, "FATAL: body part prediction %d is invalid, since there are only %d body parts\n", (int)partptr[c], avaOpt.numParts);
                    std::exit(1);
                }
                dataCloud(0, i) = ptr[3 * c];
                dataCloud(1, i) = -ptr[3 * c + 1];
                dataCloud(2, i) = ptr[3 * c + 2];
                dataPartLabels[i] = partptr[c];
                ++i;
            }
        }
        return cnz;
    }

    /** One tracked frame.  Returns true if the avatar was fitted, false if tracking was declared lost (too few body
     *  pixels; the next fitted frame reinitialises: live-demo.cpp:335-340, :379-383). */
    bool process(const float* xyz, const std::uint8_t* part_mask, int width, int height, const Rect& box) {
        (void)height;
        const size_t cnz = subsample(xyz, part_mask, width, box, dataCloud, dataPartLabels);
        // demo.cpp:225 skips a sparse frame; live-demo.cpp:379-383 also asks for a reinitialisation, which is the policy kept
        // here (documented deviation from demo.cpp).  An EMPTY frame is never fitted whatever reinitCnz says: the centroid
        // below divides by cnz.
        bool part_missing = false;       // live-demo.cpp:376-380: the FIRST fit wants every body part seen (initialPerPartCnz pixels at interval 1)
        if (firstTime && initialPerPartCnz > 0) {
            std::vector<size_t> partCnz((size_t)avaOpt.numParts, 0);
            for (size_t i = 0; i < cnz; ++i) ++partCnz[(size_t)dataPartLabels[i]];
            size_t mn = partCnz.empty() ? 0 : partCnz[0];
            for (size_t v : partCnz) mn = v < mn ? v : mn;
            const int need = initialPerPartCnz / (interval * interval);
            part_missing = mn < (size_t)(need > 1 ? need : 1);
        }
        if (cnz == 0 || part_missing || cnz < (size_t)(reinitCnz / (interval * interval))) {
            reinit = true;
            return false;
        }
        int icpIters = frameICPIters;
        if (reinit) {                                                   // demo.cpp:252-265
            double cen[3] = {0, 0, 0};
            for (size_t i = 0; i < cnz; ++i) for (int c = 0; c < 3; ++c) cen[c] += dataCloud(c, i);
            for (int c = 0; c < 3; ++c) ava.p(c) = cen[c] / (double)cnz;
            ava.w.assign(ava.w.size(), 0.0);
            for (int i = 1; i < ava.model.numJoints(); ++i) ava.r[i].setIdentity();
            // AngleAxis(pi, (0, 1, 0)).toRotationMatrix(): written out (cos(pi) and sin(pi) leave rounding residue)
            Matrix3d r0;
            r0(0, 0) = -1.0; r0(2, 2) = -1.0;
            ava.r[0] = r0;
            reinit = false;
            ava.update();
            icpIters = firstTime ? initialICPIters : reinitICPIters;    // live-demo.cpp:417-418
            firstTime = false;
        }
        avaOpt.optimize(dataCloud, dataPartLabels, icpIters, numThreads);
        ++framesFitted;
        return true;
    }

    int interval = 12;            // demo.cpp:58   --data-interval
    int frameICPIters = 3;        // demo.cpp:63   --frame-icp-iters
    int reinitICPIters = 6;       // demo.cpp:66   --reinit-icp-iters
    int initialICPIters = 6;      // live-demo.cpp:80 (demo.cpp has one budget for both)
    int reinitCnz = 1000;         // demo.cpp:71   --min-points
    int initialPerPartCnz = 0;    // live-demo.cpp:89-90 --initial-per-part-thresh (80 there); 0 = demo.cpp, which has no per-part check
    int numThreads = 4;
    bool reinit = true;           // demo.cpp:151
    bool firstTime = true;        // live-demo.cpp:256
    long framesFitted = 0;

    AvatarOptimizer& avaOpt;
    Avatar& ava;

   private:
    CloudType dataCloud;
    VectorXi dataPartLabels;
};

}  // namespace ark
