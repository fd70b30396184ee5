// ark/AvatarOptimizer.h — the reference's `ark::AvatarOptimizer` (include/AvatarOptimizer.h:11-61) on top of the C ABI.
#pragma once
#include <vector>

#include "Avatar.h"

namespace ark {
/** Optimize avatar to fit a point cloud */
class AvatarOptimizer {
   public:
    /** Same signature as the reference; `intrin` and `image_size` do not influence optimize() there either
     *  (renderer constructed but unused, AvatarOptimizer.cpp:1271,:1369-1385).  `part_map` must have >= numJoints
     *  entries (AvatarOptimizer.cpp:1229) and is kept by reference, like the reference does. */
    AvatarOptimizer(Avatar& ava, const CameraIntrin& intrin, const Size& image_size, int num_parts, const std::vector<int>& part_map)
        : ava(ava), intrin(intrin), imageSize(image_size), numParts(num_parts), partMap(part_map) {
        r.resize(ava.model.numJoints());
    }
    ~AvatarOptimizer() { if (ctx) avt_ctx_destroy(ctx); }
    // holds references (ava, intrin, partMap) and owns a device context: not copyable (the reference's class holds the same
    // references; copying it would alias the avatar it optimises)
    AvatarOptimizer(const AvatarOptimizer&) = delete;
    AvatarOptimizer& operator=(const AvatarOptimizer&) = delete;

    /** Begin full optimization on the target data cloud (AvatarOptimizer.cpp:1246-1517).  Precondition as in the
     *  reference: ava.update() has been called; postcondition: ava.p, ava.w, ava.r updated and ava.update()d. */
    void optimize(const CloudType& data_cloud, const VectorXi& data_part_labels, int icp_iters = 1, int num_threads = 4) {
        const int N = (int)data_cloud.cols(), J = ava.model.numJoints();
        if ((int)data_part_labels.size() != N) { std::fprintf(stderr, "avatar (MI355X): labels/cloud size mismatch\n"); std::exit(1); }
        if (!ctx || N > capacity) {
            if (ctx) avt_ctx_destroy(ctx);
            capacity = N > 65536 ? N : 65536;
            ARK_AVT_CHECK(avt_ctx_create(ava.device, ava.model.handle, numParts, partMap.data(), capacity, 1, &ctx));
        }
        for (int i = 0; i < J; ++i) r[i] = rotationToQuaternion(ava.r[i]);       // :1250-1254
        avt_options o;
        avt_options_default(&o);
        o.beta_pose = betaPose; o.beta_shape = betaShape; o.nn_step = nnStep; o.max_iters_per_icp = maxItersPerICP;
        o.enable_occlusion = enableOcclusion ? 1 : 0; o.icp_iters = icp_iters; o.num_threads = num_threads;
        o.function_tolerance = functionTolerance;
        std::vector<double> q(4 * (size_t)J);
        for (int i = 0; i < J; ++i) for (int c = 0; c < 4; ++c) q[4 * i + c] = r[i].c[c];
        // :1497 ava.update(): the launch sequence ends with exactly that update; its outputs come back with the fit, in the call's one synchronisation
        ava.cloud.resize(3, ava.model.numPoints());
        ava.jointPos.resize(3, J);
        ava.jointTrans.resize(12, J);
        if (icp_iters >= 1) {
            ARK_AVT_CHECK(avt_optimize_posed(ctx, data_cloud.data(), data_part_labels.data(), N, &o, ava.p.data(), q.data(), ava.w.data(), &lastStats,
                                             ava.cloud.data(), ava.jointPos.data(), ava.jointTrans.data()));
        } else {      // (no ICP iteration: no closing update in the sequence)
            ARK_AVT_CHECK(avt_optimize(ctx, data_cloud.data(), data_part_labels.data(), N, &o, ava.p.data(), q.data(), ava.w.data(), &lastStats));
            ARK_AVT_CHECK(avt_get_posed(ctx, 0, ava.cloud.data(), ava.jointPos.data(), ava.jointTrans.data()));
        }
        for (int i = 0; i < J; ++i) {
            for (int c = 0; c < 4; ++c) r[i].c[c] = q[4 * i + c];
            ava.r[i] = quaternionToRotation(r[i]);                               // :1494-1496
        }
    }

    /** Rotation representation size */
    static const int ROT_SIZE = 4;
    /** Optimization parameter r */
    std::vector<Quaterniond> r;
    /** Cost function component weights */
    double betaPose = 0.1, betaShape = 1.0;
    /** NN matching step size (unused in the inverted NN mode the reference runs) */
    int nnStep = 20;
    /** maximum inner iterations per ICP */
    int maxItersPerICP = 10;
    /** Whether to elimiate occluded points before NN matching */
    bool enableOcclusion = true;
    /** Not a member of the reference's class: the value its optimize() hard-codes as options.function_tolerance
     *  (AvatarOptimizer.cpp:1333) - a step that lowers the objective by no more than this fraction ends the inner iterations
     *  of the ICP iteration; 0 = always maxItersPerICP iterations (include/avt.h, avt_options::function_tolerance) */
    double functionTolerance = 1e-4;

    Avatar& ava;
    const CameraIntrin& intrin;
    Size imageSize;
    int numParts;
    const std::vector<int>& partMap;

    /** final cost / #correspondences of the last call, so a caller can implement the reference's tracking-loss
     *  reinit policy (demo.cpp:225,251-266) */
    avt_stats lastStats{};

   private:
    avt_ctx* ctx = nullptr;
    int capacity = 0;
};
}  // namespace ark
