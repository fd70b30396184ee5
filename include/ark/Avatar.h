// ark/Avatar.h — the reference's `ark::AvatarModel` / `ark::Avatar` (include/Avatar.h:64-220) on top of the C ABI
// (include/avt.h).  Same class names, member names, defaults and call protocol; containers from ark/Types.h stand in
// for Eigen.  All numerical work happens in libavatar_hip.so (hand-written HIP, gfx950).
#pragma once
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../avt.h"
#include "Npz.h"
#include "Types.h"

// the reference aborts on invalid input (_ARK_ASSERT*, Util.h:7-51): the facade keeps `void` APIs and exits
#define ARK_AVT_CHECK(call)                                                                       \
    do {                                                                                          \
        if ((call) != 0) {                                                                        \
            std::fprintf(stderr, "avatar (MI355X): %s\n  at %s:%d\n", avt_last_error(), __FILE__, __LINE__); \
            std::exit(1);                                                                         \
        }                                                                                         \
    } while (0)

namespace ark {

struct GaussianMixture {  // GaussianMixture.h: data only; residual/Jacobian live on the device
    int nComps = -1, nDims = 0;
    VectorXd weight, mean /* nComps x nDims row-major */, cov /* nComps x nDims x nDims */;
    int numComponents() const { return nComps; }
    void load(const std::string& path) {  // text layout of GaussianMixture.cpp:12-58
        std::ifstream ifs(path);
        if (!ifs) { std::fprintf(stderr, "Warning: pose prior file at %s does not exist or cannot be read\n", path.c_str()); nComps = -1; return; }
        ifs >> nComps >> nDims;
        weight.resize(nComps); mean.resize((size_t)nComps * nDims); cov.resize((size_t)nComps * nDims * nDims);
        for (auto& v : weight) ifs >> v;
        for (auto& v : mean) ifs >> v;
        for (auto& v : cov) ifs >> v;
    }
};

struct AvatarModel {
    /** @param model_dir directory holding model.npz (SMPL layout, AvatarModel.cpp:26-104) and optionally
     *  pose_prior.txt; `limit_one_joint_per_point` only exists for the legacy text format and is rejected. */
    explicit AvatarModel(const std::string& model_dir = "", bool limit_one_joint_per_point = false) : MODEL_DIR(model_dir) {
        if (limit_one_joint_per_point) { std::fprintf(stderr, "avatar (MI355X): limit_one_joint_per_point is not supported\n"); std::exit(1); }
        if (model_dir.empty()) { std::fprintf(stderr, "avatar (MI355X): no model directory given (the reference's data download is not bundled)\n"); std::exit(1); }
        std::map<std::string, npz::Array> z;
        try { z = npz::load(model_dir + "/model.npz"); }
        catch (const std::exception& e) { std::fprintf(stderr, "avatar (MI355X): %s\n", e.what()); std::exit(1); }
        const npz::Array &vt = z.at("v_template"), &f = z.at("f"), &kt = z.at("kintree_table"), &jr = z.at("J_regressor"),
                         &wt = z.at("weights"), &sd = z.at("shapedirs");
        const int V = (int)vt.shape[0], J = (int)kt.shape[1], F = (int)f.shape[0], K = (int)sd.shape[2];
        parent.resize(J);
        for (int j = 0; j < J; ++j) parent[j] = (int)kt.at(j);
        parent[0] = -1;
        baseCloud.resize(3 * (size_t)V);
        for (size_t i = 0; i < baseCloud.size(); ++i) baseCloud[i] = vt.at(i);
        mesh.a.resize(3 * (size_t)F);
        for (size_t i = 0; i < mesh.a.size(); ++i) mesh.a[i] = (int)f.at(i);
        keyClouds.resize((size_t)3 * V * K);       // (3V x K) column-major  <- shapedirs (V,3,K) row-major
        for (int v = 0; v < V; ++v)
            for (int c = 0; c < 3; ++c)
                for (int k = 0; k < K; ++k) keyClouds[(size_t)k * 3 * V + 3 * v + c] = sd.at(((size_t)v * 3 + c) * K + k);
        // weights (V,J) -> J x V CSC ; J_regressor (J,V) -> V x J CSC  (sparseView(): exact zeros dropped)
        w_colptr.assign(V + 1, 0);
        for (int v = 0; v < V; ++v) {
            for (int j = 0; j < J; ++j) {
                const double x = wt.at((size_t)v * J + j);
                if (x != 0.0) { w_row.push_back(j); w_val.push_back(x); }
            }
            w_colptr[v + 1] = (int)w_row.size();
        }
        r_colptr.assign(J + 1, 0);
        for (int j = 0; j < J; ++j) {
            for (int v = 0; v < V; ++v) {
                const double x = jr.at((size_t)j * V + v);
                if (x != 0.0) { r_row.push_back(v); r_val.push_back(x); }
            }
            r_colptr[j + 1] = (int)r_row.size();
        }
        posePrior.load(model_dir + "/pose_prior.txt");
        nV = V; nJ = J; nK = K; nF = F;
        createHandle();
    }
    ~AvatarModel() { if (handle) avt_model_destroy(handle); }
    AvatarModel(const AvatarModel&) = delete;
    AvatarModel& operator=(const AvatarModel&) = delete;

    inline int numJoints() const { return nJ; }
    inline int numPoints() const { return nV; }
    inline int numShapeKeys() const { return nK; }
    inline int numFaces() const { return nF; }
    inline bool hasMesh() const { return nF > 0; }
    inline bool hasPosePrior() const { return posePrior.nComps >= 0; }

    MeshType mesh;
    VectorXi parent;
    /** main (largest-weight) joint per point: assignedJoints[i][0].second */
    VectorXi mainJoint;
    GaussianMixture posePrior;
    VectorXd baseCloud;        // 3V
    VectorXd keyClouds;        // 3V x K column-major
    CloudType initialJointPos; // 3 x J
    VectorXd jointShapeReg;    // 3J x K column-major
    bool useJointShapeRegressor = true;
    const std::string MODEL_DIR;

    avt_model* handle = nullptr;   // C-ABI handle

   private:
    int nV = 0, nJ = 0, nK = 0, nF = 0;
    VectorXi w_colptr, w_row, r_colptr, r_row;
    VectorXd w_val, r_val;
    void createHandle() {
        avt_model_desc d{};
        d.num_points = nV; d.num_joints = nJ; d.num_shape_keys = nK; d.num_faces = nF;
        d.base_cloud = baseCloud.data(); d.key_clouds = keyClouds.data(); d.parent = parent.data(); d.mesh = mesh.data();
        d.weights_colptr = w_colptr.data(); d.weights_row = w_row.data(); d.weights_val = w_val.data();
        d.jreg_colptr = r_colptr.data(); d.jreg_row = r_row.data(); d.jreg_val = r_val.data();
        d.prior_ncomps = posePrior.nComps > 0 ? posePrior.nComps : 0; d.prior_ndims = posePrior.nDims;
        d.prior_weight = posePrior.weight.data(); d.prior_mean = posePrior.mean.data(); d.prior_cov = posePrior.cov.data();
        ARK_AVT_CHECK(avt_model_create(&d, &handle));
        mainJoint.resize(nV);
        ARK_AVT_CHECK(avt_model_main_joint(handle, mainJoint.data()));
        initialJointPos.resize(3, nJ);
        jointShapeReg.resize((size_t)3 * nJ * nK);
        ARK_AVT_CHECK(avt_model_joint_regression(handle, initialJointPos.data(), jointShapeReg.data()));
    }
};

/** An avatar instance (Avatar.h:155-220): state w, p, r -> update() -> cloud, jointPos, jointTrans. */
class Avatar {
   public:
    explicit Avatar(const AvatarModel& model) : model(model) {
        w.assign(model.numShapeKeys(), 0.0);          // Avatar.cpp:12-20
        r.assign(model.numJoints(), Matrix3d());
    }
    /** Copies carry the state (w, p, r and the derived clouds) like the reference's plain value type; the device context is
     *  NOT shared: a copy creates its own on its first update() (one owner per avt_ctx, no double free). */
    Avatar(const Avatar& o) : model(o.model), cloud(o.cloud), w(o.w), p(o.p), r(o.r), jointPos(o.jointPos), jointTrans(o.jointTrans), device(o.device) {}
    Avatar& operator=(const Avatar& o) {
        if (this != &o) { cloud = o.cloud; w = o.w; p = o.p; r = o.r; jointPos = o.jointPos; jointTrans = o.jointTrans; device = o.device; }
        return *this;      // (`model` is a reference: both sides must already refer to the same model, as with the reference's type)
    }
    ~Avatar() { if (ctx) avt_ctx_destroy(ctx); }

    /** Update joints and skin points from the current shape and pose (Avatar.cpp:22-75). */
    void update() {
        if (!ctx) {
            VectorXi ident(model.numJoints());
            for (int j = 0; j < model.numJoints(); ++j) ident[j] = j;
            ARK_AVT_CHECK(avt_ctx_create(device, model.handle, model.numJoints(), ident.data(), 64, 1, &ctx));
        }
        cloud.resize(3, model.numPoints());
        jointPos.resize(3, model.numJoints());
        jointTrans.resize(12, model.numJoints());
        ARK_AVT_CHECK(avt_lbs_update(ctx, 1, w.data(), p.data(), r[0].data(), cloud.data(), jointPos.data(), jointTrans.data()));
    }

    const AvatarModel& model;
    CloudType cloud;
    VectorXd w;
    Vector3d p;
    std::vector<Matrix3d> r;            // contiguous 9 doubles each, column-major
    CloudType jointPos;
    MatrixNX<12> jointTrans;
    int device = 0;                     // HIP device used by update()

   private:
    avt_ctx* ctx = nullptr;
};

}  // namespace ark
