// ark/Avatar.h — the reference's `ark::AvatarModel` / `ark::Avatar` (include/Avatar.h:64-220) on top of the C ABI
// (include/avt.h).  Same class names, member names, defaults and call protocol; containers from ark/Types.h stand in
// for Eigen.  All numerical work happens in libavatar_hip.so (hand-written HIP, gfx950).
#pragma once
#include <dirent.h>

#include <algorithm>
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
    /** @param model_dir directory holding model.npz (SMPL layout, AvatarModel.cpp:26-104) and optionally pose_prior.txt; without a
     *  model.npz the reference's deprecated ad-hoc text format is read (AvatarModel.cpp:128-288: skeleton.txt, model.pcd,
     *  shapekey/ *.pcd, joint_shape_regressor.txt or joint_regressor.txt, mesh.txt).
     *  @param limit_one_joint_per_point the optimiser's forward model binds every point to its largest-weight joint only
     *  (AvatarModel.cpp:190-196).  Like the reference it takes effect in the legacy format only: with a model.npz it is ignored (:23-127 never
     *  read it) - a warning says so; avt_model_desc::limit_one_joint_per_point is there for callers that want it on any model.  update() keeps all weights. */
    explicit AvatarModel(const std::string& model_dir = "", bool limit_one_joint_per_point = false)
        : MODEL_DIR(model_dir), limitOneJointPerPoint(limit_one_joint_per_point) {
        if (model_dir.empty()) { std::fprintf(stderr, "avatar (MI355X): no model directory given (the reference's data download is not bundled)\n"); std::exit(1); }
        if (!std::ifstream(model_dir + "/model.npz")) { loadLegacy(model_dir); return; }
        if (limitOneJointPerPoint) {
            std::fprintf(stderr, "avatar (MI355X): limit_one_joint_per_point is ignored for model.npz, as in the reference (AvatarModel.cpp:23-127)\n");
            applyLimitOneJoint = false;
        }
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
    const bool limitOneJointPerPoint = false;
    bool applyLimitOneJoint = true;         // false: the flag was given with a model.npz, where the reference ignores it

    avt_model* handle = nullptr;   // C-ABI handle

   private:
    int nV = 0, nJ = 0, nK = 0, nF = 0;
    VectorXi w_colptr, w_row, r_colptr, r_row;
    VectorXd w_val, r_val;
    VectorXd legacyJsrBase, legacyJsr;      // joint_shape_regressor.txt of the legacy format (3J; 3J x K column-major), else empty

    [[noreturn]] static void legacyFail(const std::string& what) { std::fprintf(stderr, "ERROR: %s\n", what.c_str()); std::exit(1); }

    /** loadPCDToPointVectorFast (AvatarHelpers.cpp:13-52): WIDTH = point count, DATA must be ascii, then 3 numbers per point. */
    static VectorXd loadPCD(const std::string& path) {
        std::ifstream pcd(path);
        int nPoints = -1;
        std::string label, rest;
        while (pcd >> label) {
            if (label == "DATA") {
                if (nPoints < 0) legacyFail("invalid PCD file at " + path + ": no WIDTH field before data");
                pcd >> label;
                if (label != "ascii") legacyFail("non-ascii PCD not supported! File " + path);
                break;
            } else if (label == "WIDTH") { pcd >> nPoints; std::getline(pcd, rest); }
            else std::getline(pcd, rest);
        }
        if (!pcd || nPoints < 0) legacyFail("invalid PCD file at " + path + ": unexpected EOF");
        VectorXd out((size_t)nPoints * 3);
        for (auto& v : out) pcd >> v;
        if (!pcd) legacyFail("invalid PCD file at " + path + ": unexpected EOF");
        return out;
    }

    /** The reference's old ad-hoc format (AvatarModel.cpp:128-288). */
    void loadLegacy(const std::string& dir) {
        std::fprintf(stderr, "WARNING: Using deprecated ad-hoc SMPL model format; please use the SMPL .npz model (model.npz)\n");
        baseCloud = loadPCD(dir + "/model.pcd");
        std::ifstream skel(dir + "/skeleton.txt");
        if (!skel) legacyFail("Avatar model is invalid, skeleton file not found");
        int J = 0, V = 0;
        skel >> J >> V;
        if (!skel || J <= 0 || V <= 0 || (size_t)V * 3 != baseCloud.size()) legacyFail("Invalid avatar skeleton file (joint / point counts)");
        parent.assign(J, -1);
        for (int i = 0; i < J; ++i) {                     // joints in topologically sorted order
            int id = 0, par = 0; std::string name; double x, y, z;
            skel >> id >> par >> name >> x >> y >> z;
            if (!skel || id < 0 || id >= J) legacyFail("Invalid avatar skeleton file: joint table");
            parent[id] = par;
        }
        parent[0] = -1;
        w_colptr.assign(V + 1, 0);
        for (int v = 0; v < V; ++v) {
            int n = 0;
            skel >> n;
            std::vector<std::pair<int, double>> ent((size_t)(n > 0 ? n : 0));
            for (auto& e : ent) skel >> e.first >> e.second;
            if (!skel) legacyFail("Invalid avatar skeleton file: joint assignments are not present");
            std::sort(ent.begin(), ent.end());            // compressed-column order: joints ascending within a point
            for (auto& e : ent) { w_row.push_back(e.first); w_val.push_back(e.second); }
            w_colptr[v + 1] = (int)w_row.size();
        }
        // shape keys: one PCD per key; the reference takes them in directory order (unspecified) - here sorted by file name
        std::vector<std::string> keyFiles = listDir(dir + "/shapekey");
        std::sort(keyFiles.begin(), keyFiles.end());
        const int K = (int)keyFiles.size();
        if (K == 0) std::fprintf(stderr, "WARNING: no shape key directory found for avatar\n");
        keyClouds.resize((size_t)3 * V * K);
        for (int k = 0; k < K; ++k) {
            const VectorXd kc = loadPCD(dir + "/shapekey/" + keyFiles[k]);
            if (kc.size() != (size_t)3 * V) legacyFail("shape key " + keyFiles[k] + " has another point count than the model");
            std::copy(kc.begin(), kc.end(), keyClouds.begin() + (size_t)k * 3 * V);
        }
        r_colptr.assign(J + 1, 0);
        std::ifstream jsr(dir + "/joint_shape_regressor.txt");
        if (jsr) {                                        // AvatarModel.cpp:231-243: base (3J), then the 3J x K matrix row by row
            int nk = 0;
            jsr >> nk;
            if (nk != K) legacyFail("joint_shape_regressor.txt and the shape key directory disagree on the number of keys");
            legacyJsrBase.resize((size_t)3 * J); legacyJsr.resize((size_t)3 * J * K);
            for (auto& v : legacyJsrBase) jsr >> v;
            for (int i = 0; i < 3 * J; ++i)
                for (int k = 0; k < K; ++k) jsr >> legacyJsr[(size_t)k * 3 * J + i];
            if (!jsr) legacyFail("joint_shape_regressor.txt: unexpected EOF");
            useJointShapeRegressor = true;
        } else {
            std::ifstream jr(dir + "/joint_regressor.txt");
            if (jr) {                                     // AvatarModel.cpp:246-262: per joint a list of (point, value)
                int nj = 0;
                jr >> nj;
                if (nj != J) legacyFail("joint_regressor.txt: joint count differs from the skeleton");
                for (int j = 0; j < J; ++j) {
                    int n = 0;
                    jr >> n;
                    std::vector<std::pair<int, double>> ent((size_t)(n > 0 ? n : 0));
                    for (auto& e : ent) jr >> e.first >> e.second;
                    std::sort(ent.begin(), ent.end());
                    for (auto& e : ent) { r_row.push_back(e.first); r_val.push_back(e.second); }
                    r_colptr[j + 1] = (int)r_row.size();
                }
                if (!jr) legacyFail("joint_regressor.txt: unexpected EOF");
            } else {
                std::fprintf(stderr, "WARNING: neither joint regressor nor joint shape regressor found, model may be inaccurate with nonzero shapekey weights\n");
            }
            // (the reference then runs update() through shaped * jointRegressor and leaves jointShapeReg unset for the optimiser; here the
            // regressor is folded into jointShapeRegBase / jointShapeReg like the npz branch does - the same joints, and the optimiser works)
            useJointShapeRegressor = false;
        }
        std::ifstream meshFile(dir + "/mesh.txt");
        int F = 0;
        if (meshFile) {
            meshFile >> F;
            mesh.a.resize((size_t)3 * (F > 0 ? F : 0));
            for (auto& v : mesh.a) meshFile >> v;
            if (!meshFile) legacyFail("mesh.txt: unexpected EOF");
        } else {
            std::fprintf(stderr, "WARNING: mesh not found, maybe you are using an older version of avatar data files? Some functions will not work.\n");
        }
        posePrior.load(dir + "/pose_prior.txt");
        nV = V; nJ = J; nK = K; nF = F > 0 ? F : 0;
        createHandle();
    }

    static std::vector<std::string> listDir(const std::string& path) {
        std::vector<std::string> out;
        if (DIR* d = opendir(path.c_str())) {
            while (dirent* e = readdir(d)) {
                const std::string n = e->d_name;
                if (n != "." && n != "..") out.push_back(n);
            }
            closedir(d);
        }
        return out;
    }

    void createHandle() {
        avt_model_desc d{};
        d.num_points = nV; d.num_joints = nJ; d.num_shape_keys = nK; d.num_faces = nF;
        d.base_cloud = baseCloud.data(); d.key_clouds = keyClouds.data(); d.parent = parent.data(); d.mesh = mesh.data();
        d.weights_colptr = w_colptr.data(); d.weights_row = w_row.data(); d.weights_val = w_val.data();
        d.jreg_colptr = r_colptr.data(); d.jreg_row = r_row.data(); d.jreg_val = r_val.data();
        d.prior_ncomps = posePrior.nComps > 0 ? posePrior.nComps : 0; d.prior_ndims = posePrior.nDims;
        d.prior_weight = posePrior.weight.data(); d.prior_mean = posePrior.mean.data(); d.prior_cov = posePrior.cov.data();
        d.limit_one_joint_per_point = (limitOneJointPerPoint && applyLimitOneJoint) ? 1 : 0;
        if (!legacyJsr.empty()) { d.joint_shape_reg_base = legacyJsrBase.data(); d.joint_shape_reg = legacyJsr.data(); }
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
