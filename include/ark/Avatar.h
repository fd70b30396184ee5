// ark/Avatar.h — the reference's `ark::AvatarModel` / `ark::Avatar` (include/Avatar.h:64-220) on top of the C ABI
// (include/avt.h).  Same class names, member names, defaults and call protocol; containers from ark/Types.h stand in
// for Eigen.  All numerical work happens in libavatar_hip.so (hand-written HIP, gfx950).
#pragma once
#include <dirent.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <limits>
#include <map>
#include <memory>
#include <random>
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

/** Joint ids of the SMPL skeleton in the model's breadth-first order (Avatar.h:27-59); alignToJoints() addresses three of them by name. */
namespace SmplJoint {
enum { ROOT_PELVIS = 0, L_HIP, R_HIP, SPINE1, L_KNEE, R_KNEE, SPINE2, L_ANKLE, R_ANKLE, SPINE3, L_FOOT, R_FOOT, NECK, L_COLLAR, R_COLLAR, HEAD,
       L_SHOULDER, R_SHOULDER, L_ELBOW, R_ELBOW, L_WRIST, R_WRIST, L_HAND, R_HAND, _COUNT };
}

/** GaussianMixture.h.  The optimiser's residual / Jacobian of the prior live on the device (avt_model_create factors the components there); the
 *  host-side members below serve Avatar::pdf() and Avatar::randomize() and are factored on first use. */
struct GaussianMixture {
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
        factored = false;
    }

    /** Mixture density at x (GaussianMixture.cpp:83-93), as the reference evaluates it: the constants are normalised by the smallest
     *  determinant (:64-76), and the exponent is |L (x - mu)|^2 with L = chol(cov^-1) - the reference multiplies by L, not by its
     *  transpose, so this is not the Mahalanobis distance; restated as written. */
    double pdf(const VectorXd& x) const {
        factor();
        const int n = nDims;
        double prob = 0.0;
        for (int i = 0; i < nComps; ++i) {
            const double* L = &prec_cho[(size_t)i * n * n];
            double s = 0.0;
            for (int r = 0; r < n; ++r) {
                double a = 0.0;
                for (int c = 0; c <= r; ++c) a += L[(size_t)r * n + c] * (x[c] - mean[(size_t)i * n + c]);
                s += a * a;
            }
            prob += consts[i] * std::exp(-0.5 * s);
        }
        return prob;
    }

    /** A sample (GaussianMixture.cpp:116-133).  As written there: the component loop has no break, so the component is the LAST one whose
     *  running remainder is non-positive - with positive weights that is always the last component; the draw itself is mean + chol(cov) z
     *  (the reference writes `r *= cov_cho`, a product whose dimensions only agree because they are dynamic: the intent is restated). */
    VectorXd sample() const {
        factor();
        double randf = random_util::uniform(0.0f, 1.0f);
        int component = nComps - 1;
        for (int i = 0; i < nComps; ++i) { randf -= weight[i]; if (randf <= 0) component = i; }
        const int n = nDims;
        VectorXd z(n), out(n);
        for (int i = 0; i < n; ++i) z[i] = random_util::randn();
        const double* L = &cov_cho[(size_t)component * n * n];
        for (int r = 0; r < n; ++r) {
            double a = mean[(size_t)component * n + r];
            for (int c = 0; c <= r; ++c) a += L[(size_t)r * n + c] * z[c];
            out[r] = a;
        }
        return out;
    }

    /** chol(cov), chol(cov^-1) (lower triangular, row-major) and the constants of every component (GaussianMixture.cpp:22-76) */
    mutable VectorXd cov_cho, prec_cho, consts, consts_log;

   private:
    mutable bool factored = false;
    static bool cholesky(const double* A, int n, double* L) {      // A = L L^T, row-major, upper part of L zero
        std::fill(L, L + (size_t)n * n, 0.0);
        for (int j = 0; j < n; ++j) {
            double d = A[(size_t)j * n + j];
            for (int k = 0; k < j; ++k) d -= L[(size_t)j * n + k] * L[(size_t)j * n + k];
            if (!(d > 0.0)) return false;
            const double ljj = std::sqrt(d);
            L[(size_t)j * n + j] = ljj;
            for (int i = j + 1; i < n; ++i) {
                double a = A[(size_t)i * n + j];
                for (int k = 0; k < j; ++k) a -= L[(size_t)i * n + k] * L[(size_t)j * n + k];
                L[(size_t)i * n + j] = a / ljj;
            }
        }
        return true;
    }
    void factor() const {
        if (factored || nComps <= 0) return;
        const int n = nDims;
        const size_t nn = (size_t)n * n;
        cov_cho.assign(nn * nComps, 0.0); prec_cho.assign(nn * nComps, 0.0); consts.assign(nComps, 0.0); consts_log.assign(nComps, 0.0);
        const double pi = 3.14159265358979323846, sqrt_2_pi_n = std::pow(2 * pi, n * 0.5), log_sqrt_2_pi_n = n * 0.5 * std::log(2 * pi);
        double minDet = std::numeric_limits<double>::max();
        VectorXd Linv(nn), prec(nn);
        for (int i = 0; i < nComps; ++i) {
            consts_log[i] = std::log(weight[i]) - log_sqrt_2_pi_n;
            consts[i] = weight[i] / sqrt_2_pi_n;
            double* L = &cov_cho[i * nn];
            if (!cholesky(&cov[i * nn], n, L)) { std::fprintf(stderr, "avatar (MI355X): pose prior covariance %d is not positive definite\n", i); std::exit(1); }
            // cov^-1 = L^-T L^-1
            std::fill(Linv.begin(), Linv.end(), 0.0);
            for (int c = 0; c < n; ++c) {
                Linv[(size_t)c * n + c] = 1.0 / L[(size_t)c * n + c];
                for (int r = c + 1; r < n; ++r) {
                    double a = 0.0;
                    for (int k = c; k < r; ++k) a -= L[(size_t)r * n + k] * Linv[(size_t)k * n + c];
                    Linv[(size_t)r * n + c] = a / L[(size_t)r * n + r];
                }
            }
            for (int r = 0; r < n; ++r)
                for (int c = 0; c <= r; ++c) {
                    double a = 0.0;
                    for (int k = r; k < n; ++k) a += Linv[(size_t)k * n + r] * Linv[(size_t)k * n + c];
                    prec[(size_t)r * n + c] = prec[(size_t)c * n + r] = a;
                }
            if (!cholesky(prec.data(), n, &prec_cho[i * nn])) { std::fprintf(stderr, "avatar (MI355X): pose prior precision %d is not positive definite\n", i); std::exit(1); }
            double det = 1.0;
            for (int d = 0; d < n; ++d) det *= L[(size_t)d * n + d];
            minDet = std::min(det, minDet);
            consts[i] /= det;
            consts_log[i] -= std::log(det);
        }
        for (int i = 0; i < nComps; ++i) { consts[i] *= minDet; consts_log[i] += std::log(minDet); }
        factored = true;
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

    /** Randomize pose and shape according to the PCA shape space and the GMM pose prior (Avatar.cpp:77-126), in the reference's order of draws:
     *  K normal shape coefficients; the pose from posePrior.sample() - which draws from the library's OWN generators, not from the seeded one
     *  (so, as in the reference, a seed does not pin the pose; random_util::reseed does) -; then root position x in [-1, 1), y in [-0.5, 0.5),
     *  z in [2.2, 4.5), the root's turn about the vertical pi + U[-pi/3, pi/3), and a N(0, 0.2) rad perturbation about a random axis.
     *  seed = -1: keep the generator's state. */
    void randomize(bool randomize_pose = true, bool randomize_shape = true, bool randomize_root_pos_rot = true, uint32_t seed = (uint32_t)-1) {
        thread_local static std::mt19937 rg(std::random_device{}());
        if (~seed) rg.seed(seed);
        if (randomize_shape)
            for (int i = 0; i < model.numShapeKeys(); ++i) w[i] = random_util::randn(rg);
        if (randomize_pose) {
            if (!model.hasPosePrior()) { std::fprintf(stderr, "avatar (MI355X): randomize(pose) needs pose_prior.txt\n"); std::exit(1); }
            const VectorXd samp = model.posePrior.sample();
            for (int i = 0; i < model.numJoints() - 1; ++i) {
                const double ax = samp[3 * i], ay = samp[3 * i + 1], az = samp[3 * i + 2], angle = std::sqrt(ax * ax + ay * ay + az * az);
                r[i + 1] = Matrix3d::AngleAxis(angle, ax / angle, ay / angle, az / angle);
            }
        }
        if (randomize_root_pos_rot) {
            const double pi = 3.14159265358979323846;
            p(0) = random_util::uniform(rg, -1.0, 1.0);
            p(1) = random_util::uniform(rg, -0.5, 0.5);
            p(2) = random_util::uniform(rg, 2.2, 4.5);
            const double angle_up = random_util::uniform(rg, -pi / 3., pi / 3.) + pi;
            const double theta = random_util::uniform(rg, 0, 2 * pi), phi = random_util::uniform(rg, -pi / 2, pi / 2);
            // fromSpherical(1, theta, phi) (AvatarHelpers.cpp:55-59)
            const double sx = std::sin(phi) * std::cos(theta), sy = std::cos(phi), sz = std::sin(phi) * std::sin(theta);
            const double angle_perturb = random_util::randn(rg, 0.0, 0.2);
            r[0] = Matrix3d::AngleAxis(angle_perturb, sx, sy, sz) * Matrix3d::AngleAxis(angle_up, 0., 1., 0.);
        }
    }

    /** The SMPL pose parameters: axis-angle of every joint but the root, 3 (J - 1) numbers (Avatar.cpp:128-137). */
    VectorXd smplParams() const {
        VectorXd res((size_t)(model.numJoints() - 1) * 3);
        for (int i = 1; i < model.numJoints(); ++i) {
            double angle; Vector3d axis;
            rotationToAngleAxis(r[i], angle, axis);
            for (int c = 0; c < 3; ++c) res[(size_t)(i - 1) * 3 + c] = axis(c) * angle;
        }
        return res;
    }

    /** GMM likelihood of the current joint rotations (Avatar.cpp:139). */
    double pdf() const { return model.posePrior.pdf(smplParams()); }

    /** Pose (and the first shape coefficient) from target joint positions, 3 x 24 (Avatar.cpp:141-193): the root goes to joint 0, its rotation
     *  takes the rest direction pelvis -> spine1 to the target's; every other joint's bone (parent -> joint) is turned from its rest
     *  direction to the target's, r[i] = that turn relative to the PARENT's turn; w[0] from the mean bone-length ratio.  NaN columns leave
     *  their joint at the identity (root: keeps p).  Quirks kept: r[i] names the turn of the bone that ENDS in joint i, and a joint whose
     *  target is missing inherits its parent's turn for its children. */
    void alignToJoints(const CloudType& pos) {
        if ((int)pos.cols() != SmplJoint::_COUNT || model.numJoints() != SmplJoint::_COUNT) { std::fprintf(stderr, "avatar (MI355X): alignToJoints wants 24 joint positions on a 24-joint model\n"); std::exit(1); }
        auto col = [](const CloudType& m, int i) { Vector3d v; for (int c = 0; c < 3; ++c) v(c) = m(c, (size_t)i); return v; };
        const CloudType& ij = model.initialJointPos;
        const Vector3d vr = col(ij, SmplJoint::SPINE1) - col(ij, SmplJoint::ROOT_PELVIS), vrt = col(pos, SmplJoint::SPINE1) - col(pos, SmplJoint::ROOT_PELVIS);
        if (!std::isnan(pos(0, 0))) p = col(pos, 0);
        if (!std::isnan(vr(0)) && !std::isnan(vrt(0))) r[0] = rotationFromTwoVectors(vr, vrt);
        else r[0].setIdentity();
        std::vector<Matrix3d> rotTrans(pos.cols());
        rotTrans[0] = r[0];
        double scaleAvg = 0.0;
        for (int i = 1; i < (int)pos.cols(); ++i)
            scaleAvg += norm(col(pos, i) - col(pos, model.parent[i])) / norm(col(ij, i) - col(ij, model.parent[i]));
        scaleAvg /= (pos.cols() - 1.0);
        const double baseScale = norm(col(ij, SmplJoint::SPINE2) - col(ij, SmplJoint::ROOT_PELVIS)) * (scaleAvg - 1.0);
        const double PC1_DIST_FACT = 32.0;      // shape key 0 per metre of width (Avatar.cpp:171-173)
        w[0] = baseScale * PC1_DIST_FACT;
        if (std::isnan(w[0])) w[0] = 1.5;
        for (int i = 1; i < (int)pos.cols(); ++i) {
            rotTrans[i] = rotTrans[model.parent[i]];
            if (!std::isnan(pos(0, (size_t)i))) {
                rotTrans[i] = rotationFromTwoVectors(col(ij, i) - col(ij, model.parent[i]), col(pos, i) - col(pos, model.parent[i]));
                r[i] = transpose(rotTrans[model.parent[i]]) * rotTrans[i];
            } else {
                r[i].setIdentity();
            }
        }
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
