// ark/RTree.h — the reference's `ark::RTree` (include/RTree.h:12-184) re-created over the C ABI of avt_rtree.h:
// same class name, member names, defaults and call protocol for the inference side (loadFile, exportFile, predictBest,
// postProcess, numParts, partMap, leafData, leafBestMatch).  cv::Mat is replaced by the two plain row-major images
// below, cv::Point by ark::Point, Eigen::Matrix<double,2,Dynamic> by MatrixNX<2> (same column-major layout).
// The trainers (RTree.cpp:330-2955, train / trainFromAvatar / trainTransfer) are out of scope.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../avt.h"
#include "../avt_rtree.h"
#include "Types.h"

namespace ark {

struct Point {  // cv::Point stand-in
    int x = 0, y = 0;
    Point() {}
    Point(int x_, int y_) : x(x_), y(y_) {}
};

template <class T>
struct Image {  // row-major rows x cols, the layout of a continuous single-channel cv::Mat
    int rows = 0, cols = 0;
    std::vector<T> a;
    Image() {}
    Image(int r, int c, T fill = T()) : rows(r), cols(c), a((size_t)r * c, fill) {}
    T& at(int r, int c) { return a[(size_t)r * cols + c]; }
    T at(int r, int c) const { return a[(size_t)r * cols + c]; }
    T* ptr(int r) { return a.data() + (size_t)r * cols; }
    const T* ptr(int r) const { return a.data() + (size_t)r * cols; }
    T* data() { return a.data(); }
    const T* data() const { return a.data(); }
};
using ImageF = Image<float>;      // CV_32F depth, metres, 0 = background
using Image8 = Image<uint8_t>;    // CV_8U part labels, 255 = none

class RTree {
public:
    typedef std::vector<float> Distribution;
    /** Assumed depth of background (meters), RTree.cpp:325 */
    static constexpr float BACKGROUND_DEPTH = 20.f;

    struct RNode {  // RTree.h:28-41
        float u[2] = {0, 0}, v[2] = {0, 0};
        float thresh = 0;
        int lnode = -1, rnode = -1;
        int leafid = -1;
    };

    /** Create empty RTree with number of different parts */
    explicit RTree(int num_parts, int device = 0) : numParts(num_parts), device_(device) {}
    /** Load data from path */
    explicit RTree(const std::string& path, int device = 0) : device_(device) {
        if (!loadFile(path)) fprintf(stderr, "ERROR: RTree failed to initialize from %s\n", path.c_str());
    }
    ~RTree() { avt_rtree_destroy(h_); }
    RTree(const RTree&) = delete;
    RTree& operator=(const RTree&) = delete;

    bool loadFile(const std::string& path) {
        avt_rtree_destroy(h_);
        h_ = nullptr;
        if (avt_rtree_load(path.c_str(), device_, &h_) != 0) return false;
        pull();
        return true;
    }
    bool exportFile(const std::string& path) { return ensure() && avt_rtree_export(h_, path.c_str()) == 0; }

    /** Predict best match for each pixel in image (RTree.h:63-81); num_threads is accepted and ignored (GPU). */
    Image8 predictBest(const ImageF& depth, int /*num_threads*/, int interval = 1, Point top_left = Point(0, 0), Point bot_right = Point(-1, -1),
                       bool fill_in_gaps = true) {
        Image8 result(depth.rows, depth.cols, 255);
        if (!ensure() || avt_rtree_predict_best(h_, depth.data(), depth.rows, depth.cols, interval, top_left.x, top_left.y, bot_right.x, bot_right.y,
                                                fill_in_gaps ? 1 : 0, result.data()) != 0)
            die("predictBest");
        return result;
    }

    /** Predict distribution for all of image: numParts planes of CV_32F (RTree.h:59-61) */
    std::vector<ImageF> predict(const ImageF& depth) {
        std::vector<float> all((size_t)numParts * depth.rows * depth.cols);
        if (!ensure() || avt_rtree_predict(h_, depth.data(), depth.rows, depth.cols, all.data()) != 0) die("predict");
        std::vector<ImageF> result(numParts, ImageF(depth.rows, depth.cols));
        for (int i = 0; i < numParts; ++i) result[i].a.assign(all.begin() + (size_t)i * depth.rows * depth.cols, all.begin() + (size_t)(i + 1) * depth.rows * depth.cols);
        return result;
    }

    /** RTree.h:150-166 */
    void postProcess(Image8& image, MatrixNX<2>& com_pre, int interval = 1, int /*num_threads*/ = 1, Point top_left = Point(0, 0),
                     Point bot_right = Point(-1, -1), double dist_to_pre_weight = 0.001) {
        const bool valid = (int)com_pre.cols() == numParts;
        if (!valid) com_pre.resize(2, numParts);
        if (!ensure() || avt_rtree_post_process(h_, image.data(), image.rows, image.cols, com_pre.data(), valid ? 1 : 0, interval, top_left.x, top_left.y,
                                                bot_right.x, bot_right.y, dist_to_pre_weight) != 0)
            die("postProcess");
    }

    std::vector<RNode> nodes;
    std::vector<Distribution> leafData;
    std::vector<uint8_t> leafBestMatch;
    int numParts = 0;
    std::vector<int> partMap;
    int partMapType = 0;

private:
    // a tree filled in through the public members (nodes / leafData) is uploaded on first use
    bool ensure() {
        if (h_) return true;
        if (nodes.empty()) return false;
        std::vector<float> f(5 * nodes.size()), ld;
        std::vector<int> l(3 * nodes.size());
        for (size_t i = 0; i < nodes.size(); ++i) {
            const RNode& n = nodes[i];
            f[5 * i] = n.u[0]; f[5 * i + 1] = n.u[1]; f[5 * i + 2] = n.v[0]; f[5 * i + 3] = n.v[1]; f[5 * i + 4] = n.thresh;
            l[3 * i] = n.lnode; l[3 * i + 1] = n.rnode; l[3 * i + 2] = n.leafid;
        }
        for (const Distribution& d : leafData) ld.insert(ld.end(), d.begin(), d.end());
        avt_rtree_desc d{(int)nodes.size(), (int)leafData.size(), numParts, f.data(), l.data(), ld.data(), (int)partMap.size(), partMap.data(), partMapType};
        if (avt_rtree_create(&d, device_, &h_) != 0) return false;
        pull();
        return true;
    }
    void pull() {
        int n = 0, nl = 0, pml = 0;
        avt_rtree_info(h_, &n, &nl, &numParts, &pml, &partMapType);
        std::vector<float> f(5 * (size_t)n), ld((size_t)nl * numParts);
        std::vector<int> l(3 * (size_t)n);
        leafBestMatch.assign(nl, 0);
        partMap.assign(pml, 0);
        avt_rtree_get(h_, f.data(), l.data(), ld.data(), leafBestMatch.data(), partMap.data());
        nodes.assign(n, RNode());
        for (int i = 0; i < n; ++i) {
            RNode& nd = nodes[i];
            nd.u[0] = f[5 * i]; nd.u[1] = f[5 * i + 1]; nd.v[0] = f[5 * i + 2]; nd.v[1] = f[5 * i + 3]; nd.thresh = f[5 * i + 4];
            nd.lnode = l[3 * i]; nd.rnode = l[3 * i + 1]; nd.leafid = l[3 * i + 2];
        }
        leafData.assign(nl, Distribution());
        for (int i = 0; i < nl; ++i) leafData[i].assign(ld.begin() + (size_t)i * numParts, ld.begin() + (size_t)(i + 1) * numParts);
    }
    [[noreturn]] void die(const char* what) {   // the reference's failure mode for this class is a fatal message + exit (RTree.cpp:2984-2996)
        fprintf(stderr, "FATAL: RTree::%s: %s\n", what, avt_last_error());
        std::exit(1);
    }
    avt_rtree* h_ = nullptr;
    int device_ = 0;
};

}  // namespace ark
