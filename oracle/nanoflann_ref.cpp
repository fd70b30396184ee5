// nanoflann_ref.cpp — builds the reference's own KD-tree nearest-neighbour search as a shared library.
//
// TEST INFRASTRUCTURE ONLY (see avatar_oracle.cpp header).  This file contains no reference code: it
// #includes the reference's vendored header where it lies (/root/reference/include/nanoflann.hpp, v0x130)
// and drives it exactly the way findNN(..., invert=true) does (AvatarOptimizer.cpp:841-907): per part,
// compact the visible model points in ascending index order, build a KD tree with leaf size 10, then 1-NN
// for every data point in index order with SearchParams(10).  Output goes to oracle/_ref/ (git-ignored).
// It exists only in the build container (the GPU box has no /root/reference); its outputs are committed as
// golden vectors under tests/golden/ by tests/golden/make_nn_golden.py.
#include <nanoflann.hpp>

#include <cstddef>
#include <thread>
#include <vector>

namespace {
struct PartCloud {
    std::vector<double> xyz;  // column-major 3 x n
    inline size_t kdtree_get_point_count() const { return xyz.size() / 3; }
    inline double kdtree_get_pt(const size_t idx, const size_t dim) const { return xyz[3 * idx + dim]; }
    template <class BBOX>
    bool kdtree_get_bbox(BBOX&) const { return false; }
};
typedef nanoflann::KDTreeSingleIndexAdaptor<nanoflann::L2_Simple_Adaptor<double, PartCloud>, PartCloud, 3, int> Tree;
}  // namespace

extern "C" int ref_find_nn_inverted_mt(int V, int num_parts, const int* model_part /*V: part of each model point*/,
                                       const double* model_cloud, const unsigned char* visible, const double* data,
                                       const int* labels, int N, int* model_idx_out, double* dist_out, int nthreads) {
    std::vector<PartCloud> clouds(num_parts);
    std::vector<std::vector<int>> newIdx(num_parts);
    std::vector<Tree*> trees(num_parts, nullptr);
    for (int k = 0; k < V; ++k) {
        if (!visible[k]) continue;
        const int q = model_part[k];
        clouds[q].xyz.push_back(model_cloud[3 * k]);
        clouds[q].xyz.push_back(model_cloud[3 * k + 1]);
        clouds[q].xyz.push_back(model_cloud[3 * k + 2]);
        newIdx[q].push_back(k);
    }
    for (int q = 0; q < num_parts; ++q) {
        if (newIdx[q].empty()) continue;
        trees[q] = new Tree(3, clouds[q], nanoflann::KDTreeSingleIndexAdaptorParams(10));
        trees[q]->buildIndex();
    }
    auto query = [&](int lo, int hi) {
    for (int i = lo; i < hi; ++i) {
        const int q = labels[i];
        if (trees[q] == nullptr) {
            model_idx_out[i] = -1;
            if (dist_out) dist_out[i] = -1.0;
            continue;
        }
        int index = -1;
        double dist = 0.0;
        nanoflann::KNNResultSet<double, int> rs(1);
        rs.init(&index, &dist);
        trees[q]->findNeighbors(rs, data + 3 * (size_t)i, nanoflann::SearchParams(10));
        model_idx_out[i] = newIdx[q][index];
        if (dist_out) dist_out[i] = dist;
    }
    };
    // the reference queries serially (AvatarOptimizer.cpp:896-904); nthreads > 1 is the bench's "fastest CPU" mode: the
    // built trees are read-only, every query writes its own output slot
    if (nthreads <= 1) query(0, N);
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nthreads; ++t) pool.emplace_back(query, (int)((long long)N * t / nthreads), (int)((long long)N * (t + 1) / nthreads));
        for (auto& th : pool) th.join();
    }
    for (auto* t : trees) delete t;
    return 0;
}

extern "C" int ref_find_nn_inverted(int V, int num_parts, const int* model_part, const double* model_cloud, const unsigned char* visible,
                                    const double* data, const int* labels, int N, int* model_idx_out, double* dist_out) {
    return ref_find_nn_inverted_mt(V, num_parts, model_part, model_cloud, visible, data, labels, N, model_idx_out, dist_out, 1);
}
