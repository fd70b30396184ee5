// avt_rtree.cpp — host side of the body-part forest: file formats, part map, best-match table,
// device image of the tree, image staging for avt_rtree_predict_best, and the sequential post-processing
// (RTree::postProcess, RTree.cpp:3422-3449), which the reference also runs on the host.
#include "avt_rtree.h"

#include <exception>
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <limits>
#include <map>

#include "avt_internal.h"

namespace {

#define RT_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { avt_set_error(std::string(#x) + ": " + hipGetErrorString(e_)); return 1; } } while (0)

template <class T> bool get(std::istream& is, T& v) { is.read(reinterpret_cast<char*>(&v), sizeof(T)); return (bool)is; }
template <class T> void put(std::ostream& os, T v) { os.write(reinterpret_cast<char*>(&v), sizeof(T)); }

// leafBestMatch: index of the first strict maximum of every leaf distribution (RTree.cpp:3451-3463)
void best_match_table(avt_rtree* rt) {
    const int np = rt->num_parts, nl = np ? (int)(rt->leaf_data.size() / np) : 0;
    rt->leaf_best.assign(nl, 0);
    for (int i = 0; i < nl; ++i) {
        float best = std::numeric_limits<float>::lowest();
        for (int j = 0; j < np; ++j) {
            const float v = rt->leaf_data[(size_t)i * np + j];
            if (v > best) { best = v; rt->leaf_best[i] = (unsigned char)j; }
        }
    }
}

int validate(const avt_rtree* rt) {
    const int n = (int)(rt->links.size() / 3), nl = (int)rt->leaf_best.size();
    if (n <= 0) { avt_set_error("rtree: no nodes"); return 1; }
    if (rt->num_parts <= 0 || rt->num_parts >= 128) { avt_set_error("rtree: num_parts must be in [1, 127]"); return 1; }
    for (int i = 0; i < n; ++i) {
        const int l = rt->links[3 * i], r = rt->links[3 * i + 1], leaf = rt->links[3 * i + 2];
        if (leaf < 0) {
            // children come after their parent in every file the trainers write; requiring it rules out cycles
            if (l <= i || l >= n || r <= i || r >= n) { avt_set_error("rtree: child index out of range (children must follow their parent)"); return 1; }
        } else if (leaf >= nl) { avt_set_error("rtree: leaf id out of range"); return 1; }
    }
    return 0;
}

int upload_tree(avt_rtree* rt) {
    if (rt->device < 0) return 0;   // host-only tree: file formats and post-processing work, inference refuses
    RT_HIP(hipSetDevice(rt->device));
    const int n = (int)(rt->links.size() / 3);
    std::vector<RtNodeDev> dev(n);
    for (int i = 0; i < n; ++i) {
        const float* f = &rt->feature[5 * (size_t)i];
        const int leaf = rt->links[3 * i + 2];
        dev[i] = RtNodeDev{f[0], f[1], f[2], f[3], f[4], leaf < 0 ? rt->links[3 * i] : (int)rt->leaf_best[leaf], leaf < 0 ? rt->links[3 * i + 1] : leaf,
                           leaf < 0 ? 0 : 1};
    }
    RT_HIP(hipStreamCreateWithFlags(&rt->stream, hipStreamNonBlocking));
    RT_HIP(hipMalloc((void**)&rt->d_nodes, sizeof(RtNodeDev) * n));
    RT_HIP(hipMemcpyAsync(rt->d_nodes, dev.data(), sizeof(RtNodeDev) * n, hipMemcpyHostToDevice, rt->stream));
    RT_HIP(hipMalloc((void**)&rt->d_leaf, sizeof(float) * std::max<size_t>(1, rt->leaf_data.size())));
    RT_HIP(hipMemcpyAsync(rt->d_leaf, rt->leaf_data.data(), sizeof(float) * rt->leaf_data.size(), hipMemcpyHostToDevice, rt->stream));
    RT_HIP(hipStreamSynchronize(rt->stream));     // `dev` goes out of scope; the legacy stream is never used (a host thread may be capturing)
    return 0;
}

int reserve_images(avt_rtree* rt, size_t pixels) {
    if (rt->device < 0) { avt_set_error("rtree: created host-only (device < 0): inference needs a GPU"); return 1; }
    if (pixels <= rt->cap_pixels) return 0;
    RT_HIP(hipSetDevice(rt->device));
    if (rt->d_depth) (void)hipFree(rt->d_depth);
    if (rt->d_labels) (void)hipFree(rt->d_labels);
    rt->d_depth = nullptr; rt->d_labels = nullptr; rt->cap_pixels = 0;
    RT_HIP(hipMalloc((void**)&rt->d_depth, pixels * sizeof(float)));
    RT_HIP(hipMalloc((void**)&rt->d_labels, pixels));
    rt->cap_pixels = pixels;
    return 0;
}

bool parse_part_map(std::istream& is, std::vector<int>& result, int& type) {   // RTree::readPartMap, RTree.cpp:3465-3509
    std::string tok;
    if (!(is >> tok) || tok != "partmap") return false;
    if (!(is >> tok)) return false;
    if (tok == "disjoint") type = 1;
    else if (tok == "contiguous") type = 0;
    else return false;
    int n_src = 0, n_dst = 0;
    if (!(is >> tok) || tok != "src" || !(is >> n_src) || n_src < 0) return false;
    std::map<std::string, int> src, dst;
    for (int i = 0; i < n_src; ++i) { is >> tok; src[tok] = i; }
    if (!(is >> tok) || tok != "dest" || !(is >> n_dst)) return false;
    for (int i = 0; i < n_dst; ++i) { is >> tok; dst[tok] = i; }
    result.assign(n_src, 0);
    for (int i = 0; i < n_src && is; ++i) {
        std::string a, b;
        is >> a >> b;
        result[src[a]] = dst[b];
    }
    return true;
}

int roi_ok(int rows, int cols, int interval, int& tlx, int& tly, int& brx, int& bry) {
    if (brx == -1) { brx = cols - 1; bry = rows - 1; }
    if (rows <= 0 || cols <= 0 || interval <= 0 || tlx < 0 || tly < 0 || brx >= cols || bry >= rows || tlx > brx || tly > bry || rows >= 32768 ||
        cols >= 32768) {
        avt_set_error("rtree: bad image size, interval or region of interest");
        return 1;
    }
    return 0;
}

// ---- post-processing (host, sequential: the scan order defines which component wins) -----------------------------
const int kVisited = 128;   // VISITED_OFFSET (RTree.cpp:128)

struct Filler {
    unsigned char* img;
    int cols, interval, tlx, tly, brx, bry;
    std::vector<int> stack, comp;
    // One flood fill from grid pixel (r0, c0) over equal labels (RTree.cpp:148-183 / :247-277).  Visited pixels are
    // marked +128.  The downward probe reads row + 1 but records row + interval, exactly as the reference does (on
    // an up-scaled image with interval > 1 the component therefore leaks into the cell below).  Returns the sum of
    // the coordinates of every popped pixel in (sx, sy).
    void run(int r0, int c0, double& sx, double& sy) {
        const int row_step = interval << 16;
        const unsigned char val = img[(size_t)r0 * cols + c0];
        img[(size_t)r0 * cols + c0] = (unsigned char)(val + kVisited);
        comp.clear();
        comp.push_back((r0 << 16) + c0);
        stack.push_back(comp.back());
        sx = sy = 0.0;
        auto probe = [&](int pr, int pc, int id) {
            unsigned char& v = img[(size_t)pr * cols + pc];
            if (v == val) { v = (unsigned char)(v + kVisited); comp.push_back(id); stack.push_back(id); }
        };
        while (!stack.empty()) {
            const int id = stack.back();
            stack.pop_back();
            const int c = id & 0xffff, r = id >> 16;
            if (r >= tly + interval) probe(r - interval, c, id - row_step);
            if (r <= bry - interval) probe(r + 1, c, id + row_step);
            if (c >= tlx + interval) probe(r, c - interval, id - interval);
            if (c <= brx - interval) probe(r, c + interval, id + interval);
            sx += c; sy += r;
        }
    }
    void erase(const std::vector<int>& ids) { for (int id : ids) img[(size_t)(id >> 16) * cols + (id & 0xffff)] = 255; }
    void unmark() {
        for (int r = tly; r <= bry; ++r)
            for (int c = tlx; c <= brx; ++c) {
                unsigned char& v = img[(size_t)r * cols + c];
                if (v >= kVisited && v != 255) v = (unsigned char)(v - kVisited);
            }
    }
};

}  // namespace

extern "C" {

static int avt_rtree_create_impl(const avt_rtree_desc* d, int device, avt_rtree** out) {
    if (!d || !out || d->n_nodes <= 0 || d->n_leafs < 0 || d->n_leafs > d->n_nodes || d->num_parts <= 0 || d->num_parts > 255 || !d->feature ||
        !d->links || (d->n_leafs > 0 && !d->leaf_data)) {
        avt_set_error("avt_rtree_create: bad descriptor");
        return 1;
    }
    avt_rtree* rt = new avt_rtree();
    rt->device = device;
    rt->num_parts = d->num_parts;
    rt->feature.assign(d->feature, d->feature + 5 * (size_t)d->n_nodes);
    rt->links.assign(d->links, d->links + 3 * (size_t)d->n_nodes);
    rt->leaf_data.assign(d->leaf_data, d->leaf_data + (size_t)d->n_leafs * d->num_parts);
    if (d->part_map_len > 0 && d->part_map) rt->part_map.assign(d->part_map, d->part_map + d->part_map_len);
    rt->part_map_type = d->part_map_type;
    best_match_table(rt);
    if (validate(rt) || upload_tree(rt)) { avt_rtree_destroy(rt); return 1; }
    *out = rt;
    return 0;
}

static int avt_rtree_load_impl(const char* path, int device, avt_rtree** out) {
    if (!path || !out) { avt_set_error("avt_rtree_load: null argument"); return 1; }
    std::ifstream bin(path, std::ios::in | std::ios::binary);
    if (!bin) { avt_set_error(std::string("avt_rtree_load: cannot open ") + path); return 1; }
    avt_rtree* rt = new avt_rtree();
    rt->device = device;
    auto fail = [&](const char* why) { avt_set_error(std::string("avt_rtree_load: ") + why); delete rt; return 1; };
    char marker = 0;
    bin.get(marker);
    if (marker == 'R') {   // binary format
        uint32_t n = 0, nl = 0;
        int32_t np = 0;
        if (!get(bin, n) || !get(bin, nl) || !get(bin, np) || np <= 0 || np > 255 || n == 0 || n > (1u << 28) || nl > n) return fail("bad header");
        rt->num_parts = np;
        rt->feature.assign(5 * (size_t)n, 0.f);
        rt->links.assign(3 * (size_t)n, -1);
        rt->leaf_data.assign((size_t)nl * np, 0.f);
        uint32_t next_leaf = 0;
        for (uint32_t i = 0; i < n; ++i) {
            uint8_t is_leaf = 0;
            if (!get(bin, is_leaf)) return fail("truncated file");
            if (is_leaf) {
                uint8_t cnt = 0;
                if (next_leaf >= nl || !get(bin, cnt) || cnt > np) return fail("bad leaf record");
                for (uint8_t j = 0; j < cnt; ++j) {
                    uint8_t k = 0;
                    float v = 0.f;
                    if (!get(bin, k) || k >= np || !get(bin, v)) return fail("bad leaf entry");
                    rt->leaf_data[(size_t)next_leaf * np + k] = v;
                }
                rt->links[3 * (size_t)i + 2] = (int)next_leaf++;
            } else {
                int32_t l = 0, r = 0;
                float* f = &rt->feature[5 * (size_t)i];
                if (!get(bin, l) || !get(bin, r) || !get(bin, f[4]) || !get(bin, f[0]) || !get(bin, f[1]) || !get(bin, f[2]) || !get(bin, f[3]))
                    return fail("truncated node");
                rt->links[3 * (size_t)i] = l; rt->links[3 * (size_t)i + 1] = r;
            }
        }
        bin.get(marker);
        if (marker != 'T') return fail("end marker missing");
    } else {               // legacy text format
        bin.close();
        std::ifstream txt(path);
        size_t n = 0, nl = 0;
        int np = 0;
        if (!(txt >> n >> nl >> np) || n == 0 || np <= 0) return fail("bad text header");
        rt->num_parts = np;
        rt->feature.assign(5 * n, 0.f);
        rt->links.assign(3 * n, -1);
        rt->leaf_data.assign(nl * np, 0.f);
        for (size_t i = 0; i < n; ++i) {
            int leafid = 0;
            txt >> leafid;
            rt->links[3 * i + 2] = leafid;
            if (leafid < 0) {
                float* f = &rt->feature[5 * i];
                txt >> rt->links[3 * i] >> rt->links[3 * i + 1] >> f[4] >> f[0] >> f[1] >> f[2] >> f[3];
            }
        }
        for (size_t i = 0; i < nl * (size_t)np; ++i) txt >> rt->leaf_data[i];
        if (!txt) return fail("truncated text file");
    }
    best_match_table(rt);
    std::ifstream pm(std::string(path) + ".partmap");
    if (pm) {
        std::vector<int> m;
        int type = 0;
        if (parse_part_map(pm, m, type)) { rt->part_map = m; rt->part_map_type = type; }
    }
    if (validate(rt) || upload_tree(rt)) { avt_rtree_destroy(rt); return 1; }
    *out = rt;
    return 0;
}

static int avt_rtree_export_impl(const avt_rtree* rt, const char* path) {
    std::ofstream ofs(path, std::ios::out | std::ios::binary);
    if (!ofs) { avt_set_error(std::string("avt_rtree_export: cannot open ") + path); return 1; }
    const int n = (int)(rt->links.size() / 3), np = rt->num_parts;
    ofs.put('R');
    put<uint32_t>(ofs, (uint32_t)n);
    put<uint32_t>(ofs, (uint32_t)rt->leaf_best.size());
    put<int32_t>(ofs, np);
    for (int i = 0; i < n; ++i) {
        const int leaf = rt->links[3 * (size_t)i + 2];
        put<uint8_t>(ofs, leaf < 0 ? (uint8_t)0 : (uint8_t)255);
        if (leaf < 0) {
            const float* f = &rt->feature[5 * (size_t)i];
            put<int32_t>(ofs, rt->links[3 * (size_t)i]); put<int32_t>(ofs, rt->links[3 * (size_t)i + 1]);
            put<float>(ofs, f[4]); put<float>(ofs, f[0]); put<float>(ofs, f[1]); put<float>(ofs, f[2]); put<float>(ofs, f[3]);
        } else {
            const float* dd = &rt->leaf_data[(size_t)leaf * np];
            uint8_t cnt = 0;
            for (int j = 0; j < np; ++j) cnt += dd[j] != 0.0f;
            put<uint8_t>(ofs, cnt);
            for (int j = 0; j < np; ++j)
                if (dd[j] != 0.0f) { put<uint8_t>(ofs, (uint8_t)j); put<float>(ofs, dd[j]); }
        }
    }
    ofs.put('T');
    if (!ofs) { avt_set_error("avt_rtree_export: write failed"); return 1; }
    return 0;
}

void avt_rtree_destroy(avt_rtree* rt) {
    if (!rt) return;
    if (rt->d_nodes) (void)hipFree(rt->d_nodes);
    if (rt->d_leaf) (void)hipFree(rt->d_leaf);
    if (rt->d_depth) (void)hipFree(rt->d_depth);
    if (rt->d_labels) (void)hipFree(rt->d_labels);
    if (rt->stream) (void)hipStreamDestroy(rt->stream);
    delete rt;
}

int avt_rtree_info(const avt_rtree* rt, int* n_nodes, int* n_leafs, int* num_parts, int* part_map_len, int* part_map_type) {
    if (!rt) { avt_set_error("avt_rtree_info: null tree"); return 1; }
    if (n_nodes) *n_nodes = (int)(rt->links.size() / 3);
    if (n_leafs) *n_leafs = (int)rt->leaf_best.size();
    if (num_parts) *num_parts = rt->num_parts;
    if (part_map_len) *part_map_len = (int)rt->part_map.size();
    if (part_map_type) *part_map_type = rt->part_map_type;
    return 0;
}

int avt_rtree_get(const avt_rtree* rt, float* feature, int* links, float* leaf_data, unsigned char* leaf_best, int* part_map) {
    if (!rt) { avt_set_error("avt_rtree_get: null tree"); return 1; }
    if (feature) std::copy(rt->feature.begin(), rt->feature.end(), feature);
    if (links) std::copy(rt->links.begin(), rt->links.end(), links);
    if (leaf_data) std::copy(rt->leaf_data.begin(), rt->leaf_data.end(), leaf_data);
    if (leaf_best) std::copy(rt->leaf_best.begin(), rt->leaf_best.end(), leaf_best);
    if (part_map) std::copy(rt->part_map.begin(), rt->part_map.end(), part_map);
    return 0;
}

static int avt_rtree_images_upload_impl(avt_rtree* rt, int n_images, int rows, int cols, const float* depth) {
    if (!rt || !depth || n_images <= 0 || rows <= 0 || cols <= 0) { avt_set_error("avt_rtree_images_upload: bad arguments"); return 1; }
    const size_t pixels = (size_t)n_images * rows * cols;
    if (reserve_images(rt, pixels)) return 1;
    RT_HIP(hipSetDevice(rt->device));
    RT_HIP(hipMemcpyAsync(rt->d_depth, depth, pixels * sizeof(float), hipMemcpyHostToDevice, rt->stream));
    rt->n_images = n_images; rt->rows = rows; rt->cols = cols;
    return 0;
}

int avt_rtree_predict_best_resident(avt_rtree* rt, int interval, int tlx, int tly, int brx, int bry, int fill) {
    if (!rt || rt->n_images <= 0) { avt_set_error("avt_rtree_predict_best_resident: no images resident"); return 1; }
    if (roi_ok(rt->rows, rt->cols, interval, tlx, tly, brx, bry)) return 1;
    RT_HIP(hipSetDevice(rt->device));
    if (avt_rtree_launch_predict(rt, rt->n_images, rt->rows, rt->cols, interval, tlx, tly, brx, bry, fill)) { avt_set_error("rtree: kernel launch failed"); return 1; }
    return 0;
}

int avt_rtree_labels_download(avt_rtree* rt, int image, unsigned char* out) {
    if (!rt || !out || image < 0 || image >= rt->n_images) { avt_set_error("avt_rtree_labels_download: bad arguments"); return 1; }
    const size_t px = (size_t)rt->rows * rt->cols;
    RT_HIP(hipMemcpyAsync(out, rt->d_labels + px * image, px, hipMemcpyDeviceToHost, rt->stream));
    RT_HIP(hipStreamSynchronize(rt->stream));
    return 0;
}

int avt_rtree_sync(avt_rtree* rt) {
    if (!rt) { avt_set_error("avt_rtree_sync: null tree"); return 1; }
    RT_HIP(hipStreamSynchronize(rt->stream));
    return 0;
}

static int avt_rtree_predict_best_impl(avt_rtree* rt, const float* depth, int rows, int cols, int interval, int tlx, int tly, int brx, int bry, int fill,
                           unsigned char* labels_out) {
    if (!rt || !depth || !labels_out) { avt_set_error("avt_rtree_predict_best: null argument"); return 1; }
    if (roi_ok(rows, cols, interval, tlx, tly, brx, bry)) return 1;
    if (avt_rtree_images_upload(rt, 1, rows, cols, depth)) return 1;
    if (avt_rtree_launch_predict(rt, 1, rows, cols, interval, tlx, tly, brx, bry, fill)) { avt_set_error("rtree: kernel launch failed"); return 1; }
    return avt_rtree_labels_download(rt, 0, labels_out);
}

static int avt_rtree_predict_impl(avt_rtree* rt, const float* depth, int rows, int cols, float* dist_out) {
    if (!rt || !depth || !dist_out || rows <= 0 || cols <= 0) { avt_set_error("avt_rtree_predict: bad arguments"); return 1; }
    if (avt_rtree_images_upload(rt, 1, rows, cols, depth)) return 1;
    const size_t n = (size_t)rt->num_parts * rows * cols;
    float* d_out = nullptr;
    RT_HIP(hipMalloc((void**)&d_out, n * sizeof(float)));
    int rc = avt_rtree_launch_predict_dist(rt, rows, cols, d_out);
    if (rc) avt_set_error("rtree: kernel launch failed");
    if (!rc && hipMemcpyAsync(dist_out, d_out, n * sizeof(float), hipMemcpyDeviceToHost, rt->stream) != hipSuccess) { avt_set_error("rtree: download failed"); rc = 1; }
    if (hipStreamSynchronize(rt->stream) != hipSuccess && !rc) { avt_set_error("rtree: stream failed"); rc = 1; }
    (void)hipFree(d_out);
    return rc;
}

static int avt_rtree_post_process_impl(const avt_rtree* rt, unsigned char* image, int rows, int cols, double* com_pre, int com_pre_valid, int interval, int tlx,
                           int tly, int brx, int bry, double dist_to_pre_weight) {
    if (!rt || !image || !com_pre) { avt_set_error("avt_rtree_post_process: null argument"); return 1; }
    if (roi_ok(rows, cols, interval, tlx, tly, brx, bry)) return 1;
    const int np = rt->num_parts;
    if (!com_pre_valid)                               // the resize branch of RTree.cpp:3431-3435
        for (int i = 0; i < np; ++i) { com_pre[2 * i] = -1.; com_pre[2 * i + 1] = 0.; }
    Filler fl{image, cols, interval, tlx, tly, brx, bry, {}, {}};
    if (rt->part_map_type == 0) {
        // 'contiguous' part map: per part keep the component with the best score = size - weight * squared distance of
        // its centre of mass to the previous frame's (suppressPartNonMax, RTree.cpp:125-237)
        std::vector<std::vector<int>> kept(np);
        std::vector<double> kept_score(np, 0.0), kept_com(2 * (size_t)np, 0.0);
        for (int r = tly; r <= bry; r += interval)
            for (int c = tlx; c <= brx; c += interval) {
                const unsigned char val = image[(size_t)r * cols + c];
                if (val >= kVisited) continue;
                if (val >= np) { avt_set_error("avt_rtree_post_process: label out of range"); return 1; }
                double sx, sy;
                fl.run(r, c, sx, sy);
                const double n = (double)fl.comp.size(), cx = sx / n, cy = sy / n;
                double score = n;
                if (com_pre[2 * val] >= 0.) {
                    const double dx = cx - com_pre[2 * val], dy = cy - com_pre[2 * val + 1];
                    score -= (dx * dx + dy * dy) * dist_to_pre_weight;
                }
                if (score > kept_score[val]) {
                    kept_score[val] = score;
                    kept_com[2 * val] = cx; kept_com[2 * val + 1] = cy;
                    fl.erase(kept[val]);
                    kept[val].swap(fl.comp);
                } else {
                    fl.erase(fl.comp);
                }
            }
        for (int i = 0; i < np; ++i) {
            if (kept[i].empty()) com_pre[2 * i] = -1.;
            else { com_pre[2 * i] = kept_com[2 * i]; com_pre[2 * i + 1] = kept_com[2 * i + 1]; }
        }
    } else {
        // 'disjoint' part map: only drop pieces smaller than 0.05 % of the grid (removeSmallPieces, RTree.cpp:239-323)
        const size_t min_size = (size_t)(rows * cols / (interval * interval) * 0.0005);
        for (int r = tly; r <= bry; r += interval)
            for (int c = tlx; c <= brx; c += interval) {
                if (image[(size_t)r * cols + c] >= kVisited) continue;
                double sx, sy;
                fl.run(r, c, sx, sy);
                if (fl.comp.size() < min_size) fl.erase(fl.comp);
            }
    }
    fl.unmark();
    if (interval > 1)                                  // upscaleGrid (RTree.cpp:70-99), fill clamped to the image width
        for (int rr = tly + interval; rr <= bry; rr += interval) {
            const unsigned char* ref = image + (size_t)rr * cols;
            for (int r = rr; r < rr + interval && r <= bry; ++r) {
                unsigned char* row = image + (size_t)r * cols;
                for (int c = tlx; c <= brx; c += interval) std::memset(row + c, ref[c], (size_t)std::min(interval, cols - c));
            }
        }
    return 0;
}

}  // extern "C"

// ---- exported entry points of the functions above: no C++ exception crosses the C ABI
extern "C" {
int avt_rtree_create(const avt_rtree_desc* d, int device, avt_rtree** out) {
    try { return avt_rtree_create_impl(d, device, out); }
    catch (const std::exception& e) { avt_set_error(std::string("avt_rtree_create: ") + e.what()); return 1; }
    catch (...) { avt_set_error("avt_rtree_create: unknown exception"); return 1; }
}

int avt_rtree_load(const char* path, int device, avt_rtree** out) {
    try { return avt_rtree_load_impl(path, device, out); }
    catch (const std::exception& e) { avt_set_error(std::string("avt_rtree_load: ") + e.what()); return 1; }
    catch (...) { avt_set_error("avt_rtree_load: unknown exception"); return 1; }
}

int avt_rtree_export(const avt_rtree* rt, const char* path) {
    try { return avt_rtree_export_impl(rt, path); }
    catch (const std::exception& e) { avt_set_error(std::string("avt_rtree_export: ") + e.what()); return 1; }
    catch (...) { avt_set_error("avt_rtree_export: unknown exception"); return 1; }
}

int avt_rtree_images_upload(avt_rtree* rt, int n_images, int rows, int cols, const float* depth) {
    try { return avt_rtree_images_upload_impl(rt, n_images, rows, cols, depth); }
    catch (const std::exception& e) { avt_set_error(std::string("avt_rtree_images_upload: ") + e.what()); return 1; }
    catch (...) { avt_set_error("avt_rtree_images_upload: unknown exception"); return 1; }
}

int avt_rtree_predict_best(avt_rtree* rt, const float* depth, int rows, int cols, int interval, int tlx, int tly, int brx, int bry, int fill,
                           unsigned char* labels_out) {
    try { return avt_rtree_predict_best_impl(rt, depth, rows, cols, interval, tlx, tly, brx, bry, fill, labels_out); }
    catch (const std::exception& e) { avt_set_error(std::string("avt_rtree_predict_best: ") + e.what()); return 1; }
    catch (...) { avt_set_error("avt_rtree_predict_best: unknown exception"); return 1; }
}

int avt_rtree_predict(avt_rtree* rt, const float* depth, int rows, int cols, float* dist_out) {
    try { return avt_rtree_predict_impl(rt, depth, rows, cols, dist_out); }
    catch (const std::exception& e) { avt_set_error(std::string("avt_rtree_predict: ") + e.what()); return 1; }
    catch (...) { avt_set_error("avt_rtree_predict: unknown exception"); return 1; }
}

int avt_rtree_post_process(const avt_rtree* rt, unsigned char* image, int rows, int cols, double* com_pre, int com_pre_valid, int interval, int tlx,
                           int tly, int brx, int bry, double dist_to_pre_weight) {
    try { return avt_rtree_post_process_impl(rt, image, rows, cols, com_pre, com_pre_valid, interval, tlx, tly, brx, bry, dist_to_pre_weight); }
    catch (const std::exception& e) { avt_set_error(std::string("avt_rtree_post_process: ") + e.what()); return 1; }
    catch (...) { avt_set_error("avt_rtree_post_process: unknown exception"); return 1; }
}
}  // extern "C"
