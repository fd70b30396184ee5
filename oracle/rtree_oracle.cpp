// rtree_oracle.cpp — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of the reference's body-part forest inference, the stage right before the fitting path:
//   file formats        RTree::loadFile / exportFile            RTree.cpp:2967-3120
//   part map file       RTree::readPartMap                      RTree.cpp:3465-3509
//   best-match table    RTree::updateBestMatchTable             RTree.cpp:3451-3463
//   per-pixel feature   scoreByFeature / getDepth               RTree.cpp:39-68
//   image inference     RTree::predictBest(depth, ...)          RTree.cpp:3184-3262 (+ upscaleGrid :70-99)
//                       RTree::predict(depth) distributions     RTree.cpp:3122-3132, :3156-3182
//   post-processing     RTree::postProcess                      RTree.cpp:3422-3449
//                       suppressPartNonMax / removeSmallPieces  RTree.cpp:125-323
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
// (avatar_amd/csrc/avt_rtree.*) never does.  The reference has no tests or golden vectors for this stage and cannot be
// built here (OpenCV/Eigen/Boost are absent): parity unpinned against reference outputs; the restatement is pinned by
// hand-computed known answers (tests/test_rtree_cpu.py) and follows the cited lines literally, quirks included:
//   * predictBest and upscaleGrid pre-increment their row counter, so the first row they touch is top_left.y + interval;
//   * probes outside the region of interest (not the image) read BACKGROUND_DEPTH;
//   * the flood fills look "down" at row+1 but enqueue row+interval (RTree.cpp:175, :270), so on an up-scaled label
//     image (interval > 1) a component leaks into the cell below whatever its label.
// One deliberate difference: upscaleGrid's memset may run past the row end in the reference (cc + interval > cols);
// here the fill is clamped to the image width.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <limits>
#include <map>
#include <sstream>
#include <string>
#include <vector>

namespace {

const float BACKGROUND_DEPTH = 20.f;   // RTree.cpp:325

struct Node {
    float ux, uy, vx, vy, thresh;
    int lnode, rnode, leafid;            // leafid == -1: internal node (RTree.h:28-41)
};

struct Tree {
    std::vector<Node> nodes;
    std::vector<std::vector<float>> leaf;   // leaf distributions [nLeafs][numParts]
    std::vector<uint8_t> best;              // leafBestMatch
    int num_parts = 0;
    std::vector<int> part_map;
    int part_map_type = 0;                  // 0 contiguous, 1 disjoint
};

void update_best(Tree& t) {   // RTree.cpp:3451-3463: first strict maximum
    t.best.assign(t.leaf.size(), 0);
    for (size_t i = 0; i < t.leaf.size(); ++i) {
        float best = std::numeric_limits<float>::lowest();
        for (int j = 0; j < t.num_parts; ++j)
            if (t.leaf[i][j] > best) { best = t.leaf[i][j]; t.best[i] = (uint8_t)j; }
    }
}

template <class T> bool rd(std::istream& is, T& v) { is.read(reinterpret_cast<char*>(&v), sizeof(T)); return (bool)is; }
template <class T> void wr(std::ostream& os, T v) { os.write(reinterpret_cast<char*>(&v), sizeof(T)); }

bool read_part_map(std::istream& is, std::vector<int>& result, int& type) {   // RTree.cpp:3465-3509
    std::string marker;
    is >> marker;
    if (marker != "partmap") return false;
    is >> marker;
    if (marker == "disjoint") type = 1;
    else if (marker == "contiguous") type = 0;
    else return false;
    int n_old = 0, n_new = 0;
    is >> marker;
    if (marker != "src") return false;
    is >> n_old;
    std::map<std::string, int> old_enum, new_enum;
    for (int i = 0; i < n_old; ++i) { std::string name; is >> name; old_enum[name] = i; }
    is >> marker;
    if (marker != "dest") return false;
    is >> n_new;
    for (int i = 0; i < n_new; ++i) { std::string name; is >> name; new_enum[name] = i; }
    result.assign(n_old, 0);
    for (int i = 0; i < n_old; ++i) {
        if (!is) break;
        std::string a, b;
        is >> a >> b;
        result[old_enum[a]] = new_enum[b];
    }
    return true;
}

bool load(Tree& t, const std::string& path) {   // RTree.cpp:2967-3064
    std::ifstream bifs(path, std::ios::in | std::ios::binary);
    if (!bifs) return false;
    char marker = 0;
    bifs.get(marker);
    if (marker == 'R') {
        uint32_t n_nodes = 0, n_leafs = 0;
        int32_t np = 0;
        rd(bifs, n_nodes); rd(bifs, n_leafs); rd(bifs, np);
        t.num_parts = np;
        t.nodes.assign(n_nodes, Node{0, 0, 0, 0, 0, -1, -1, -1});
        t.leaf.assign(n_leafs, std::vector<float>());
        uint32_t last = 0;
        for (uint32_t i = 0; i < n_nodes; ++i) {
            uint8_t is_leaf = 0;
            rd(bifs, is_leaf);
            if (is_leaf) {
                if (last >= n_leafs) return false;
                t.leaf[last].assign(np, 0.f);
                uint8_t cnt = 0;
                rd(bifs, cnt);
                if (cnt > np) return false;
                for (uint8_t j = 0; j < cnt; ++j) {
                    uint8_t k = 0;
                    float v = 0;
                    rd(bifs, k);
                    if (k >= np) return false;
                    rd(bifs, v);
                    t.leaf[last][k] = v;
                }
                t.nodes[i].leafid = (int)last++;
            } else {
                int32_t l = 0, r = 0;
                rd(bifs, l); rd(bifs, r);
                t.nodes[i].lnode = l; t.nodes[i].rnode = r;
                rd(bifs, t.nodes[i].thresh);
                rd(bifs, t.nodes[i].ux); rd(bifs, t.nodes[i].uy);
                rd(bifs, t.nodes[i].vx); rd(bifs, t.nodes[i].vy);
            }
        }
        bifs.get(marker);
        if (marker != 'T') return false;
    } else {   // legacy text format
        bifs.close();
        std::ifstream ifs(path);
        if (!ifs) return false;
        size_t n_nodes = 0, n_leafs = 0;
        ifs >> n_nodes >> n_leafs >> t.num_parts;
        if (!ifs) return false;
        t.nodes.assign(n_nodes, Node{0, 0, 0, 0, 0, -1, -1, -1});
        t.leaf.assign(n_leafs, std::vector<float>());
        for (size_t i = 0; i < n_nodes; ++i) {
            ifs >> t.nodes[i].leafid;
            if (t.nodes[i].leafid < 0)
                ifs >> t.nodes[i].lnode >> t.nodes[i].rnode >> t.nodes[i].thresh >> t.nodes[i].ux >> t.nodes[i].uy >> t.nodes[i].vx >> t.nodes[i].vy;
        }
        for (size_t i = 0; i < n_leafs; ++i) {
            t.leaf[i].assign(t.num_parts, 0.f);
            for (int j = 0; j < t.num_parts; ++j) ifs >> t.leaf[i][j];
        }
        if (!ifs) return false;
    }
    update_best(t);
    std::ifstream pm(path + ".partmap");
    if (pm) {
        std::vector<int> m;
        int type = 0;
        if (read_part_map(pm, m, type)) { t.part_map = m; t.part_map_type = type; }
    }
    return true;
}

bool export_file(const Tree& t, const std::string& path) {   // RTree.cpp:3066-3120
    std::ofstream ofs(path, std::ios::out | std::ios::binary);
    if (!ofs) return false;
    ofs.put('R');
    wr<uint32_t>(ofs, (uint32_t)t.nodes.size());
    wr<uint32_t>(ofs, (uint32_t)t.leaf.size());
    wr<int32_t>(ofs, t.num_parts);
    for (const Node& n : t.nodes) {
        wr<uint8_t>(ofs, n.leafid < 0 ? (uint8_t)0 : (uint8_t)255);
        if (n.leafid < 0) {
            wr<int32_t>(ofs, n.lnode); wr<int32_t>(ofs, n.rnode); wr<float>(ofs, n.thresh);
            wr<float>(ofs, n.ux); wr<float>(ofs, n.uy); wr<float>(ofs, n.vx); wr<float>(ofs, n.vy);
        } else {
            const std::vector<float>& d = t.leaf[n.leafid];
            uint8_t cnt = 0;
            for (int j = 0; j < t.num_parts; ++j) if (d[j] != 0.0f) ++cnt;
            wr<uint8_t>(ofs, cnt);
            for (int j = 0; j < t.num_parts; ++j)
                if (d[j] != 0.0f) { wr<uint8_t>(ofs, (uint8_t)j); wr<float>(ofs, d[j]); }
        }
    }
    ofs.put('T');
    return (bool)ofs;
}

// RTree.cpp:3184-3262 with upscaleGrid (:70-99)
void predict_best(const Tree& t, const float* depth, int rows, int cols, int interval, int tlx, int tly, int brx, int bry,
                  int fill_in_gaps, uint8_t* out) {
    std::memset(out, 255, (size_t)rows * cols);
    if (brx == -1) { brx = cols - 1; bry = rows - 1; }
    for (int r = tly + interval; r <= bry; r += interval) {          // r = (row += interval): the first row is skipped
        const float* in = depth + (size_t)r * cols;
        uint8_t* ptr = out + (size_t)r * cols;
        for (int c = tlx; c <= brx; c += interval) {
            if (in[c] == 0.f) continue;
            int nodeid = 0;
            const float sample = in[c];
            while (t.nodes[nodeid].leafid == -1) {
                const Node& nd = t.nodes[nodeid];
                const float utx = nd.ux / sample, uty = nd.uy / sample, vtx = nd.vx / sample, vty = nd.vy / sample;
                const int ux = (int32_t)std::round(utx) + c, uy = (int32_t)std::round(uty) + r;
                const int vx = (int32_t)std::round(vtx) + c, vy = (int32_t)std::round(vty) + r;
                float zu, zv;
                if (ux < tlx || uy < tly || ux > brx || uy > bry) zu = BACKGROUND_DEPTH;
                else { zu = depth[(size_t)uy * cols + ux]; if (zu == 0.0f) zu = BACKGROUND_DEPTH; }
                if (vx < tlx || vy < tly || vx > brx || vy > bry) zv = BACKGROUND_DEPTH;
                else { zv = depth[(size_t)vy * cols + vx]; if (zv == 0.0f) zv = BACKGROUND_DEPTH; }
                nodeid = (zu - zv < nd.thresh) ? nd.lnode : nd.rnode;
            }
            ptr[c] = t.best[t.nodes[nodeid].leafid];
        }
    }
    if (fill_in_gaps && interval > 1) {
        for (int rr = tly + interval; rr <= bry; rr += interval) {
            const uint8_t* ref = out + (size_t)rr * cols;
            for (int r = rr; r < rr + interval && r <= bry; ++r) {
                uint8_t* ptr = out + (size_t)r * cols;
                for (int cc = tlx; cc <= brx; cc += interval) std::memset(ptr + cc, ref[cc], (size_t)std::min(interval, cols - cc));
            }
        }
    }
}

// RTree.cpp:3156-3182 with predictRecursive (:3122-3132) and scoreByFeature / getDepth (:39-68): every pixel with
// depth > 0, probes bounded by the IMAGE, out[part][r][c] = leaf distribution (0 elsewhere)
void predict_dist(const Tree& t, const float* depth, int rows, int cols, float* out) {
    std::memset(out, 0, sizeof(float) * (size_t)t.num_parts * rows * cols);
    auto get = [&](int x, int y) {
        if (y < 0 || x < 0 || y >= rows || x >= cols) return BACKGROUND_DEPTH;
        const float z = depth[(size_t)y * cols + x];
        return z == 0.0f ? BACKGROUND_DEPTH : z;
    };
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) {
            const float sample = depth[(size_t)r * cols + c];
            if (sample <= 0.f) continue;
            int nodeid = 0;
            while (t.nodes[nodeid].leafid == -1) {
                const Node& nd = t.nodes[nodeid];
                const float utx = nd.ux / sample, uty = nd.uy / sample, vtx = nd.vx / sample, vty = nd.vy / sample;
                const int ux = (int32_t)std::round(utx) + c, uy = (int32_t)std::round(uty) + r;
                const int vx = (int32_t)std::round(vtx) + c, vy = (int32_t)std::round(vty) + r;
                nodeid = (get(ux, uy) - get(vx, vy) < nd.thresh) ? nd.lnode : nd.rnode;
            }
            const std::vector<float>& d = t.leaf[t.nodes[nodeid].leafid];
            for (int i = 0; i < t.num_parts; ++i) out[((size_t)i * rows + r) * cols + c] = d[i];
        }
}

void upscale(uint8_t* image, int rows, int cols, int interval, int tlx, int tly, int brx, int bry) {
    for (int rr = tly + interval; rr <= bry; rr += interval) {
        const uint8_t* ref = image + (size_t)rr * cols;
        for (int r = rr; r < rr + interval && r <= bry; ++r) {
            uint8_t* ptr = image + (size_t)r * cols;
            for (int cc = tlx; cc <= brx; cc += interval) std::memset(ptr + cc, ref[cc], (size_t)std::min(interval, cols - cc));
        }
    }
}

const int VISITED = 128;

// flood fill shared by both post-processing variants (RTree.cpp:148-183, :247-277): fills comp with the visited ids
// (row << 16 | col) and, if com != nullptr, adds every popped pixel to the centre-of-mass sum
void flood(uint8_t* image, int cols, int interval, int tlx, int tly, int brx, int bry, int rr, int cc, std::vector<int>& stk,
           std::vector<int>& comp, double* com) {
    const int hi_bit = (1 << 16) * interval, lo_mask = (1 << 16) - 1;
    const uint8_t val = image[(size_t)rr * cols + cc];
    image[(size_t)rr * cols + cc] += VISITED;
    stk.push_back((rr << 16) + cc);
    comp.clear();
    comp.push_back(stk.back());
    auto maybe_visit = [&](int nr, int nc, int nid) {
        uint8_t& v = image[(size_t)nr * cols + nc];
        if (v == val) { v += VISITED; comp.push_back(nid); stk.push_back(nid); }
    };
    while (!stk.empty()) {
        const int id = stk.back();
        const int cur_c = id & lo_mask, cur_r = id >> 16;
        stk.pop_back();
        if (cur_r >= tly + interval) maybe_visit(cur_r - interval, cur_c, id - hi_bit);
        if (cur_r <= bry - interval) maybe_visit(cur_r + 1, cur_c, id + hi_bit);          // sic: row + 1, id of row + interval
        if (cur_c >= tlx + interval) maybe_visit(cur_r, cur_c - interval, id - interval);
        if (cur_c <= brx - interval) maybe_visit(cur_r, cur_c + interval, id + interval);
        if (com) { com[0] += cur_c; com[1] += cur_r; }
    }
}

void unmark(uint8_t* image, int cols, int tlx, int tly, int brx, int bry) {   // RTree.cpp:214-236
    for (int r = tly; r <= bry; ++r)
        for (int c = tlx; c <= brx; ++c) {
            uint8_t& v = image[(size_t)r * cols + c];
            if (v >= VISITED && v != 255) v -= VISITED;
        }
}

void suppress_non_max(uint8_t* image, int cols, int interval, int num_parts, int tlx, int tly, int brx, int bry, double* com_pre, double w) {
    std::vector<int> stk, cur;
    std::vector<std::vector<int>> best_comp(num_parts);
    std::vector<double> best_score(num_parts, 0.0), com_best(2 * (size_t)num_parts, 0.0);
    const int lo_mask = (1 << 16) - 1;
    for (int rr = tly; rr <= bry; rr += interval)
        for (int cc = tlx; cc <= brx; cc += interval) {
            const uint8_t val = image[(size_t)rr * cols + cc];
            if (val >= VISITED) continue;
            double com[2] = {0.0, 0.0};
            const bool has_prev = com_pre[2 * val] >= 0.;
            flood(image, cols, interval, tlx, tly, brx, bry, rr, cc, stk, cur, com);
            double score = (double)cur.size();
            com[0] /= (double)cur.size(); com[1] /= (double)cur.size();
            if (has_prev) {
                const double dx = com[0] - com_pre[2 * val], dy = com[1] - com_pre[2 * val + 1];
                score -= (dx * dx + dy * dy) * w;
            }
            if (score > best_score[val]) {
                best_score[val] = score;
                com_best[2 * val] = com[0]; com_best[2 * val + 1] = com[1];
                for (int id : best_comp[val]) image[(size_t)(id >> 16) * cols + (id & lo_mask)] = 255;
                best_comp[val].swap(cur);
            } else {
                for (int id : cur) image[(size_t)(id >> 16) * cols + (id & lo_mask)] = 255;
            }
        }
    for (int i = 0; i < num_parts; ++i) {
        if (best_comp[i].empty()) com_pre[2 * i] = -1.;
        else { com_pre[2 * i] = com_best[2 * i]; com_pre[2 * i + 1] = com_best[2 * i + 1]; }
    }
    unmark(image, cols, tlx, tly, brx, bry);
}

void remove_small(uint8_t* image, int rows, int cols, int interval, int tlx, int tly, int brx, int bry, double thresh = 0.0005) {
    std::vector<int> stk, cur;
    const size_t scaled = (size_t)(rows * cols / (interval * interval) * thresh);
    const int lo_mask = (1 << 16) - 1;
    for (int rr = tly; rr <= bry; rr += interval)
        for (int cc = tlx; cc <= brx; cc += interval) {
            if (image[(size_t)rr * cols + cc] >= VISITED) continue;
            flood(image, cols, interval, tlx, tly, brx, bry, rr, cc, stk, cur, nullptr);
            if (cur.size() < scaled)
                for (int id : cur) image[(size_t)(id >> 16) * cols + (id & lo_mask)] = 255;
        }
    unmark(image, cols, tlx, tly, brx, bry);
}

}  // namespace

extern "C" {

void* orc_rtree_create(int n_nodes, const float* uvt, const int* lrl, int n_leafs, const float* leaf_data, int num_parts) {
    Tree* t = new Tree();
    t->num_parts = num_parts;
    t->nodes.resize(n_nodes);
    for (int i = 0; i < n_nodes; ++i)
        t->nodes[i] = Node{uvt[5 * i], uvt[5 * i + 1], uvt[5 * i + 2], uvt[5 * i + 3], uvt[5 * i + 4], lrl[3 * i], lrl[3 * i + 1], lrl[3 * i + 2]};
    t->leaf.resize(n_leafs);
    for (int i = 0; i < n_leafs; ++i) t->leaf[i].assign(leaf_data + (size_t)i * num_parts, leaf_data + (size_t)(i + 1) * num_parts);
    update_best(*t);
    return t;
}

void* orc_rtree_load(const char* path) {
    Tree* t = new Tree();
    if (!load(*t, path)) { delete t; return nullptr; }
    return t;
}

int orc_rtree_export(const void* h, const char* path) { return export_file(*(const Tree*)h, path) ? 0 : 1; }

void orc_rtree_destroy(void* h) { delete (Tree*)h; }

void orc_rtree_dims(const void* h, int* n_nodes, int* n_leafs, int* num_parts, int* part_map_len, int* part_map_type) {
    const Tree* t = (const Tree*)h;
    *n_nodes = (int)t->nodes.size(); *n_leafs = (int)t->leaf.size(); *num_parts = t->num_parts;
    *part_map_len = (int)t->part_map.size(); *part_map_type = t->part_map_type;
}

void orc_rtree_get(const void* h, float* uvt, int* lrl, float* leaf_data, unsigned char* best, int* part_map) {
    const Tree* t = (const Tree*)h;
    for (size_t i = 0; i < t->nodes.size(); ++i) {
        const Node& n = t->nodes[i];
        uvt[5 * i] = n.ux; uvt[5 * i + 1] = n.uy; uvt[5 * i + 2] = n.vx; uvt[5 * i + 3] = n.vy; uvt[5 * i + 4] = n.thresh;
        lrl[3 * i] = n.lnode; lrl[3 * i + 1] = n.rnode; lrl[3 * i + 2] = n.leafid;
    }
    for (size_t i = 0; i < t->leaf.size(); ++i) {
        std::copy(t->leaf[i].begin(), t->leaf[i].end(), leaf_data + i * t->num_parts);
        best[i] = t->best[i];
    }
    std::copy(t->part_map.begin(), t->part_map.end(), part_map);
}

void orc_rtree_predict_best(const void* h, const float* depth, int rows, int cols, int interval, int tlx, int tly, int brx, int bry,
                            int fill_in_gaps, unsigned char* out) {
    predict_best(*(const Tree*)h, depth, rows, cols, interval, tlx, tly, brx, bry, fill_in_gaps, out);
}

void orc_rtree_predict(const void* h, const float* depth, int rows, int cols, float* out) { predict_dist(*(const Tree*)h, depth, rows, cols, out); }

// com_pre: 2 x num_parts column-major; com_pre_valid == 0 reproduces the resize branch of RTree.cpp:3431-3435
void orc_rtree_post_process(const void* h, unsigned char* image, int rows, int cols, double* com_pre, int com_pre_valid, int interval,
                            int tlx, int tly, int brx, int bry, double dist_to_pre_weight) {
    const Tree* t = (const Tree*)h;
    if (brx == -1) { brx = cols - 1; bry = rows - 1; }
    if (!com_pre_valid)
        for (int i = 0; i < t->num_parts; ++i) { com_pre[2 * i] = -1.; com_pre[2 * i + 1] = 0.; }
    if (t->part_map_type == 0) suppress_non_max(image, cols, interval, t->num_parts, tlx, tly, brx, bry, com_pre, dist_to_pre_weight);
    else remove_small(image, rows, cols, interval, tlx, tly, brx, bry);
    if (interval > 1) upscale(image, rows, cols, interval, tlx, tly, brx, bry);
}

}  // extern "C"
