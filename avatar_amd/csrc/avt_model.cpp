// avt_model.cpp — host-side, pose-independent model preparation behind avt_model_create().
//
// Replaces, for the hot path only:
//   * AvatarModel::AvatarModel data derivation (AvatarModel.cpp:74-127): assignedJoints (threshold 1e-12,
//     sorted by descending (weight, joint)), initialJointPos, jointShapeReg;
//   * the pose-independent tables of AvatarEvaluationCommonData (AvatarOptimizer.cpp:187-245): per-point
//     deduplicated ancestor lists and the shape tables S, Sp;
//   * GaussianMixture::load factorisations (GaussianMixture.cpp:44-76): Cholesky of the precision, consts.
// and lays everything out SoA for coalesced device access.
#include <exception>
#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <functional>
#include <limits>
#include <memory>
#include <numeric>

#include "avt_internal.h"

static thread_local std::string g_err;
void avt_set_error(const std::string& s) { g_err = s; }
extern "C" const char* avt_last_error(void) { return g_err.c_str(); }

static bool chol_lower(const double* A, int n, double* L) {
    std::fill(L, L + (size_t)n * n, 0.0);
    for (int j = 0; j < n; ++j) {
        double d = A[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) d -= L[(size_t)j * n + k] * L[(size_t)j * n + k];
        if (!(d > 0.0)) return false;
        const double l = std::sqrt(d);
        L[(size_t)j * n + j] = l;
        for (int i = j + 1; i < n; ++i) {
            double s = A[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) s -= L[(size_t)i * n + k] * L[(size_t)j * n + k];
            L[(size_t)i * n + j] = s / l;
        }
    }
    return true;
}

// -------------------------------------------------------------------------------------------------
// Column layout of the evaluation tile [J | r] (avt_eval.hip).  A model point's rows are non-zero only in the columns
// of its ancestors' rotations, the root translation, the shape keys and the residual, so the columns are grouped
// into the 16-wide MFMA tiles by branch of the kinematic tree and J^T J skips the tile pairs a batch of points does
// not touch:
//   tile 0: root translation, shape keys, root rotation (every point);
//   one tile per group of <= 5 joints, packed bottom-up along the tree (SMPL: spine+head, each arm, each leg);
//   the residual column rides in tile 0 if it has room, else in the group tile most vertices touch.
// Vertices are then ordered by the set of tiles they touch, so that batches of 16 matched points are (mostly)
// uniform.  Skeletons that do not pack into the tile grid keep the plain order (storage column = parameter index,
// every tile live).  k_reduce maps tile coordinates back to parameter indices: nothing outside k_eval sees the order.
// -------------------------------------------------------------------------------------------------
static void build_tile_layout(avt_model* m, const int* parent) {
    AvtDims& d = m->d;
    const int V = d.V, J = d.J, K = d.K, P = d.P, NT = d.NT, NC = P + 1;
    auto plain = [&]() {
        m->tile_col.assign((size_t)16 * NT, NC); m->tile_param.assign((size_t)16 * NT, -1);
        for (int tc = 0; tc < NC; ++tc) { m->tile_col[tc] = tc; m->tile_param[tc] = tc; }
        m->joint_col.resize(J);
        for (int j = 0; j < J; ++j) m->joint_col[j] = 3 + 3 * j;
        d.col_tr = 0; d.col_shape = 3 + 3 * J; d.col_res = P;
        m->vorder.resize(V); std::iota(m->vorder.begin(), m->vorder.end(), 0);
        m->vmask.assign(V, (unsigned short)((1u << NT) - 1));
    };
    if (6 + K > 16 || NT > 8) { plain(); return; }
    std::vector<std::vector<int>> children(J), groups;
    for (int j = 1; j < J; ++j) children[parent[j]].push_back(j);
    const size_t cap = 5;
    std::function<std::vector<int>(int)> pack = [&](int j) {
        std::vector<std::vector<int>> open;
        size_t total = 1;
        for (int c : children[j]) { open.push_back(pack(c)); total += open.back().size(); }
        std::stable_sort(open.begin(), open.end(), [](const std::vector<int>& a, const std::vector<int>& b) { return a.size() > b.size(); });
        size_t first = 0;
        while (total > cap && first < open.size()) { groups.push_back(open[first]); total -= open[first].size(); ++first; }
        std::vector<int> res{j};
        for (size_t i = first; i < open.size(); ++i) res.insert(res.end(), open[i].begin(), open[i].end());
        return res;
    };
    for (int c : children[0]) { std::vector<int> g = pack(c); if (!g.empty()) groups.push_back(g); }
    while ((int)groups.size() > NT - 1) {       // too many branches: merge the two smallest while they fit one tile
        std::stable_sort(groups.begin(), groups.end(), [](const std::vector<int>& a, const std::vector<int>& b) { return a.size() < b.size(); });
        if (groups[0].size() + groups[1].size() > cap) { plain(); return; }
        groups[1].insert(groups[1].end(), groups[0].begin(), groups[0].end());
        groups.erase(groups.begin());
    }
    for (auto& g : groups) std::sort(g.begin(), g.end());
    std::sort(groups.begin(), groups.end());            // by smallest joint id: a fixed, model-only order
    std::vector<int> tile_of(J, 0);
    for (size_t gi = 0; gi < groups.size(); ++gi)
        for (int j : groups[gi]) tile_of[j] = (int)gi + 1;
    // tiles a vertex touches (without the residual's tile), from its ancestor list
    std::vector<unsigned short> vm(V, 1);
    std::vector<int> touched(NT, 0);
    for (int v = 0; v < V; ++v) {
        for (int a = 0; a < m->anc_n[v]; ++a) vm[v] |= (unsigned short)(1u << tile_of[m->anc[(size_t)a * V + v] & 0xff]);
        for (int ti = 0; ti < NT; ++ti) touched[ti] += (vm[v] >> ti) & 1;
    }
    int res_tile = 0;
    if (6 + K >= 16) {
        res_tile = -1;
        for (size_t gi = 0; gi < groups.size(); ++gi)
            if (3 * groups[gi].size() < 16 && (res_tile < 0 || touched[gi + 1] > touched[res_tile])) res_tile = (int)gi + 1;
        if (res_tile < 0) { plain(); return; }
    }
    // storage order = tile order with the padding squeezed out
    m->tile_col.assign((size_t)16 * NT, NC); m->tile_param.assign((size_t)16 * NT, -1);
    m->joint_col.assign(J, 0);
    int store = 0;
    auto put = [&](int tile, int& fill, int param) { m->tile_col[16 * tile + fill] = store; m->tile_param[16 * tile + fill] = param; ++fill; return store++; };
    {
        int fill = 0;
        d.col_tr = store; for (int c = 0; c < 3; ++c) put(0, fill, c);
        d.col_shape = store; for (int k = 0; k < K; ++k) put(0, fill, 3 + 3 * J + k);
        m->joint_col[0] = store; for (int c = 0; c < 3; ++c) put(0, fill, 3 + c);
        if (res_tile == 0) d.col_res = put(0, fill, P);
    }
    for (size_t gi = 0; gi < groups.size(); ++gi) {
        int fill = 0;
        for (int j : groups[gi]) { m->joint_col[j] = store; for (int c = 0; c < 3; ++c) put((int)gi + 1, fill, 3 + 3 * j + c); }
        if (res_tile == (int)gi + 1) d.col_res = put((int)gi + 1, fill, P);
    }
    m->vmask.resize(V);
    for (int v = 0; v < V; ++v) m->vmask[v] = (unsigned short)(vm[v] | (1u << res_tile));
    m->vorder.resize(V); std::iota(m->vorder.begin(), m->vorder.end(), 0);
    std::stable_sort(m->vorder.begin(), m->vorder.end(), [&](int a, int b) { return m->vmask[a] < m->vmask[b]; });
}

// Which wave of k_eval's workgroup contracts which of the 21 tile pairs of the six-tile layout.  The pairs a batch of 16
// matched points touches depend on the tiles of its vertices (vmask), and a wave works through its live pairs one after
// the other (12 dependent matrix instructions each): the phase lasts as long as the busiest wave.  One pair - the most
// frequent one - is split four ways by k-steps; the other 20 are dealt 5 per wave by descending frequency onto the wave
// with the least expected load, then improved by pairwise swaps against the batches of the model's own vertex order
// (mean over batches of the busiest wave's pair count).  Deterministic; the default p mod 4 deal when NT != 6.
static void deal_tile_pairs(avt_model* m) {
    AvtDims& d = m->d;
    for (int w = 0; w < 4; ++w) d.pair_deal[w] = 0;
    for (int p = 0; p < 20; ++p) d.pair_deal[p % 4] |= (unsigned)p << (5 * (p / 4));
    d.pair_split = 20;
    if (d.NT != 6) return;
    const int V = d.V, NP = 21;
    int pti[NP], ptj[NP];
    for (int i = 0, p = 0; i < 6; ++i) for (int j = i; j < 6; ++j, ++p) { pti[p] = i; ptj[p] = j; }
    std::vector<std::array<unsigned char, NP>> live;
    for (int b = 0; b < V; b += 16) {
        unsigned mask = 0;
        for (int v = b; v < std::min(V, b + 16); ++v) mask |= m->vmask[m->vorder[v]];
        std::array<unsigned char, NP> l{};
        for (int p = 0; p < NP; ++p) l[p] = ((mask >> pti[p]) & (mask >> ptj[p]) & 1u) ? 1 : 0;
        live.push_back(l);
    }
    std::vector<double> freq(NP, 0.0);
    for (auto& l : live) for (int p = 0; p < NP; ++p) freq[p] += l[p];
    std::vector<int> order(NP);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return freq[a] > freq[b]; });
    int owner[NP];                                   // wave of every pair, 4 = split
    owner[order[0]] = 4;
    double wload[4] = {0, 0, 0, 0};
    int wcount[4] = {0, 0, 0, 0};
    for (int k = 1; k < NP; ++k) {
        int best = -1;
        for (int w = 0; w < 4; ++w) if (wcount[w] < 5 && (best < 0 || wload[w] < wload[best])) best = w;
        owner[order[k]] = best; wload[best] += freq[order[k]]; ++wcount[best];
    }
    auto cost = [&]() {
        long long total4 = 0;                         // 4 x (sum over batches of the busiest wave's load), the split pair counts 1/4
        for (auto& l : live) {
            int c[4] = {0, 0, 0, 0}, split = 0;
            for (int p = 0; p < NP; ++p) if (l[p]) { if (owner[p] == 4) split = 1; else c[owner[p]] += 4; }
            total4 += std::max(std::max(c[0], c[1]), std::max(c[2], c[3])) + split;
        }
        return total4;
    };
    long long cur = cost();
    for (bool improved = true; improved;) {
        improved = false;
        for (int a = 0; a < NP; ++a)
            for (int b = a + 1; b < NP; ++b) {
                if (owner[a] == owner[b]) continue;
                std::swap(owner[a], owner[b]);
                const long long c = cost();
                if (c < cur) { cur = c; improved = true; } else std::swap(owner[a], owner[b]);
            }
    }
    int fill[4] = {0, 0, 0, 0};
    for (int w = 0; w < 4; ++w) d.pair_deal[w] = 0;
    for (int p = 0; p < NP; ++p) {
        if (owner[p] == 4) d.pair_split = p;
        else d.pair_deal[owner[p]] |= (unsigned)p << (5 * fill[owner[p]]++);
    }
}

// -------------------------------------------------------------------------------------------------
// Static tables of the moment form of the data term (avt_moments.hip; tools/moment_proto2.py is the executable specification).
// Every row of [J | r] of a model point m is linear in psi_m = [base_m | keys_m | 1] with coefficients that depend on the state
// alone (AvatarOptimizer.cpp:507-582), so J^T J, J^T r and the cost are contractions of
//   T_kk' = sum_m c_m a_mk a_mk' psi_m psi_m^T     per unordered pair (k <= k') of joints assigned to a common vertex,
// accumulated once per ICP iteration.  Here: the pairs, per pair the vertices that carry both joints, psi, and the index lists of the
// tree sums the assembly runs (lever joints k under rotation joints j).
// -------------------------------------------------------------------------------------------------
static void build_moment_tables(avt_model* m) {
    AvtDims& d = m->d;
    const int V = d.V, J = d.J, K = d.K, S1 = K + 1;
    d.mom_npsi = 3 * S1 + 1;
    d.mom_ntp = (d.mom_npsi + 15) / 16;
    d.mom_ok = (S1 <= 16 && d.HS / 4 <= 22) ? 1 : 0;
    // psi
    const int PW = 16 * d.mom_ntp;
    m->mom_psi.assign((size_t)V * PW, 0.0);
    for (int v = 0; v < V; ++v) {
        double* ps = &m->mom_psi[(size_t)v * PW];
        for (int i = 0; i < 3; ++i) {
            ps[S1 * i] = m->shape_planes[((size_t)K * 3 + i) * V + v];                                   // s = 0: the base cloud
            for (int s = 0; s < K; ++s) ps[S1 * i + 1 + s] = m->shape_planes[((size_t)s * 3 + i) * V + v];
        }
        ps[3 * S1] = 1.0;
    }
    // pairs and their vertex lists
    std::vector<int> pid((size_t)J * J, -1);
    for (int v = 0; v < V; ++v)
        for (int a = 0; a < 4; ++a)
            for (int b = 0; b < 4; ++b) {
                if (!(m->asg_w[(size_t)a * V + v] > 0.0) || !(m->asg_w[(size_t)b * V + v] > 0.0)) continue;
                const int ja = m->asg_j[(size_t)a * V + v], jb = m->asg_j[(size_t)b * V + v];
                if (ja <= jb) pid[(size_t)ja * J + jb] = 0;
            }
    m->mom_pair.clear();
    for (int a = 0; a < J; ++a)
        for (int b = a; b < J; ++b)
            if (pid[(size_t)a * J + b] == 0) { pid[(size_t)a * J + b] = (int)m->mom_pair.size() / 2; m->mom_pair.push_back(a); m->mom_pair.push_back(b); }
    const int NP = (int)m->mom_pair.size() / 2;
    d.mom_np = NP;
    std::vector<std::vector<int>> lv(NP);
    std::vector<std::vector<double>> lw(NP);
    for (int v = 0; v < V; ++v)
        for (int a = 0; a < 4; ++a)
            for (int b = 0; b < 4; ++b) {
                const double wa = m->asg_w[(size_t)a * V + v], wb = m->asg_w[(size_t)b * V + v];
                if (!(wa > 0.0) || !(wb > 0.0)) continue;
                const int ja = m->asg_j[(size_t)a * V + v], jb = m->asg_j[(size_t)b * V + v];
                if (ja > jb || (ja == jb && a != b)) continue;
                const int p = pid[(size_t)ja * J + jb];
                lv[p].push_back(v); lw[p].push_back(wa); lw[p].push_back(wb);
            }
    m->mom_lstart.assign(NP + 1, 0); m->mom_lv.clear(); m->mom_lw.clear();
    d.mom_lmax = 0;
    for (int p = 0; p < NP; ++p) {
        m->mom_lv.insert(m->mom_lv.end(), lv[p].begin(), lv[p].end());
        m->mom_lw.insert(m->mom_lw.end(), lw[p].begin(), lw[p].end());
        m->mom_lstart[p + 1] = (int)m->mom_lv.size();
        d.mom_lmax = std::max(d.mom_lmax, (int)lv[p].size());
    }
    // tree lists
    auto under = [&](int k, int j) { for (; k >= 0; k = m->parent[k]) if (k == j) return true; return false; };
    auto first = [&](int op) { return m->mom_pair[2 * (op >> 1) + (op & 1)]; };
    auto second = [&](int op) { return m->mom_pair[2 * (op >> 1) + 1 - (op & 1)]; };
    auto exists = [&](int op) { return !(op & 1) || m->mom_pair[2 * (op >> 1)] != m->mom_pair[2 * (op >> 1) + 1]; };
    m->mom_opk_start.assign(J + 1, 0); m->mom_opk.clear();
    m->mom_sub_start.assign(J + 1, 0); m->mom_sub.clear();
    for (int k = 0; k < J; ++k) {
        for (int op = 0; op < 2 * NP; ++op) if (exists(op) && first(op) == k) m->mom_opk.push_back(op);
        m->mom_opk_start[k + 1] = (int)m->mom_opk.size();
        for (int c = 0; c < J; ++c) if (under(c, k)) m->mom_sub.push_back(c);
        m->mom_sub_start[k + 1] = (int)m->mom_sub.size();
    }
    // Index lists of the assembly (k_assemble), 16-bit, as ONE block the kernel copies into LDS: every list is padded so that it is
    // walked in fixed-size trips without bounds checks - with an index of an all-zero row (2 np for X16 rows, J for per-joint rows).
    //   opk   per lever joint k: its ordered pairs, padded to 4
    //   sub   per joint j: the joints under it (itself included), padded to 8
    //   seg   rot-rot: block (j <= j') sums the ordered pairs (k under j, k' under j'); a block's list is cut into SEGMENTS of 16 (padded)
    //         so that no thread walks more than one segment; bseg[b] .. bseg[b + 1]: the segments of block b; jj[b] = j | j' << 8.
    //         Only the blocks with such pairs are listed (left leg against right arm: none); z2: the others, structural zeros.
    {
        std::vector<int> st(J + 1, 0), li;
        for (int k = 0; k < J; ++k) {
            for (int e = m->mom_opk_start[k]; e < m->mom_opk_start[k + 1]; ++e) li.push_back(m->mom_opk[e]);
            while (li.size() % 4) li.push_back(2 * NP);
            st[k + 1] = (int)li.size();
        }
        m->mom_opk_start = st; m->mom_opk = li;
    }
    {
        std::vector<int> st(J + 1, 0), li;
        for (int k = 0; k < J; ++k) {
            for (int e = m->mom_sub_start[k]; e < m->mom_sub_start[k + 1]; ++e) li.push_back(m->mom_sub[e]);
            while (li.size() % 8) li.push_back(J);
            st[k + 1] = (int)li.size();
        }
        m->mom_sub_start = st; m->mom_sub = li;
    }
    m->mom_s2_start.assign(1, 0); m->mom_s2.clear(); m->mom_s2_jj.clear(); m->mom_z2_jj.clear();
    for (int j = 0; j < J; ++j)
        for (int jp = j; jp < J; ++jp) {
            std::vector<int> ops;
            for (int op = 0; op < 2 * NP; ++op) if (exists(op) && under(first(op), j) && under(second(op), jp)) ops.push_back(op);
            if (ops.empty()) { m->mom_z2_jj.push_back(j | (jp << 8)); continue; }      // a structural zero block
            while (ops.size() % 16) ops.push_back(2 * NP);
            m->mom_s2.insert(m->mom_s2.end(), ops.begin(), ops.end());
            m->mom_s2_start.push_back((int)m->mom_s2.size() / 16);      // in segments
            m->mom_s2_jj.push_back(j | (jp << 8));
        }
    d.mom_nz2 = (int)m->mom_z2_jj.size();
    d.mom_nb2 = (int)m->mom_s2_jj.size();
    d.mom_nseg = (int)m->mom_s2.size() / 16;
    {   // the listed blocks dealt to the rot-rot roles of k_assemble_parts in contiguous ranges of (nearly) equal segment counts
        const int NR = 3;
        for (int r = 0; r <= 4; ++r) d.mom_rsplit[r] = d.mom_nb2;
        d.mom_rsplit[0] = 0;
        for (int r = 1; r < NR; ++r) {
            const long long want = (long long)d.mom_nseg * r / NR;
            int b = d.mom_rsplit[r - 1];
            while (b < d.mom_nb2 && m->mom_s2_start[b] < want) ++b;
            d.mom_rsplit[r] = b;
        }
        d.mom_rr_doubles = 0;
        for (int r = 0; r < NR; ++r) {
            const int b0 = d.mom_rsplit[r], b1 = d.mom_rsplit[r + 1];
            d.mom_rr_doubles = std::max(d.mom_rr_doubles, 16 * ((m->mom_s2_start[b1] - m->mom_s2_start[b0]) + (b1 - b0)));
        }
    }
    d.mom_nm1 = 0;
    m->mom_m1_start.assign(1, 0); m->mom_m1.clear();
    d.mom_nopk = (int)m->mom_opk.size(); d.mom_nsub = (int)m->mom_sub.size(); d.mom_nm1l = 0; d.mom_ns2l = (int)m->mom_s2.size();
    {   // the block: [opk_start | opk | sub_start | sub | bseg | seg | jj], every part starting on a multiple of four words
        m->mom_tab16.clear();
        auto put = [&](const std::vector<int>& v, int& off) {
            while (m->mom_tab16.size() % 4) m->mom_tab16.push_back(0);
            off = (int)m->mom_tab16.size();
            for (int x : v) m->mom_tab16.push_back((unsigned short)x);
        };
        put(m->mom_opk_start, d.mom_toff[0]); put(m->mom_opk, d.mom_toff[1]); put(m->mom_sub_start, d.mom_toff[2]); put(m->mom_sub, d.mom_toff[3]);
        put(m->mom_s2_start, d.mom_toff[4]); put(m->mom_s2, d.mom_toff[5]); put(m->mom_s2_jj, d.mom_toff[6]);
        while (m->mom_tab16.size() % 4) m->mom_tab16.push_back(0);
        d.mom_toff[7] = (int)m->mom_tab16.size();
    }
    if ((int)m->mom_s2.size() > 65535 || 2 * NP + 1 > 65535) d.mom_ok = 0;      // 16-bit index lists
}

static int avt_model_create_impl(const avt_model_desc* desc, avt_model** out) {
    if (!desc || !out) { avt_set_error("avt_model_create: null argument"); return 1; }
    const int V = desc->num_points, J = desc->num_joints, K = desc->num_shape_keys, F = desc->num_faces;
    if (V <= 0 || J <= 0 || J > AVT_MAX_JOINTS || K < 0 || K > AVT_MAX_SHAPE || F < 0) {
        avt_set_error("avt_model_create: unsupported dimensions (J<=64, K<=16)");
        return 1;
    }
    if (3 + 3 * J + K > AVT_MAX_P) {  // k_solve: 256 threads up to P = 87, 1024 threads (packed factor in LDS) up to 179
        avt_set_error("avt_model_create: 3+3J+K must be <= 179 in this build (SMPL: 85, SMPL-H: 169, SMPL-X with 10 shape keys: 178)");
        return 1;
    }
    if (desc->parent[0] != -1) { avt_set_error("avt_model_create: parent[0] must be -1 (AvatarModel.cpp:41)"); return 1; }
    for (int j = 1; j < J; ++j)
        if (desc->parent[j] < 0 || desc->parent[j] >= j) { avt_set_error("avt_model_create: parent[] must be topologically sorted"); return 1; }
    std::unique_ptr<avt_model> holder(new avt_model());      // released to the caller only on success (also on an exception)
    avt_model* m = holder.get();
    AvtDims& d = m->d;
    d.V = V; d.J = J; d.K = K; d.F = F; d.P = 3 + 3 * J + K;
    d.NT = (d.P + 1 + AVT_TILE - 1) / AVT_TILE;
    d.NPAIR = d.NT * (d.NT + 1) / 2;
    d.xsize = 3 + 4 * J + K;
    d.prep_size = prep_total(d);
    d.rec_quad = 12 * d.K + 84;
    d.nb_max = (d.V + AVT_EVAL_PTS - 1) / AVT_EVAL_PTS;
    d.num_parts = 0;
    m->parent.assign(desc->parent, desc->parent + J);
    m->jlevel.assign(J, 0);
    d.nlevels = 1;
    for (int j = 1; j < J; ++j) { m->jlevel[j] = m->jlevel[desc->parent[j]] + 1; d.nlevels = std::max(d.nlevels, m->jlevel[j] + 1); }
    d.HS = 4 * ((d.P + 4) / 4);
    // per-level work lists for the forward-kinematics / shape-table pass of k_solve (avt_lm.hip): per joint 9 world-rotation
    // entries, 3 world-origin entries and 3K shape-table entries, each one B[out] = B[rp..rp+2] . B[v, v+st, v+2st] + B[add]
    {
        const PrepLayout L = prep_layout(J, K, d.xsize);
        if (L.ndoubles >= 16384) { avt_set_error("avt_model_create: skeleton scratch exceeds the 14-bit item offsets"); return 1; }
        auto item = [&](int rp, int v, int scode, int add, int out) {
            m->fk_items.push_back(rp | (v << 14) | (scode << 28));
            m->fk_items.push_back(add | (out << 14));
        };
        m->fk_level_off.assign(d.nlevels + 1, 0);
        for (int lv = 0; lv < d.nlevels; ++lv) {
            for (int j = 0; j < J; ++j) {
                if (m->jlevel[j] != lv) continue;
                const int pa = desc->parent[j];
                const int Rp = pa < 0 ? L.ident : L.Rw + 9 * pa;       // R(-1, parent): identity above the root
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 3; ++c) item(Rp + 3 * r, L.rot + 9 * j + c, 2, L.zero, L.Rw + 9 * j + 3 * r + c);
                for (int r = 0; r < 3; ++r) item(Rp + 3 * r, L.dv + 3 * j, 1, pa < 0 ? L.zero : L.o + 3 * pa + r, L.o + 3 * j + r);
                for (int r = 0; r < 3; ++r)
                    for (int k = 0; k < K; ++k) {
                        if (pa < 0) item(L.ident + 3 * r, L.zero, 0, L.zero, L.H + r * K + k);               // H[root] = 0
                        else item(Rp + 3 * r, L.Sp + j * 3 * K + k, 3, L.H + pa * 3 * K + r * K + k, L.H + j * 3 * K + r * K + k);
                    }
            }
            m->fk_level_off[lv + 1] = (int)m->fk_items.size() / 2;
        }
        // the same items by (level, thread) for the 256-thread skeleton pass (avt_prep.h)
        d.fk_reg = d.nlevels <= AVT_PREP_LEVELS_REG;
        for (int lv = 0; lv < d.nlevels; ++lv) d.fk_reg = d.fk_reg && m->fk_level_off[lv + 1] - m->fk_level_off[lv] <= AVT_PREP_TITEM_THREADS;
        m->fk_titems.assign((size_t)AVT_PREP_LEVELS_REG * AVT_PREP_TITEM_THREADS * 2, -1);
        if (d.fk_reg)
            for (int lv = 0; lv < d.nlevels; ++lv)
                for (int i = m->fk_level_off[lv]; i < m->fk_level_off[lv + 1]; ++i) {
                    // (the threads of the LAST waves first: wave 0 runs the back substitution in front of the pass and does the retraction's quaternions)
                    const int th = AVT_PREP_TITEM_THREADS - 1 - (i - m->fk_level_off[lv]);
                    m->fk_titems[((size_t)lv * AVT_PREP_TITEM_THREADS + th) * 2] = m->fk_items[2 * (size_t)i];
                    m->fk_titems[((size_t)lv * AVT_PREP_TITEM_THREADS + th) * 2 + 1] = m->fk_items[2 * (size_t)i + 1];
                }
    }

    // shape planes
    m->shape_planes.assign((size_t)(K + 1) * 3 * V, 0.0);
    for (int k = 0; k < K; ++k)
        for (int v = 0; v < V; ++v)
            for (int c = 0; c < 3; ++c)
                m->shape_planes[((size_t)k * 3 + c) * V + v] = desc->key_clouds[(size_t)k * 3 * V + 3 * v + c];
    for (int v = 0; v < V; ++v)
        for (int c = 0; c < 3; ++c) m->shape_planes[((size_t)K * 3 + c) * V + v] = desc->base_cloud[3 * v + c];

    // skinning weights: raw CSC (<=4 nnz) and assignedJoints
    m->lbs_w.assign((size_t)4 * V, 0.0); m->lbs_j.assign((size_t)4 * V, 0);
    m->asg_w.assign((size_t)4 * V, 0.0); m->asg_j.assign((size_t)4 * V, 0);
    m->main_joint.assign(V, 0);
    std::vector<std::vector<std::pair<double, int>>> assigned(V);
    for (int v = 0; v < V; ++v) {
        const int b = desc->weights_colptr[v], e = desc->weights_colptr[v + 1];
        if (e - b > AVT_MAX_ASSIGN) {
            avt_set_error("avt_model_create: more than 4 skinning weights on a vertex (MAX_ASSIGN, AvatarOptimizer.cpp:164)");
            return 1;
        }
        for (int i = b; i < e; ++i) {
            const int j = desc->weights_row[i];
            if (j < 0 || j >= J) { avt_set_error("avt_model_create: weight row out of range"); return 1; }
            m->lbs_w[(size_t)(i - b) * V + v] = desc->weights_val[i];
            m->lbs_j[(size_t)(i - b) * V + v] = j;
            if (desc->weights_val[i] > 1e-12) assigned[v].push_back({desc->weights_val[i], j});
        }
        std::sort(assigned[v].begin(), assigned[v].end(), std::greater<std::pair<double, int>>());
        if (assigned[v].empty()) { avt_set_error("avt_model_create: vertex without skinning weights"); return 1; }
        if (desc->limit_one_joint_per_point) { assigned[v].resize(1); assigned[v][0].first = 1.0; }      // AvatarModel.cpp:190-196
        for (size_t a = 0; a < assigned[v].size(); ++a) {
            m->asg_w[a * V + v] = assigned[v][a].first;
            m->asg_j[a * V + v] = assigned[v][a].second;
        }
        m->main_joint[v] = assigned[v][0].second;
    }

    // joint regression (AvatarModel.cpp:112-127)
    m->jsr_base.assign(3 * J, 0.0);
    m->jsr.assign((size_t)3 * J * K, 0.0);
    for (int j = 0; j < J; ++j)
        for (int e = desc->jreg_colptr[j]; e < desc->jreg_colptr[j + 1]; ++e) {
            const int v = desc->jreg_row[e];
            const double wt = desc->jreg_val[e];
            for (int c = 0; c < 3; ++c) m->jsr_base[3 * j + c] += desc->base_cloud[3 * v + c] * wt;
            for (int k = 0; k < K; ++k)
                for (int c = 0; c < 3; ++c)
                    m->jsr[(size_t)(3 * j + c) * K + k] += desc->key_clouds[(size_t)k * 3 * V + 3 * v + c] * wt;
        }
    if ((desc->joint_shape_reg_base != nullptr) != (desc->joint_shape_reg != nullptr)) {
        avt_set_error("avt_model_create: joint_shape_reg_base and joint_shape_reg come together");
        return 1;
    }
    if (desc->joint_shape_reg) {       // joint_shape_regressor.txt of the legacy format (AvatarModel.cpp:231-243): taken as given
        for (int i = 0; i < 3 * J; ++i) {
            m->jsr_base[i] = desc->joint_shape_reg_base[i];
            for (int k = 0; k < K; ++k) m->jsr[(size_t)i * K + k] = desc->joint_shape_reg[(size_t)k * 3 * J + i];
        }
    }
    // S, Sp (AvatarOptimizer.cpp:215-245)
    m->S.assign((size_t)J * 3 * K, 0.0); m->Sp.assign((size_t)J * 3 * K, 0.0);
    for (int j = 0; j < J; ++j)
        for (int c = 0; c < 3; ++c)
            for (int k = 0; k < K; ++k) m->S[((size_t)j * 3 + c) * K + k] = m->jsr[(size_t)(3 * j + c) * K + k];
    for (int j = 1; j < J; ++j)
        for (int e = 0; e < 3 * K; ++e)
            m->Sp[(size_t)j * 3 * K + e] = m->S[(size_t)j * 3 * K + e] - m->S[(size_t)desc->parent[j] * 3 * K + e];

    // ancestors (AvatarOptimizer.cpp:187-213): union of root chains of the assigned joints, sorted by id; for
    // each ancestor a bit mask of the assigned joints beneath (or equal to) it.
    m->anc_n.assign(V, 0);
    m->anc.assign((size_t)AVT_ANC_MAX * V, 0);
    int anc_max = 0;
    for (int v = 0; v < V; ++v) {
        unsigned mask[AVT_MAX_JOINTS] = {0};     // per joint: which of the <= 4 assigned joints lie under it
        for (size_t a = 0; a < assigned[v].size(); ++a)
            for (int j = assigned[v][a].second; j != -1; j = desc->parent[j]) mask[j] |= 1u << a;
        int n = 0;
        for (int j = 0; j < J; ++j)
            if (mask[j]) {
                if (n >= AVT_ANC_MAX) { avt_set_error("avt_model_create: more than 16 ancestors on a vertex"); return 1; }
                m->anc[(size_t)n * V + v] = (unsigned short)(j | (mask[j] << 8));
                ++n;
            }
        m->anc_n[v] = (unsigned char)n;
        anc_max = std::max(anc_max, n);
    }
    d.anc_max = anc_max;
    build_tile_layout(m, desc->parent);
    d.res_tile = 0;
    for (int tc = 0; tc < 16 * d.NT; ++tc) if (m->tile_param[tc] == d.P) d.res_tile = tc / 16;
    d.res_pair = 0;
    for (int i = 0; i < d.res_tile; ++i) d.res_pair += d.NT - i;
    d.res_elem = 0;            // element e of a partial tile: row (e >> 4 & 3) + 4 (e >> 6), column e & 15
    for (int rr = 0; rr < 16; ++rr) if (m->tile_param[d.res_tile * 16 + rr] == d.P) d.res_elem = ((rr >> 2) << 6) | ((rr & 3) << 4) | rr;
    deal_tile_pairs(m);
    m->deal_col.assign(4 * 6 * 2 * 16, d.P + 1);
    if (d.NT == 6)
        for (int w = 0; w < 4; ++w)
            for (int i = 0; i < 6; ++i) {
                int p = i < 5 ? (int)((d.pair_deal[w] >> (5 * i)) & 31u) : d.pair_split, ti = 0;
                while (p >= 6 - ti) { p -= 6 - ti; ++ti; }
                if (p == 0) d.pair_deal[w] |= 1u << (25 + i);      // bits 25..30: pair i of the wave (30: the split pair) is a diagonal one
                for (int l = 0; l < 16; ++l) {
                    m->deal_col[((w * 6 + i) * 2 + 0) * 16 + l] = m->tile_col[ti * 16 + l];
                    m->deal_col[((w * 6 + i) * 2 + 1) * 16 + l] = m->tile_col[(ti + p) * 16 + l];
                }
            }
    for (int ti = 0; ti < AVT_MAX_TILES; ++ti) {
        int lo = d.P + 1, hi = -1;
        if (ti < d.NT)
            for (int i = 0; i < 16; ++i) { const int c = m->tile_col[ti * 16 + i]; if (c <= d.P) { lo = std::min(lo, c); hi = std::max(hi, c); } }
        d.tile_zpass[ti] = hi >= lo ? (((2ull << (hi / 5)) - 1ull) & ~((1ull << (lo / 5)) - 1ull)) : 0ull;
    }

    // per-vertex static records (field order of the matched-point records, avt_eval.hip): 3(K+1) shape-plane values, 3 zeros
    // (mean data point), 1 zero (sqrt count), 4 weights; then 20 ints: 4 assigned joints, 16 ancestor words
    // joint | mask << 8 | (parent + 1) << 16 | storage column << 24
    {
        const int RV = d.rec_quad / 4, ND = 3 * K + 11;
        m->vrec.assign((size_t)V * RV, 0.0);
        for (int v = 0; v < V; ++v) {
            double* r = &m->vrec[(size_t)v * RV];
            for (int fld = 0; fld < 3 * (K + 1); ++fld) r[fld] = m->shape_planes[(size_t)fld * V + v];
            for (int a = 0; a < 4; ++a) r[3 * K + 7 + a] = m->asg_w[(size_t)a * V + v];
            int* ri = (int*)(r + ND);
            for (int a = 0; a < 4; ++a) ri[a] = m->asg_j[(size_t)a * V + v];
            for (int a = 0; a < (int)m->anc_n[v]; ++a) {
                const int w = (int)m->anc[(size_t)a * V + v], j = w & 0xff;
                ri[4 + a] = w | ((desc->parent[j] + 1) << 16) | (m->joint_col[j] << 24);
            }
        }
    }

    // mesh SoA
    m->mesh_soa.assign((size_t)3 * F, 0);
    for (int f = 0; f < F; ++f)
        for (int c = 0; c < 3; ++c) {
            const int idx = desc->mesh[3 * f + c];
            if (idx < 0 || idx >= V) { avt_set_error("avt_model_create: mesh index out of range"); return 1; }
            m->mesh_soa[(size_t)c * F + f] = idx;
        }

    // GMM (GaussianMixture.cpp:12-77)
    d.ncomps = desc->prior_ncomps > 0 ? desc->prior_ncomps : 0;
    if (d.ncomps > AVT_MAX_COMPS) { avt_set_error("avt_model_create: more than 16 GMM components"); return 1; }
    d.ndims = d.ncomps ? desc->prior_ndims : 0;
    if (d.ncomps) {
        const int n = d.ndims;
        if (n != 3 * (J - 1)) { avt_set_error("avt_model_create: prior_ndims must be 3*(J-1)"); return 1; }
        m->prior_mean.assign(desc->prior_mean, desc->prior_mean + (size_t)d.ncomps * n);
        m->prior_prec.assign((size_t)d.ncomps * n * n, 0.0);
        m->prior_L.assign((size_t)d.ncomps * n * n, 0.0);
        m->prior_clog.assign(d.ncomps, 0.0);
        const double log_sqrt_2_pi_n = n * 0.5 * std::log(2 * M_PI);
        double minDet = std::numeric_limits<double>::max();
        std::vector<double> L((size_t)n * n), Li((size_t)n * n);
        for (int c = 0; c < d.ncomps; ++c) {
            m->prior_clog[c] = std::log(desc->prior_weight[c]) - log_sqrt_2_pi_n;
            if (!chol_lower(desc->prior_cov + (size_t)c * n * n, n, L.data())) {
                avt_set_error("avt_model_create: prior covariance not positive definite (\"Decomposition failed!\")");
                return 1;
            }
            std::fill(Li.begin(), Li.end(), 0.0);  // L^-1 by forward substitution
            for (int col = 0; col < n; ++col)
                for (int i = col; i < n; ++i) {
                    double s = (i == col) ? 1.0 : 0.0;
                    for (int k = col; k < i; ++k) s -= L[(size_t)i * n + k] * Li[(size_t)k * n + col];
                    Li[(size_t)i * n + col] = s / L[(size_t)i * n + i];
                }
            double* prec = &m->prior_prec[(size_t)c * n * n];
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) {
                    double s = 0.0;
                    for (int k = std::max(i, j); k < n; ++k) s += Li[(size_t)k * n + i] * Li[(size_t)k * n + j];
                    prec[(size_t)i * n + j] = s;
                }
            if (!chol_lower(prec, n, &m->prior_L[(size_t)c * n * n])) {
                avt_set_error("avt_model_create: precision factorisation failed");
                return 1;
            }
            double det = 1.0;
            for (int i = 0; i < n; ++i) det *= L[(size_t)i * n + i];
            minDet = std::min(minDet, det);
            m->prior_clog[c] -= std::log(det);
        }
        for (int c = 0; c < d.ncomps; ++c) m->prior_clog[c] += std::log(minDet);
        // the solve kernel uses precision = Lp Lp^T recomposed from the factor the reference keeps (prec_cho)
        for (int c = 0; c < d.ncomps; ++c) {
            const double* Lp = &m->prior_L[(size_t)c * n * n];
            double* prec = &m->prior_prec[(size_t)c * n * n];
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j) {
                    double s = 0.0;
                    for (int k = 0; k <= std::min(i, j); ++k) s += Lp[(size_t)i * n + k] * Lp[(size_t)j * n + k];
                    prec[(size_t)i * n + j] = s;
                }
        }
    }
    build_moment_tables(m);
    *out = holder.release();
    return 0;
}

extern "C" void avt_model_destroy(avt_model* m) { delete m; }

extern "C" int avt_model_dims(const avt_model* m, int* V, int* J, int* K, int* F, int* P) {
    if (!m) { avt_set_error("avt_model_dims: null model"); return 1; }
    if (V) *V = m->d.V;
    if (J) *J = m->d.J;
    if (K) *K = m->d.K;
    if (F) *F = m->d.F;
    if (P) *P = m->d.P;
    return 0;
}

extern "C" int avt_model_main_joint(const avt_model* m, int* out) {
    if (!m || !out) { avt_set_error("avt_model_main_joint: null argument"); return 1; }
    std::copy(m->main_joint.begin(), m->main_joint.end(), out);
    return 0;
}

extern "C" int avt_model_joint_regression(const avt_model* m, double* ijp, double* jsr_colmajor) {
    if (!m) { avt_set_error("avt_model_joint_regression: null model"); return 1; }
    const int J = m->d.J, K = m->d.K;
    if (ijp) std::copy(m->jsr_base.begin(), m->jsr_base.end(), ijp);
    if (jsr_colmajor)
        for (int i = 0; i < 3 * J; ++i)
            for (int k = 0; k < K; ++k) jsr_colmajor[(size_t)k * 3 * J + i] = m->jsr[(size_t)i * K + k];
    return 0;
}

extern "C" void avt_options_default(avt_options* o) {
    if (!o) return;
    std::memset(o, 0, sizeof(*o));
    o->beta_pose = 0.1; o->beta_shape = 1.0; o->nn_step = 20; o->max_iters_per_icp = 10; o->enable_occlusion = 1;
    o->icp_iters = 1; o->num_threads = 4;
    // the step rule: the gain-ratio schedule ends at a lower objective than the fixed factors on every bench seed and accepts
    // 0.87 instead of 0.57 of its iterations, so it is the default although a rejected iteration is the cheaper one to count (Weak 3)
    o->lm_policy = AVT_LM_GAIN_RATIO; o->lm_up = AVT_LM_UP_GAIN_RATIO;
    o->lm_lambda0 = 1e-3; o->lm_down = 1.0 / 3.0; o->lm_lambda_min = 1e-12; o->lm_lambda_max = 1e8;
    o->function_tolerance = 1e-4;      // AvatarOptimizer.cpp:1333
}

extern "C" void avt_options_fixed_factors(avt_options* o) {
    if (!o) return;
    avt_options_default(o);
    o->lm_policy = AVT_LM_FIXED_FACTORS; o->lm_up = AVT_LM_UP_FIXED_FACTORS;
}

extern "C" const char* avt_kernel_name(int k) {
    static const char* names[AVT_K_COUNT] = {"lbs", "visibility", "bucket", "nn", "aggregate", "prepare", "eval", "reduce", "solve", "decide", "moments"};
    return (k >= 0 && k < AVT_K_COUNT) ? names[k] : "?";
}

extern "C" int avt_model_tile_layout(const avt_model* m, int* ntiles, int* tile_param, unsigned short* vertex_tiles, int* vertex_order) {
    if (!m) { avt_set_error("avt_model_tile_layout: null model"); return 1; }
    if (ntiles) *ntiles = m->d.NT;
    if (tile_param) std::copy(m->tile_param.begin(), m->tile_param.end(), tile_param);
    if (vertex_tiles) std::copy(m->vmask.begin(), m->vmask.end(), vertex_tiles);
    if (vertex_order) std::copy(m->vorder.begin(), m->vorder.end(), vertex_order);
    return 0;
}

// ---- exported entry points of the functions above: no C++ exception crosses the C ABI
extern "C" {
int avt_model_create(const avt_model_desc* desc, avt_model** out) {
    try { return avt_model_create_impl(desc, out); }
    catch (const std::exception& e) { avt_set_error(std::string("avt_model_create: ") + e.what()); return 1; }
    catch (...) { avt_set_error("avt_model_create: unknown exception"); return 1; }
}
}  // extern "C"
