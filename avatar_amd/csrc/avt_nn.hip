// avt_nn.hip — brute-force, LDS-tiled, per-part exact nearest neighbour (replaces nanoflann in
// findNN(..., invert=true), AvatarOptimizer.cpp:841-907).
//
// THIS TRANSLATION UNIT IS COMPILED WITH -ffp-contract=off.  The squared distance is evaluated exactly like
// nanoflann's L2_Simple_Adaptor::evalMetric (nanoflann.hpp:432-440) in an x86-64 baseline build (no FMA,
// CMakeLists.txt:37): r = 0; r += d0*d0; r += d1*d1; r += d2*d2, each product and sum rounded separately,
// compared with strict '<' (KNNResultSet::addPoint, nanoflann.hpp:175-199) while scanning the part's visible
// model points in ascending vertex order (the order findNN compacts them in, :873-878).  For inputs without
// exactly tied distances this returns the same index, bit for bit, as the KD-tree search.
//
// Layout: data points are bucketed by part (k_bucket), model points are stored in part-sorted SoA
// (pcx/pcy/pcz written by k_lbs).  A workgroup owns 256 consecutive bucketed data points (one per lane);
// the candidate range of the parts those points span is contiguous in the part-sorted model arrays and is
// streamed through LDS in tiles of 1024 candidates (24 KB).  Lanes of a wave almost always share a part, so
// the LDS reads are broadcasts.  Work is fp64-VALU bound: 8 flops + one v_min_f64 per candidate, one compare/select per
// group of 8 candidates (the position inside the winning group is resolved after the scan).
//
// The kernel also accumulates, per matched model vertex, the correspondence count and the fixed-point
// (2^40, frame-centred) sum of its data points with integer atomics: order-independent, hence bit-wise
// reproducible run to run (k_finalize turns them into sqrt(count) and the mean data point).
#include "avt_device.h"
#include "avt_bucket.h"

#define NN_TILE 1024
#define NN_SORTED_FLAG 0x40000000      // in FrameBuffers::vcount: the part's compacted candidates are sorted by (y, vertex id)
#define NN_SORT_CAP 1024               // largest part k_compact sorts (bitonic sort in LDS)
#ifndef NN_PART_NQ
#define NN_PART_NQ 2                  // k_nn_part: queries per lane (every candidate fetched is used for that many distances)
#endif
#ifndef NN_PART_GROUP
#define NN_PART_GROUP 4               // k_nn_part: candidates per group of the scan (one compare against the best per group)
#endif
#define NN_ACC_CAP 512                // candidates of a part whose match counts / sums k_nn_part accumulates in LDS (14 KB)
#ifndef NN_SLAB_CHUNK
#define NN_SLAB_CHUNK 16                // candidates per side and round of the slab scan (a multiple of the group size; with two queries per lane 16 / 24 / 32 / 48: 498 / 500 / 507 / 513 us per 256-frame launch of the class)
#endif


// round-to-nearest-even of x (|x| < 2^51) as an integer: the low mantissa bits of x + 1.5*2^52 (what __double2ll_rn
// returns, in two instructions instead of a conversion sequence)
__device__ __forceinline__ long long rint_to_ll(double x) {
    const double C = 6755399441055744.0;
    return __double_as_longlong(x + C) - __double_as_longlong(C);
}

// ---- correspondence bookkeeping.  Neighbouring pixels usually hit the same model vertex, so runs of equal
// vertices among the wave's consecutive queries are merged first (segmented inclusive scan over the query lanes,
// integer adds: order-independent) and only the last lane of each run issues the global atomics (which are what this
// costs: a scan over runs of at most 8 queries with DPP row shifts instead of LDS permutes measured slower).
template <int LANES>
__device__ __forceinline__ void nn_record(const FrameBuffers& fb, const AvtFrameCtl& ctl, int f, int V, size_t base, int s, bool active, int sub,
                                          int mv, double a0, double a1, double a2) {
    // mv: the matched model vertex of this lane's query (-1: none)
    const bool qlane = active && sub == 0;
    const int m = qlane ? mv : -1;
    if (qlane) {
        fb.corr_sorted[base + s] = m;
        fb.corr[base + fb.dorig[base + s]] = m;
    }
    int cnt = (m >= 0) ? 1 : 0;
    long long s0q = 0, s1q = 0, s2q = 0;
    if (m >= 0) {
        s0q = rint_to_ll((a0 - ctl.centre[0]) * AVT_FIX_SCALE);
        s1q = rint_to_ll((a1 - ctl.centre[1]) * AVT_FIX_SCALE);
        s2q = rint_to_ll((a2 - ctl.centre[2]) * AVT_FIX_SCALE);
    }
#ifdef AVT_NN_NO_MERGE       // (timing experiment: every query lane issues its own atomics, no in-wave merge)
    if (sub == 0 && m >= 0) {
        atomicAdd(fb.cnt + (size_t)f * V + m, cnt);
        unsigned long long* fs = (unsigned long long*)(fb.fsum + (size_t)f * 3 * V);
        atomicAdd(fs + m, (unsigned long long)s0q);
        atomicAdd(fs + (size_t)V + m, (unsigned long long)s1q);
        atomicAdd(fs + 2 * (size_t)V + m, (unsigned long long)s2q);
    }
    return;
#endif
    const int ql = lane_id() / LANES;                     // index of my query among the wave's 64/LANES queries
    const int mprev = __shfl_up(m, LANES, 64);
    bool head = (ql == 0) || (mprev != m);
#pragma unroll
    for (int dq = 1; dq < 64 / LANES; dq <<= 1) {
        const int d = dq * LANES;
        const int c_up = __shfl_up(cnt, d, 64);
        const long long a_up = __shfl_up(s0q, d, 64), b_up = __shfl_up(s1q, d, 64), e_up = __shfl_up(s2q, d, 64);
        const int h_up = __shfl_up((int)head, d, 64);
        if (ql >= dq) {
            if (!head) { cnt += c_up; s0q += a_up; s1q += b_up; s2q += e_up; }
            head = head || (h_up != 0);
        }
    }
    const int mnext = __shfl_down(m, LANES, 64);
    const bool tail = (ql == 64 / LANES - 1) || (mnext != m);
#ifndef AVT_NN_NO_ATOMICS      // (timing experiment only, tools/nn_atomics_probe.sh: what the integer atomics cost; results are wrong without them)
    if (sub == 0 && m >= 0 && tail) {
        atomicAdd(fb.cnt + (size_t)f * V + m, cnt);
        unsigned long long* fs = (unsigned long long*)(fb.fsum + (size_t)f * 3 * V);
        atomicAdd(fs + m, (unsigned long long)s0q);
        atomicAdd(fs + (size_t)V + m, (unsigned long long)s1q);
        atomicAdd(fs + 2 * (size_t)V + m, (unsigned long long)s2q);
    }
#endif
}

// LANES lanes cooperate on one query (4: low-latency single-frame shape; 1: throughput shape for large batches)
template <int LANES>
__global__ __launch_bounds__(256) void k_nn(DeviceModel dm, FrameBuffers fb) {
    constexpr int NN_QPB = 256 / LANES;
    const int f = blockIdx.y + fb.f0, t = threadIdx.x;
    const int V = dm.d.V, np = dm.d.num_parts;
    const AvtFrameCtl& ctl = fb.ctl[f];
    const int* po = fb.part_off + (size_t)f * (np + 1);
    const int nvalid = po[np];  // bucketed points with a valid label
    const int s0 = blockIdx.x * NN_QPB;
    if (s0 >= nvalid) return;
    const size_t base = (size_t)f * fb.max_points;
    const int sub = t % LANES;
    const int s = s0 + t / LANES;
    const bool active = s < nvalid;

    __shared__ double c_x[NN_TILE], c_y[NN_TILE], c_z[NN_TILE];
    __shared__ int s_qlo, s_qhi;
    // part of the first / last point of this workgroup (binary search over part_off)
    if (t == 0) {
        int lo = 0, hi = np - 1;
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (po[mid] <= s0) lo = mid; else hi = mid - 1; }
        s_qlo = lo;
        const int last = min(s0 + NN_QPB - 1, nvalid - 1);
        lo = 0; hi = np - 1;
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (po[mid] <= last) lo = mid; else hi = mid - 1; }
        s_qhi = lo;
    }
    __syncthreads();
    const int qlo = s_qlo, qhi = s_qhi;
    // my part: walk up from qlo (a workgroup rarely spans more than 2 parts)
    int q = qlo;
    if (active) while (q < qhi && po[q + 1] <= s) ++q;
    const int my_b = dm.part_start[q], my_e = my_b + fb.vcount[(size_t)f * np + q];   // visible candidates of my part
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    if (active) { a0 = fb.dx[base + s]; a1 = fb.dy[base + s]; a2 = fb.dz[base + s]; }

    const int cb = dm.part_start[qlo], ce = dm.part_start[qhi] + fb.vcount[(size_t)f * np + qhi];
    const double* pcx = fb.vcx + (size_t)f * V;
    const double* pcy = fb.vcy + (size_t)f * V;
    const double* pcz = fb.vcz + (size_t)f * V;
    double best = 1.7976931348623157e308;  // numeric_limits<double>::max(), KNNResultSet::init
    // The scan keeps the smallest distance and the START of the group of NN_GROUP candidates that first reached it (one compare
    // and one select per group instead of per candidate); which member of that group it was is settled once, after the scan.
    // Equivalent to one ascending scan with strict '<': a group replaces the running minimum only if its own minimum is
    // strictly smaller, and inside the group the first member that attains it wins.
    constexpr int NN_GROUP = 8;
    int gpos = -1;                          // absolute position (part-sorted arrays) of the winning group's first candidate
    auto dist2 = [&](double cx, double cy, double cz) {
        const double d0 = a0 - cx, d1 = a1 - cy, d2 = a2 - cz;
        double r = d0 * d0;                 // (0 + d0*d0) == d0*d0 exactly
        r = r + d1 * d1;
        r = r + d2 * d2;
        return r;
    };
    for (int tb = cb; tb < ce; tb += NN_TILE) {
        const int tn = min(NN_TILE, ce - tb);
        __syncthreads();
        for (int e = t; e < tn; e += 256) {
            const int pos = tb + e;
            c_x[e] = pcx[pos];
            c_y[e] = pcy[pos];
            c_z[e] = pcz[pos];
        }
        __syncthreads();
        if (active) {
            const int b = max(my_b, tb) - tb, e = min(my_e, tb + tn) - tb;
            // lane `sub` of the query's LANES-lane group scans candidates b+sub, b+sub+LANES, ... in ascending order
            int c = b + sub;
            for (; c + (NN_GROUP - 1) * LANES < e; c += NN_GROUP * LANES) {
                double r[NN_GROUP];
#pragma unroll
                for (int u = 0; u < NN_GROUP; ++u) r[u] = dist2(c_x[c + u * LANES], c_y[c + u * LANES], c_z[c + u * LANES]);
#pragma unroll
                for (int w = 1; w < NN_GROUP; w <<= 1)
#pragma unroll
                    for (int u = 0; u + w < NN_GROUP; u += 2 * w) r[u] = __builtin_fmin(r[u], r[u + w]);
                gpos = (r[0] < best) ? tb + c : gpos;
                best = __builtin_fmin(best, r[0]);
            }
            for (; c < e; c += LANES) {     // the tail of the range: groups of one
                const double r = dist2(c_x[c], c_y[c], c_z[c]);
                gpos = (r < best) ? tb + c : gpos;
                best = __builtin_fmin(best, r);
            }
        }
    }
    // which member of the winning group: the first whose distance IS the minimum (same operations on the same values, read
    // back from the part-sorted arrays: the group's tile may have left the LDS)
    int bi = 0x7fffffff;
    if (gpos >= 0) {
        double r[NN_GROUP];
#pragma unroll
        for (int u = 0; u < NN_GROUP; ++u) {
            const int pos = min(gpos + u * LANES, my_e - 1);
            r[u] = dist2(pcx[pos], pcy[pos], pcz[pos]);
        }
#pragma unroll
        for (int u = NN_GROUP - 1; u >= 0; --u)
            if (gpos + u * LANES < my_e && r[u] == best) bi = gpos + u * LANES;
    }
    // combine the 4 sub-scans: smallest distance, ties to the lowest candidate position — exactly the winner of
    // one ascending scan with strict '<'
#pragma unroll
    for (int m = 1; m < LANES; m <<= 1) {
        const double ob = __shfl_xor(best, m, 64);
        const int oi = __shfl_xor(bi, m, 64);
        if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    const int mv = (active && sub == 0 && bi != 0x7fffffff) ? fb.vcid[(size_t)f * V + bi] : -1;
    nn_record<LANES>(fb, ctl, f, V, base, s, active, sub, mv, a0, a1, a2);
}

// -------------------------------------------------------------------------------------------------
// k_nn_vis: k_nn<LANES> with the compaction of the visible model points inside it (few frames, inside optimize()): every
// workgroup compacts the tile of part-sorted model points it is about to scan - positions, vertex ids and the part-sorted
// visibility flags k_visibility / k_lbs keep (vis_sorted) - into its LDS instead of reading what a k_compact launch wrote:
// one launch and one pass through memory less on the dependency chain of a frame.  A tile is NN_TILE consecutive positions of
// the part-sorted arrays (all points, visible or not); thread t owns positions 4t .. 4t+3, an order-preserving compaction
// (wave scan of the per-thread counts) keeps ascending vertex order inside every part, s_pre[p] = number of visible points
// before position p of the tile gives a part's candidate range.  Same scan, same tie rule and the same result as k_compact +
// k_nn; the member of the winning group is settled at the end of the tile that produced it (the tile then leaves the LDS).
// -------------------------------------------------------------------------------------------------
template <int LANES>
__global__ __launch_bounds__(256) void k_nn_vis(DeviceModel dm, FrameBuffers fb) {
    constexpr int NN_QPB = 256 / LANES;
    const int f = blockIdx.y + fb.f0, t = threadIdx.x;
    const int V = dm.d.V, np = dm.d.num_parts;
    const AvtFrameCtl& ctl = fb.ctl[f];
    const int* po = fb.part_off + (size_t)f * (np + 1);
    const int nvalid = po[np];  // bucketed points with a valid label
    const int s0 = blockIdx.x * NN_QPB;
    if (s0 >= nvalid) return;
    const size_t base = (size_t)f * fb.max_points;
    const int sub = t % LANES;
    const int s = s0 + t / LANES;
    const bool active = s < nvalid;

    __shared__ double c_x[NN_TILE], c_y[NN_TILE], c_z[NN_TILE];
    __shared__ int c_id[NN_TILE], s_pre[NN_TILE + 4], s_wtot[4];
    __shared__ int s_qlo, s_qhi;
    if (t == 0) {
        int lo = 0, hi = np - 1;
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (po[mid] <= s0) lo = mid; else hi = mid - 1; }
        s_qlo = lo;
        const int last = min(s0 + NN_QPB - 1, nvalid - 1);
        lo = 0; hi = np - 1;
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (po[mid] <= last) lo = mid; else hi = mid - 1; }
        s_qhi = lo;
    }
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    if (active) { a0 = fb.dx[base + s]; a1 = fb.dy[base + s]; a2 = fb.dz[base + s]; }
    __syncthreads();
    const int qlo = s_qlo, qhi = s_qhi;
    int q = qlo;
    if (active) while (q < qhi && po[q + 1] <= s) ++q;
    const int my_b = dm.part_start[q], my_e = dm.part_start[q + 1];        // my part's positions (visible or not)
    const int ub = dm.part_start[qlo], ue = dm.part_start[qhi + 1];
    const double* pcx = fb.pcx + (size_t)f * V;
    const double* pcy = fb.pcy + (size_t)f * V;
    const double* pcz = fb.pcz + (size_t)f * V;
    const unsigned char* vs = fb.vis_sorted + (size_t)f * V;
    double best = 1.7976931348623157e308;  // numeric_limits<double>::max(), KNNResultSet::init
    constexpr int NN_GROUP = 8;
    int bkey = 0x7fffffff, bv = -1;         // winner so far: its rank in scan order (tile base + compacted index) and its vertex
    auto dist2 = [&](double cx, double cy, double cz) {
        const double d0 = a0 - cx, d1 = a1 - cy, d2 = a2 - cz;
        double r = d0 * d0;                 // (0 + d0*d0) == d0*d0 exactly
        r = r + d1 * d1;
        r = r + d2 * d2;
        return r;
    };
    for (int tb = ub; tb < ue; tb += NN_TILE) {
        const int tn = min(NN_TILE, ue - tb);
        // ---- compaction of positions tb .. tb + tn - 1 ----
        double px[4], py[4], pz[4];
        int pid[4];
        bool keep[4];
        int n = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = 4 * t + i, pos = tb + min(e, tn - 1);
            px[i] = pcx[pos]; py[i] = pcy[pos]; pz[i] = pcz[pos]; pid[i] = dm.part_vertices[pos];
            keep[i] = e < tn && vs[pos] != 0;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) n += keep[i] ? 1 : 0;
        int incl = n;
#pragma unroll
        for (int sft = 1; sft < 64; sft <<= 1) { const int v = __shfl_up(incl, sft, 64); if (lane_id() >= sft) incl += v; }
        __syncthreads();                    // the previous tile has been scanned by everybody
        if (lane_id() == 63) s_wtot[wave_id()] = incl;
        __syncthreads();
        int off = incl - n;
        for (int w = 0; w < wave_id(); ++w) off += s_wtot[w];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = 4 * t + i;
            if (e <= tn) s_pre[e] = off;
            if (keep[i]) { c_x[off] = px[i]; c_y[off] = py[i]; c_z[off] = pz[i]; c_id[off] = pid[i]; ++off; }
        }
        if (t == 255) s_pre[NN_TILE] = off;              // (tn == NN_TILE: the entry behind the last position)
        __syncthreads();
        // ---- scan of my part's visible points in this tile ----
        if (active) {
            const int b = s_pre[min(max(my_b - tb, 0), tn)], e = s_pre[min(max(my_e - tb, 0), tn)];
            int gpos = -1;
            int c = b + sub;
            for (; c + (NN_GROUP - 1) * LANES < e; c += NN_GROUP * LANES) {
                double r[NN_GROUP];
#pragma unroll
                for (int u = 0; u < NN_GROUP; ++u) r[u] = dist2(c_x[c + u * LANES], c_y[c + u * LANES], c_z[c + u * LANES]);
#pragma unroll
                for (int w = 1; w < NN_GROUP; w <<= 1)
#pragma unroll
                    for (int u = 0; u + w < NN_GROUP; u += 2 * w) r[u] = __builtin_fmin(r[u], r[u + w]);
                gpos = (r[0] < best) ? c : gpos;
                best = __builtin_fmin(best, r[0]);
            }
            for (; c < e; c += LANES) {     // the tail of the range: groups of one
                const double r = dist2(c_x[c], c_y[c], c_z[c]);
                gpos = (r < best) ? c : gpos;
                best = __builtin_fmin(best, r);
            }
            if (gpos >= 0) {                // this tile improved the minimum: the first member of the group whose distance IS the minimum
                double r[NN_GROUP];
#pragma unroll
                for (int u = 0; u < NN_GROUP; ++u) {
                    const int pos = min(gpos + u * LANES, e - 1);
                    r[u] = dist2(c_x[pos], c_y[pos], c_z[pos]);
                }
                int bi = gpos;
#pragma unroll
                for (int u = NN_GROUP - 1; u >= 0; --u)
                    if (gpos + u * LANES < e && r[u] == best) bi = gpos + u * LANES;
                bkey = tb + bi;
                bv = c_id[bi];
            }
        }
    }
    // combine the sub-scans: smallest distance, ties to the earliest candidate in scan order - exactly the winner of one
    // ascending scan with strict '<'
#pragma unroll
    for (int m = 1; m < LANES; m <<= 1) {
        const double ob = __shfl_xor(best, m, 64);
        const int ok = __shfl_xor(bkey, m, 64), ov = __shfl_xor(bv, m, 64);
        if (ob < best || (ob == best && ok < bkey)) { best = ob; bkey = ok; bv = ov; }
    }
    nn_record<LANES>(fb, ctl, f, V, base, s, active, sub, bv, a0, a1, a2);
}

// max over the wave of NON-NEGATIVE doubles, identical in every lane's return value: row_shr scans inside the rows of 16 lanes
// (lanes shifted in from outside a row read 0.0), then the four row maxima through v_readlane - no LDS permutes
template <int SH>
__device__ __forceinline__ double nn_row_shr0(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x110 + SH, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x110 + SH, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double nn_wave_max_nonneg(double v) {
    v = __builtin_fmax(v, nn_row_shr0<1>(v));
    v = __builtin_fmax(v, nn_row_shr0<2>(v));
    v = __builtin_fmax(v, nn_row_shr0<4>(v));
    v = __builtin_fmax(v, nn_row_shr0<8>(v));         // lane 15 of every row: the row's maximum
    const int lo = __double2loint(v), hi = __double2hiint(v);
    double m = __hiloint2double(__builtin_amdgcn_readlane(hi, 15), __builtin_amdgcn_readlane(lo, 15));
    m = __builtin_fmax(m, __hiloint2double(__builtin_amdgcn_readlane(hi, 31), __builtin_amdgcn_readlane(lo, 31)));
    m = __builtin_fmax(m, __hiloint2double(__builtin_amdgcn_readlane(hi, 47), __builtin_amdgcn_readlane(lo, 47)));
    m = __builtin_fmax(m, __hiloint2double(__builtin_amdgcn_readlane(hi, 63), __builtin_amdgcn_readlane(lo, 63)));
    return m;
}

// min (MAX = false) / max over the wave, identical in every lane's return value, without LDS permutes: row_shr scans that keep a lane's
// own value where the shift leaves its row (update_dpp with old = the value itself, bound_ctrl off), then the four rows through v_readlane
template <bool MAX, int SH>
__device__ __forceinline__ double nn_row_step(double v) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), 0x110 + SH, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), 0x110 + SH, 0xf, 0xf, false);
    const double o = __hiloint2double(hi, lo);
    return MAX ? __builtin_fmax(v, o) : __builtin_fmin(v, o);
}
template <bool MAX>
__device__ __forceinline__ double nn_wave_reduce(double v) {
    v = nn_row_step<MAX, 1>(v); v = nn_row_step<MAX, 2>(v); v = nn_row_step<MAX, 4>(v); v = nn_row_step<MAX, 8>(v);   // lane 15 of every row
    const int lo = __double2loint(v), hi = __double2hiint(v);
    double m = __hiloint2double(__builtin_amdgcn_readlane(hi, 15), __builtin_amdgcn_readlane(lo, 15));
#pragma unroll
    for (int l = 31; l < 64; l += 16) {
        const double o = __hiloint2double(__builtin_amdgcn_readlane(hi, l), __builtin_amdgcn_readlane(lo, l));
        m = MAX ? __builtin_fmax(m, o) : __builtin_fmin(m, o);
    }
    return m;
}

// -------------------------------------------------------------------------------------------------
// k_nn_part: the throughput shape (frame batches).  A workgroup owns up to 256 NN_PART_NQ consecutive bucketed data points of ONE part
// (NN_PART_NQ per lane), so every lane scans the same candidates: the part's visible model points are read with SCALAR loads
// (constant address space: k_compact wrote them in an earlier kernel) straight into the VALU's scalar operand - no LDS
// tile, no barrier, no per-lane range arithmetic; what remains per candidate is 8 flops and one v_min_f64.  Same distance
// expression, same strict '<' in ascending candidate order, hence the same index as k_nn and nanoflann.
// grid: (upper bound of sum_q ceil(points of part q / 256), frames): lane l of every wave looks at part l, a wave prefix sum
// of the parts' workgroup counts maps blockIdx.x to (part, chunk).
// -------------------------------------------------------------------------------------------------
typedef const __attribute__((address_space(4))) double* nn_cptr;

__global__ __launch_bounds__(256) void k_nn_part(DeviceModel dm, FrameBuffers fb) {
    // NQ queries per lane: every candidate a wave fetches (scalar loads: one request per 32 bytes, a miss in the scalar cache for every other
    // one - the candidates are streamed, 28 % of the requests missed with one query per lane and the kernel ran at the rate of those misses,
    // neither shorter instruction streams nor requesting a group ahead moved it) is used for NQ distances instead of one.
    constexpr int NQ = NN_PART_NQ, CH = 256 * NQ;      // queries per lane / per workgroup: wave w owns CH / 4 consecutive ones, lane l the (64 u + l)-th of them
    int bx, fy;
    xcd_frame_block(fb, bx, fy);      // the chunks of a part all scan the part's candidates, a frame's workgroups add to the same counts and sums
    const int f = fy + fb.f0, t = threadIdx.x, lane = t & 63;
    const int V = dm.d.V, np = dm.d.num_parts;
    const AvtFrameCtl& ctl = fb.ctl[f];
    const int* po = fb.part_off + (size_t)f * (np + 1);
    const int po_l = (lane <= np) ? po[min(lane, np)] : 0, po_n = (lane < np) ? po[lane + 1] : 0;
    const int nbq = (lane < np) ? (po_n - po_l + CH - 1) / CH : 0;
    int incl = nbq;
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) { const int v = __shfl_up(incl, sft, 64); if (lane >= sft) incl += v; }
    const int excl = incl - nbq, blk = bx;
    const unsigned long long hit = __ballot(lane < np && blk >= excl && blk < incl);
    if (hit == 0ull) return;                                  // past the last chunk of the last part (grid is an upper bound)
    const int q = __ffsll((long long)hit) - 1;
    const int s_beg = __shfl(po_l, q, 64) + (blk - __shfl(excl, q, 64)) * CH, s_end = __shfl(po_n, q, 64);
    const size_t base = (size_t)f * fb.max_points;
    int s[NQ];
    bool active[NQ], any_active = false;
    double a0[NQ], a1[NQ], a2[NQ];
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        s[u] = s_beg + (t >> 6) * (64 * NQ) + 64 * u + lane;
        active[u] = s[u] < s_end;
        any_active = any_active || active[u];
        a0[u] = 0.0; a1[u] = 0.0; a2[u] = 0.0;
        if (active[u]) { a0[u] = fb.dx[base + s[u]]; a1[u] = fb.dy[base + s[u]]; a2[u] = fb.dz[base + s[u]]; }
    }
    const int pb = __builtin_amdgcn_readfirstlane(dm.part_start[q]);
    const int vc = __builtin_amdgcn_readfirstlane(fb.vcount[(size_t)f * np + q]);
    const bool sorted = (vc & NN_SORTED_FLAG) != 0;                    // k_compact sorted them by (y, vertex id): slab scan below
#ifdef AVT_NN_NO_SCAN          // (timing experiment: no candidate is looked at - what remains is the kernel's per-wave fixed work)
    const int pe = pb;
#else
    const int pe = pb + (vc & ~NN_SORTED_FLAG);                         // visible candidates of the part
#endif
    const nn_cptr cx = (nn_cptr)(uintptr_t)(fb.vcx + (size_t)f * V), cy = (nn_cptr)(uintptr_t)(fb.vcy + (size_t)f * V),
                  cz = (nn_cptr)(uintptr_t)(fb.vcz + (size_t)f * V);
    auto dist2 = [&](int u, double px, double py, double pz) {
        const double d0 = a0[u] - px, d1 = a1[u] - py, d2 = a2[u] - pz;
        double r = d0 * d0;
        r = r + d1 * d1;
        r = r + d2 * d2;
        return r;
    };
    constexpr int NN_GROUP = NN_PART_GROUP;
    double best[NQ];
    int gpos[NQ];
    bool tie[NQ];                 // (slab scan only) two candidates at exactly the minimum distance: settled by vertex id below
#pragma unroll
    for (int u = 0; u < NQ; ++u) { best[u] = 1.7976931348623157e308; gpos[u] = -1; tie[u] = false; }
    // the groups of NN_GROUP consecutive candidates of [lo, hi) (lo a multiple of NN_GROUP past pb), then the members of the last, partial one
    auto scan_groups = [&](int lo, int hi, bool watch_ties) {
        int c = lo;
        for (; c + NN_GROUP <= hi; c += NN_GROUP) {
            double px[NN_GROUP], py[NN_GROUP], pz[NN_GROUP];
#pragma unroll
            for (int g = 0; g < NN_GROUP; ++g) { px[g] = cx[c + g]; py[g] = cy[c + g]; pz[g] = cz[c + g]; }
#pragma unroll
            for (int u = 0; u < NQ; ++u) {
                double r[NN_GROUP];
#pragma unroll
                for (int g = 0; g < NN_GROUP; ++g) r[g] = dist2(u, px[g], py[g], pz[g]);
#pragma unroll
                for (int w = 1; w < NN_GROUP; w <<= 1)
#pragma unroll
                    for (int g = 0; g + w < NN_GROUP; g += 2 * w) r[g] = __builtin_fmin(r[g], r[g + w]);
                if (watch_ties) tie[u] = tie[u] || (r[0] == best[u]);
                gpos[u] = (r[0] < best[u]) ? c : gpos[u];
                best[u] = __builtin_fmin(best[u], r[0]);
            }
        }
        for (; c < hi; ++c) {
            const double px = cx[c], py = cy[c], pz = cz[c];
#pragma unroll
            for (int u = 0; u < NQ; ++u) {
                const double r = dist2(u, px, py, pz);
                if (watch_ties) tie[u] = tie[u] || (r == best[u]);
                gpos[u] = (r < best[u]) ? c : gpos[u];
                best[u] = __builtin_fmin(best[u], r);
            }
        }
    };
    if (!sorted) {
        scan_groups(pb, pe, false);      // ascending vertex order: strict '<' alone implements the tie rule
    } else {
        // ---- slab scan.  k_compact sorted this part's visible candidates by (y, vertex id).  A wave's 64 NQ queries are consecutive
        // pixels of one part - a few image rows, i.e. a thin slab in y - so the scan starts at the candidate nearest to the
        // slab and walks outwards on both sides, NN_SLAB_CHUNK candidates a side per round, until on each side the next candidate is further
        // from the slab IN Y ALONE than the worst current best distance of the wave: every candidate beyond it has
        // (y_c - y_q)^2 >= (y_edge - y_slab)^2 > max_q best_q, so its rounded distance exceeds every lane's best (margin 1e-12 >>
        // the 5 ulp the rounding of the two expressions can differ by) and it can neither win nor tie.  Exact: same distance
        // expression, same winner as the full ascending scan - ties (equal distances, measure zero) are detected and settled by
        // vertex id in a full rescan.  (tools/nn_slab_sim.py, tools/nn_count_probe.py: the share of the candidates evaluated.)
        double ymin = 1.7976931348623157e308, ymax = -1.7976931348623157e308;
#pragma unroll
        for (int u = 0; u < NQ; ++u) if (active[u]) { ymin = __builtin_fmin(ymin, a1[u]); ymax = __builtin_fmax(ymax, a1[u]); }
        const double ylo = nn_wave_reduce<false>(ymin), yhi = nn_wave_reduce<true>(ymax);
        const int n = pe - pb;
        bool rdone = true, ldone = true;
        int R = pb, L = pb;
        if (__ballot(any_active) != 0ull && n > 0) {
            const double ymid = 0.5 * (ylo + yhi);
            const int step = (n + 63) >> 6;                                  // 64 samples of the sorted y's locate the slab
            const int si = lane * step;
            const double ys = fb.vcy[(size_t)f * V + pb + min(si, n - 1)];
            const int below = __popcll(__ballot(si < n && ys < ymid));
            const int st = pb + ((min(max(below - 1, 0) * step, n - 1)) & ~(NN_GROUP - 1));   // groups stay aligned to pb + NN_GROUP k
            R = L = __builtin_amdgcn_readfirstlane(st);
            rdone = R >= pe; ldone = L <= pb;
        }
#ifdef AVT_NN_COUNT
        int cnt_eval = 0, cnt_rounds = 0;
#endif
        while (!(rdone && ldone)) {                                          // wave-uniform
            // the candidates just outside this round's chunks decide whether there is a next round: requested with the chunks
            const int e = min(R + NN_SLAB_CHUNK, pe), b = max(L - NN_SLAB_CHUNK, pb);
            const double edge_r = cy[min(e, pe - 1)], edge_l = cy[max(b - 1, pb)];
#ifdef AVT_NN_COUNT
            ++cnt_rounds; cnt_eval += (rdone ? 0 : e - R) + (ldone ? 0 : L - b);
#endif
            if (!rdone) { scan_groups(R, e, true); R = e; }
            if (!ldone) { scan_groups(b, L, true); L = b; }
            double worst = 0.0;
#pragma unroll
            for (int u = 0; u < NQ; ++u) worst = __builtin_fmax(worst, active[u] ? best[u] : 0.0);
            const double dmax = nn_wave_max_nonneg(worst);
            const double bound = dmax * (1.0 + 1e-12);
            if (!rdone) {
                const double gap = edge_r - yhi;
                rdone = R >= pe || __builtin_amdgcn_readfirstlane((int)(gap > 0.0 && gap * gap > bound)) != 0;
            }
            if (!ldone) {
                const double gap = ylo - edge_l;
                ldone = L <= pb || __builtin_amdgcn_readfirstlane((int)(gap > 0.0 && gap * gap > bound)) != 0;
            }
        }
#ifdef AVT_NN_COUNT   // tools/nn_count_probe.py (make libavatar_hip_nn_count.so): waves, candidates evaluated / available, rounds, slab width
        if (lane == 0 && __ballot(any_active) != 0ull) {
            double* tr = fb.trace + (size_t)f * 64;
            atomicAdd(tr + 58, 1.0); atomicAdd(tr + 60, (double)cnt_eval); atomicAdd(tr + 61, (double)n); atomicAdd(tr + 62, (double)cnt_rounds); atomicAdd(tr + 63, yhi - ylo);
        }
#endif
    }
    int bi[NQ];
    const double* pcx = fb.vcx + (size_t)f * V;
    const double* pcy = fb.vcy + (size_t)f * V;
    const double* pcz = fb.vcz + (size_t)f * V;
    bool any_tie = false;
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        bi[u] = 0x7fffffff;
        if (gpos[u] >= 0) {      // the first member of the winning group whose distance is the minimum
            double r[NN_GROUP];
#pragma unroll
            for (int g = 0; g < NN_GROUP; ++g) {
                const int pos = min(gpos[u] + g, pe - 1);
                r[g] = dist2(u, pcx[pos], pcy[pos], pcz[pos]);
            }
            int hits = 0;
#pragma unroll
            for (int g = NN_GROUP - 1; g >= 0; --g)
                if (gpos[u] + g < pe && r[g] == best[u]) { bi[u] = gpos[u] + g; ++hits; }
            if (sorted && hits > 1) tie[u] = true;
        }
        any_tie = any_tie || tie[u];
    }
#ifdef AVT_NN_COUNT
    if (sorted && __ballot(any_tie) != 0ull && lane == 0) atomicAdd(fb.trace + (size_t)f * 64 + 59, 1.0);
#endif
    if (sorted && __ballot(any_tie) != 0ull) {      // exact ties: the candidate with the smallest vertex id among those at the minimum distance
        const int* ids = fb.vcid + (size_t)f * V;
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            double tb = 1.7976931348623157e308;
            int tid = 0x7fffffff, tpos = 0x7fffffff;
            for (int c = pb; c < pe; ++c) {
                const double r = dist2(u, cx[c], cy[c], cz[c]);
                const int id = ids[c];
                if (r < tb || (r == tb && id < tid)) { tb = r; tid = id; tpos = c; }
            }
            if (tie[u]) bi[u] = tpos;
        }
    }
    int mv[NQ];
#pragma unroll
    for (int u = 0; u < NQ; ++u) mv[u] = (active[u] && bi[u] != 0x7fffffff) ? fb.vcid[(size_t)f * V + bi[u]] : -1;
    if (pe - pb > NN_ACC_CAP) {        // a part with more candidates than the accumulators below hold: per-match atomics, merged in-wave
#pragma unroll
        for (int u = 0; u < NQ; ++u) nn_record<1>(fb, ctl, f, V, base, s[u], active[u], 0, mv[u], a0[u], a1[u], a2[u]);
        return;
    }
    // ---- bookkeeping.  The workgroup's queries belong to ONE part, so their matches are among that part's candidates: counts
    // and fixed-point sums are accumulated per candidate POSITION in LDS (integer atomics: order-independent) and every matched
    // candidate is flushed to the per-vertex arrays once per workgroup - a vertex's matches are neighbouring pixels, mostly of one
    // workgroup, so the global atomics fall from one per run of equal matches in a wave (~12 k x 4 per frame) to ~1.3 per matched
    // vertex (~2 k x 4), and with them the count / sum lines written back more than once.
    __shared__ int s_acnt[NN_ACC_CAP];
    __shared__ unsigned long long s_asum[3][NN_ACC_CAP];
    const int nacc = pe - pb;
    for (int i = t; i < nacc; i += 256) { s_acnt[i] = 0; s_asum[0][i] = 0ull; s_asum[1][i] = 0ull; s_asum[2][i] = 0ull; }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        if (!active[u]) continue;
        fb.corr_sorted[base + s[u]] = mv[u];
        fb.corr[base + fb.dorig[base + s[u]]] = mv[u];
        if (mv[u] >= 0) {
            const int i = bi[u] - pb;
            atomicAdd(&s_acnt[i], 1);
            atomicAdd(&s_asum[0][i], (unsigned long long)rint_to_ll((a0[u] - ctl.centre[0]) * AVT_FIX_SCALE));
            atomicAdd(&s_asum[1][i], (unsigned long long)rint_to_ll((a1[u] - ctl.centre[1]) * AVT_FIX_SCALE));
            atomicAdd(&s_asum[2][i], (unsigned long long)rint_to_ll((a2[u] - ctl.centre[2]) * AVT_FIX_SCALE));
        }
    }
    __syncthreads();
    unsigned long long* fs = (unsigned long long*)(fb.fsum + (size_t)f * 3 * V);
    for (int i = t; i < nacc; i += 256) {
        const int c = s_acnt[i];
        if (c == 0) continue;
        const int v = fb.vcid[(size_t)f * V + pb + i];
        atomicAdd(fb.cnt + (size_t)f * V + v, c);
        atomicAdd(fs + v, s_asum[0][i]);
        atomicAdd(fs + (size_t)V + v, s_asum[1][i]);
        atomicAdd(fs + 2 * (size_t)V + v, s_asum[2][i]);
    }
}

// Visible model points of every part, compacted in ascending vertex order inside the part's segment of the
// part-sorted arrays (exactly the per-part clouds findNN builds at AvatarOptimizer.cpp:860-878).  grid (parts, frames).
// Workgroups past the parts (first ICP iteration of a frame batch): the scatter pass of the data bucketing (avt_bucket.h).
// from_cloud: the coordinates come from the cloud through the vertex id (frame batches inside optimize(): k_lbs then skips
// the part-sorted copy, three scattered stores per vertex) instead of from pcx/pcy/pcz.
__global__ __launch_bounds__(256) void k_compact(DeviceModel dm, FrameBuffers fb, int from_cloud, int sort_y) {
    int bx, fy;
    xcd_frame_block(fb, bx, fy);      // what it writes is what the frame's k_nn_part workgroups read
    const int f = fy + fb.f0, q = bx, t = threadIdx.x, V = dm.d.V, np = dm.d.num_parts;
    if (q >= np) { bucket_scatter_block<true>(dm, fb, f, q - np); return; }
    const int b = dm.part_start[q], e = dm.part_start[q + 1];
    __shared__ int s_wcnt[4];
    __shared__ int s_run;
    struct __attribute__((aligned(16))) Ent { double y; int vid; int slot; };
    __shared__ Ent s_ent[NN_SORT_CAP];            // sort_y: the part's visible candidates (y, vertex id, slot of its x and z below)
    // (x and z are gathered a second time when the sorted candidates are written: keeping them here as well cost 16 KB of LDS, i.e. the
    // fourth to sixth workgroup of a CU - the kernel is dependent round trips, so what it has resident is what it runs at)
    if (t == 0) s_run = 0;
    __syncthreads();
    const unsigned char* vis = fb.visible + (size_t)f * V;
    // pass 0 writes the compacted candidates in ascending vertex order to LDS (sort_y) or straight to memory; if a part turns out
    // to be larger than the sort's capacity, pass 1 writes it to memory unsorted
    for (int pass = sort_y ? 0 : 1; pass < 2; ++pass) {
        if (pass == 1 && sort_y) {
            if (s_run <= NN_SORT_CAP) break;
            __syncthreads();
            if (t == 0) s_run = 0;
            __syncthreads();
        }
        for (int c0 = b; c0 < e; c0 += 256) {
            const int pos = c0 + t;
            int v = -1;
            bool keep = false;
            if (pos < e) { v = dm.part_vertices[pos]; keep = vis[v] != 0; }
            const unsigned long long bal = __ballot(keep);
            const int rank = __popcll(bal & ((1ull << lane_id()) - 1ull));
            if (lane_id() == 0) s_wcnt[wave_id()] = __popcll(bal);
            __syncthreads();
            int off = s_run;
            for (int w = 0; w < wave_id(); ++w) off += s_wcnt[w];
            if (keep) {
                if (pass == 0) {
                    // (ids only: the coordinates are gathered after the loop, all at once - inside it every iteration would wait for
                    // its own dependent gather in front of the barrier)
                    if (off + rank < NN_SORT_CAP) s_ent[off + rank] = Ent{0.0, v, pos};
                } else {
                    const size_t o = (size_t)f * V + b + off + rank;
                    if (from_cloud) {
                        const double* cl = fb.cloud + ((size_t)f * V + v) * 3;
                        fb.vcx[o] = cl[0]; fb.vcy[o] = cl[1]; fb.vcz[o] = cl[2];
                    } else {
                        fb.vcx[o] = fb.pcx[(size_t)f * V + pos];
                        fb.vcy[o] = fb.pcy[(size_t)f * V + pos];
                        fb.vcz[o] = fb.pcz[(size_t)f * V + pos];
                    }
                    fb.vcid[o] = v;
                }
            }
            __syncthreads();
            if (t == 0) s_run += s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3];
            __syncthreads();
        }
    }
    const int n = s_run;
    const bool sorted = sort_y && n <= NN_SORT_CAP;
    if (sorted) {
        for (int i = t; i < n; i += 256) {           // coordinates of the compacted candidates: independent gathers, one exposed latency
            const int v = s_ent[i].vid, pos = s_ent[i].slot;
            double x, y, z;
            if (from_cloud) { const double* cl = fb.cloud + ((size_t)f * V + v) * 3; x = cl[0]; y = cl[1]; z = cl[2]; }
            else { x = fb.pcx[(size_t)f * V + pos]; y = fb.pcy[(size_t)f * V + pos]; z = fb.pcz[(size_t)f * V + pos]; }
            s_ent[i] = Ent{y, v, pos};
            (void)x; (void)z;
        }
        // bitonic sort by (y, vertex id) in LDS, padded to a power of two with +inf keys (a rank sort - every element counting
        // the elements in front of it - is simpler but quadratic: 62 us against 8 for the 32-frame launch with parts of 900)
        int P2 = 64;
        while (P2 < n) P2 <<= 1;
        for (int i = n + t; i < P2; i += 256) s_ent[i] = Ent{1.7976931348623157e308, 0x7fffffff, 0};
        // pair pi of a stage with distance j: elements i = pi with a zero bit inserted at bit log2(j), and i | j.  With j <= 64 both
        // lie in the 128-element block pi >> 6, and thread t only ever takes pair indices t + 256 m, i.e. blocks of its own wave: runs
        // of stages with j <= 64 need no workgroup barrier (LDS operations of one wave execute in order), only the stages with
        // j >= 128 and their neighbours do - five barriers instead of forty-five for 512 elements.
        bool prev_big = true;
        for (int k = 2; k <= P2; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                const bool big = j >= 128;
                if (big || prev_big) __syncthreads();
                else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
                prev_big = big;
                for (int pi = t; pi < (P2 >> 1); pi += 256) {
                    const int i = ((pi & ~(j - 1)) << 1) | (pi & (j - 1)), p = i | j;
                    const Ent a = s_ent[i], c = s_ent[p];
                    const bool a_after_c = c.y < a.y || (c.y == a.y && c.vid < a.vid);
                    if (a_after_c == ((i & k) == 0)) { s_ent[i] = c; s_ent[p] = a; }
                }
            }
        __syncthreads();
        for (int i = t; i < n; i += 256) {
            const Ent me = s_ent[i];
            const size_t o = (size_t)f * V + b + i;
            double x, z;
            if (from_cloud) { const double* cl = fb.cloud + ((size_t)f * V + me.vid) * 3; x = cl[0]; z = cl[2]; }
            else { x = fb.pcx[(size_t)f * V + me.slot]; z = fb.pcz[(size_t)f * V + me.slot]; }
            fb.vcx[o] = x; fb.vcy[o] = me.y; fb.vcz[o] = z;
            fb.vcid[o] = me.vid;
        }
    }
    if (t == 0) fb.vcount[(size_t)f * np + q] = n | (sorted ? NN_SORTED_FLAG : 0);
}

// few queries: the latency shape of the nearest-neighbour stage (4 lanes per query); many: one lane per query, one part per workgroup
bool avt_nn_few(const avt_ctx* c, int nframes) {
    return !c->tun.nn_force_part && (long long)nframes * c->launch_maxN <= 400000;      // (nn_force_part: tests run the throughput shape on small inputs too)
}

void launch_nn(avt_ctx* c, int nframes) {
    const int V = c->dm.d.V;
    // cnt and fsum are adjacent: one memset clears both for the frames in use
    if (!c->lbs_cleared) {   // stand-alone avt_nn(): no preceding k_lbs cleared the bookkeeping
        (void)hipMemsetAsync(c->fb.cnt + (size_t)c->fb.f0 * V, 0, (size_t)nframes * V * sizeof(int), c->cur_stream);
        (void)hipMemsetAsync(c->fb.fsum + (size_t)c->fb.f0 * 3 * V, 0, (size_t)nframes * 3 * V * sizeof(long long), c->cur_stream);
    }
    const int maxN = c->launch_maxN;
    if (maxN <= 0) return;
    // few queries: 4 lanes per query (more workgroups, shorter scans); many: one lane per query, one part per workgroup
    const bool few = avt_nn_few(c, nframes);
    if (few && c->lbs_cleared) {   // inside optimize() (k_lbs / k_visibility keep the part-sorted visibility flags): compaction fused into the scan
        hipLaunchKernelGGL(HIP_KERNEL_NAME(k_nn_vis<4>), dim3((maxN + 63) / 64, nframes), dim3(256), 0, c->cur_stream, c->dm, c->fb);
        return;
    }
    const int nscat = c->scatter_in_compact ? std::max(1, (maxN + BUCKET_TILE - 1) / BUCKET_TILE) : 0;
    c->scatter_in_compact = false;
    // (the throughput scan walks y-sorted candidates outwards from the wave's slab of queries; the latency shape keeps ascending vertex order)
    const bool no_slab = !c->tun.nn_slab;
    hipLaunchKernelGGL(k_compact, dim3(c->dm.d.num_parts + nscat, nframes), dim3(256), 0, c->cur_stream, c->dm, c->fb, c->nn_from_cloud ? 1 : 0,
                       (!few && !no_slab) ? 1 : 0);
    if (few) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_nn<4>), dim3((maxN + 63) / 64, nframes), dim3(256), 0, c->cur_stream, c->dm, c->fb);
    else hipLaunchKernelGGL(k_nn_part, dim3((maxN + 256 * NN_PART_NQ - 1) / (256 * NN_PART_NQ) + c->dm.d.num_parts, nframes), dim3(256), 0, c->cur_stream, c->dm, c->fb);      // (an upper bound of the parts' chunks)
}
