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
// the LDS reads are broadcasts.  Work is fp64-VALU bound: 8 flops + compare/select per candidate.
//
// The kernel also accumulates, per matched model vertex, the correspondence count and the fixed-point
// (2^40, frame-centred) sum of its data points with integer atomics: order-independent, hence bit-wise
// reproducible run to run (k_finalize turns them into sqrt(count) and the mean data point).
#include "avt_device.h"

#define NN_TILE 1024

#define NN_QPB 64   // queries per workgroup: 4 lanes cooperate on one query

__global__ __launch_bounds__(256) void k_nn(DeviceModel dm, FrameBuffers fb) {
    const int f = blockIdx.y, t = threadIdx.x;
    const int V = dm.d.V, np = dm.d.num_parts;
    const AvtFrameCtl& ctl = fb.ctl[f];
    const int* po = fb.part_off + (size_t)f * (np + 1);
    const int nvalid = po[np];  // bucketed points with a valid label
    const int s0 = blockIdx.x * NN_QPB;
    if (s0 >= nvalid) return;
    const size_t base = (size_t)f * fb.max_points;
    const int sub = t & 3;
    const int s = s0 + (t >> 2);
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
    const int my_b = dm.part_start[q], my_e = dm.part_start[q + 1];
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    if (active) { a0 = fb.dx[base + s]; a1 = fb.dy[base + s]; a2 = fb.dz[base + s]; }

    const int cb = dm.part_start[qlo], ce = dm.part_start[qhi + 1];
    const double* pcx = fb.pcx + (size_t)f * V;
    const double* pcy = fb.pcy + (size_t)f * V;
    const double* pcz = fb.pcz + (size_t)f * V;
    const unsigned char* vis = fb.visible + (size_t)f * V;
    double best = 1.7976931348623157e308;  // numeric_limits<double>::max(), KNNResultSet::init
    int bi = 0x7fffffff;
    for (int tb = cb; tb < ce; tb += NN_TILE) {
        const int tn = min(NN_TILE, ce - tb);
        __syncthreads();
        for (int e = t; e < tn; e += 256) {
            const int pos = tb + e;
            const bool vz = vis[dm.part_vertices[pos]] != 0;
            c_x[e] = vz ? pcx[pos] : AVT_INF;   // invisible candidates can never win: inf < best is false
            c_y[e] = pcy[pos];
            c_z[e] = pcz[pos];
        }
        __syncthreads();
        if (active) {
            const int b = max(my_b, tb) - tb, e = min(my_e, tb + tn) - tb;
            // lane `sub` of the query's 4-lane group scans candidates b+sub, b+sub+4, ... in ascending order
            for (int c = b + sub; c < e; c += 4) {
                const double d0 = a0 - c_x[c];
                const double d1 = a1 - c_y[c];
                const double d2 = a2 - c_z[c];
                double r = d0 * d0;          // (0 + d0*d0) == d0*d0 exactly
                r = r + d1 * d1;
                r = r + d2 * d2;
                if (r < best) { best = r; bi = tb + c; }
            }
        }
    }
    // combine the 4 sub-scans: smallest distance, ties to the lowest candidate position — exactly the winner of
    // one ascending scan with strict '<'
#pragma unroll
    for (int m = 1; m <= 2; m <<= 1) {
        const double ob = __shfl_xor(best, m, 64);
        const int oi = __shfl_xor(bi, m, 64);
        if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (!active || sub != 0) return;
    const int m = bi != 0x7fffffff ? dm.part_vertices[bi] : -1;
    fb.corr_sorted[base + s] = m;
    fb.corr[base + fb.dorig[base + s]] = m;
    if (m >= 0) {
        atomicAdd(fb.cnt + (size_t)f * V + m, 1);
        unsigned long long* fs = (unsigned long long*)(fb.fsum + (size_t)f * 3 * V);
        const long long q0 = __double2ll_rn((a0 - ctl.centre[0]) * AVT_FIX_SCALE);
        const long long q1 = __double2ll_rn((a1 - ctl.centre[1]) * AVT_FIX_SCALE);
        const long long q2 = __double2ll_rn((a2 - ctl.centre[2]) * AVT_FIX_SCALE);
        atomicAdd(fs + m, (unsigned long long)q0);
        atomicAdd(fs + (size_t)V + m, (unsigned long long)q1);
        atomicAdd(fs + 2 * (size_t)V + m, (unsigned long long)q2);
    }
}

void launch_nn(avt_ctx* c, int nframes) {
    const int V = c->dm.d.V;
    // cnt and fsum are adjacent: one memset clears both for the frames in use
    hipMemsetAsync(c->fb.cnt, 0, (size_t)nframes * V * sizeof(int), c->stream);
    hipMemsetAsync(c->fb.fsum, 0, (size_t)nframes * 3 * V * sizeof(long long), c->stream);
    const int maxN = c->launch_maxN;
    const int nb = (maxN + NN_QPB - 1) / NN_QPB;
    if (nb > 0) hipLaunchKernelGGL(k_nn, dim3(nb, nframes), dim3(256), 0, c->stream, c->dm, c->fb);
}
