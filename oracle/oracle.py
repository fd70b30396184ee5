"""ctypes binding of oracle/liboracle.so (and, when present, oracle/_ref/libnanoflann_ref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Nothing under avatar_amd/ imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from avatar_amd.capi import ModelArrays, ModelDesc, Options, Stats, bptr, dptr, iptr

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")
_REF_SO = os.path.join(_HERE, "_ref", "libnanoflann_ref.so")


def build(force=False):
    src = os.path.join(_HERE, "avatar_oracle.cpp")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    if os.path.exists("/root/reference/include/nanoflann.hpp") and (force or not os.path.exists(_REF_SO)):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_model_create.restype = C.c_void_p
        L.orc_model_create.argtypes = [C.POINTER(ModelDesc)]
        L.orc_model_destroy.argtypes = [C.c_void_p]
        L.orc_model_num_ancestors.restype = C.c_int
        L.orc_pose_prior_residual.restype = C.c_int
        _lib = L
    return _lib


def have_reference_nn():
    return os.path.exists(_REF_SO)


def reference_nn(model_part, model_cloud, visible, data, labels, num_parts):
    """The reference's own nanoflann KD-tree search (build container only)."""
    L = C.CDLL(_REF_SO)
    V = len(model_part); N = len(labels)
    out = np.empty(N, np.int32); dist = np.empty(N, np.float64)
    mp = np.ascontiguousarray(model_part, np.int32)
    mc = np.ascontiguousarray(model_cloud, np.float64); vis = np.ascontiguousarray(visible, np.uint8)
    d = np.ascontiguousarray(data, np.float64); lb = np.ascontiguousarray(labels, np.int32)
    L.ref_find_nn_inverted(C.c_int(V), C.c_int(num_parts), iptr(mp), dptr(mc), bptr(vis), dptr(d), iptr(lb),
                           C.c_int(N), iptr(out), dptr(dist))
    return out, dist


class OracleModel:
    def __init__(self, smpl: dict, limit_one_joint_per_point=False):
        self.arrays = ModelArrays(smpl, limit_one_joint_per_point)
        self._desc = self.arrays.desc()
        self.h = C.c_void_p(lib().orc_model_create(C.byref(self._desc)))
        self.V, self.J, self.K, self.F, self.P = (self.arrays.V, self.arrays.J, self.arrays.K, self.arrays.F,
                                                  self.arrays.P)

    def __del__(self):
        try:
            lib().orc_model_destroy(self.h)
        except Exception:
            pass

    # --- derived model data
    def joint_regression(self):
        ijp = np.empty(3 * self.J); jsr = np.empty(3 * self.J * self.K)
        lib().orc_model_joint_regression(self.h, dptr(ijp), dptr(jsr))
        return ijp.reshape(self.J, 3), jsr.reshape(self.K, 3 * self.J).T  # (J,3), (3J,K)

    def main_joint(self):
        out = np.empty(self.V, np.int32)
        lib().orc_model_main_joint(self.h, iptr(out))
        return out

    def ancestors(self, point):
        n = lib().orc_model_num_ancestors(self.h, C.c_int(point))
        out = np.empty(n, np.int32)
        lib().orc_model_ancestors(self.h, C.c_int(point), iptr(out))
        return out

    # --- Avatar::update
    def update(self, w, p, R):
        """R: (J,3,3) rotation matrices. Returns cloud (V,3), jointPos (J,3), jointTrans (J,12 col-major 3x4)."""
        w = np.ascontiguousarray(w, np.float64); p = np.ascontiguousarray(p, np.float64)
        Rcm = np.ascontiguousarray(np.transpose(np.asarray(R, np.float64), (0, 2, 1))).reshape(-1)
        cloud = np.empty(3 * self.V); jp = np.empty(3 * self.J); jt = np.empty(12 * self.J)
        lib().orc_update(self.h, dptr(w), dptr(p), dptr(Rcm), dptr(cloud), dptr(jp), dptr(jt))
        return cloud.reshape(self.V, 3), jp.reshape(self.J, 3), jt.reshape(self.J, 12)

    def visibility(self, cloud, enable=True):
        cloud = np.ascontiguousarray(cloud, np.float64)
        vis = np.empty(self.V, np.uint8)
        lib().orc_visibility(self.h, dptr(cloud), C.c_int(int(enable)), bptr(vis))
        return vis

    def nn(self, part_map, num_parts, cloud, vis, data, labels):
        pm = np.ascontiguousarray(part_map, np.int32)
        cloud = np.ascontiguousarray(cloud, np.float64); vis = np.ascontiguousarray(vis, np.uint8)
        data = np.ascontiguousarray(data, np.float64); labels = np.ascontiguousarray(labels, np.int32)
        out = np.empty(len(labels), np.int32)
        lib().orc_nn(self.h, C.c_int(num_parts), iptr(pm), dptr(cloud), bptr(vis), dptr(data), iptr(labels),
                     C.c_int(len(labels)), iptr(out))
        return out

    # --- evaluation pieces
    def points(self, p, q, w):
        p, q, w = (np.ascontiguousarray(a, np.float64) for a in (p, q, w))
        cloud = np.empty(3 * self.V)
        lib().orc_points(self.h, dptr(p), dptr(q), dptr(w), dptr(cloud))
        return cloud.reshape(self.V, 3)

    def point_jacobian(self, p, q, w, point):
        p, q, w = (np.ascontiguousarray(a, np.float64) for a in (p, q, w))
        x = np.empty(3); Jd = np.empty(3 * self.P)
        lib().orc_point_jacobian(self.h, dptr(p), dptr(q), dptr(w), C.c_int(point), dptr(x), dptr(Jd))
        return x, Jd.reshape(3, self.P)

    def retract(self, p, q, w, delta):
        p, q, w, delta = (np.ascontiguousarray(a, np.float64) for a in (p, q, w, delta))
        p2 = np.empty(3); q2 = np.empty(4 * self.J); w2 = np.empty(self.K)
        lib().orc_retract(self.h, dptr(p), dptr(q), dptr(w), dptr(delta), dptr(p2), dptr(q2), dptr(w2))
        return p2, q2.reshape(self.J, 4), w2

    def pose_prior_residual(self, q):
        q = np.ascontiguousarray(q, np.float64)
        nd = self.arrays.ndims
        x = np.empty(nd); res = np.empty(nd + 1)
        comp = lib().orc_pose_prior_residual(self.h, dptr(q), dptr(x), dptr(res))
        return comp, x, res

    def prior_factors(self, comp):
        nd = self.arrays.ndims
        L = np.empty(nd * nd); cl = C.c_double()
        lib().orc_prior_factors(self.h, C.c_int(comp), dptr(L), C.byref(cl))
        return L.reshape(nd, nd), cl.value

    def evaluate(self, p, q, w, corr_idx, data, beta_pose, beta_shape, aggregate=0, nthreads=1):
        p, q, w = (np.ascontiguousarray(a, np.float64) for a in (p, q, w))
        ci = np.ascontiguousarray(corr_idx, np.int32); data = np.ascontiguousarray(data, np.float64)
        cost = C.c_double(); comp = C.c_int()
        g = np.empty(self.P); H = np.empty(self.P * self.P)
        lib().orc_evaluate(self.h, dptr(p), dptr(q), dptr(w), iptr(ci), dptr(data), C.c_int(len(ci)),
                           C.c_double(beta_pose), C.c_double(beta_shape), C.c_int(aggregate), C.c_int(nthreads),
                           C.byref(cost), dptr(g), dptr(H), C.byref(comp))
        return cost.value, g, H.reshape(self.P, self.P), comp.value

    def optimize(self, part_map, num_parts, data, labels, opt: Options, p, q, w, aggregate=0, nthreads=1):
        """Runs the restated optimize(); returns dict(p,q,w,stats,trace_cost,trace_acc,corr,cloud)."""
        pm = np.ascontiguousarray(part_map, np.int32)
        data = np.ascontiguousarray(data, np.float64); labels = np.ascontiguousarray(labels, np.int32)
        p = np.array(p, np.float64).copy(); q = np.array(q, np.float64).reshape(-1).copy()
        w = np.array(w, np.float64).copy()
        N = len(labels)
        st = Stats()
        tc = np.zeros(opt.icp_iters * (opt.max_iters_per_icp + 1)); ta = np.zeros(opt.icp_iters * opt.max_iters_per_icp,
                                                                                 np.int32)
        corr = np.empty(N, np.int32); cloud = np.empty(3 * self.V)
        lib().orc_optimize(self.h, C.c_int(num_parts), iptr(pm), dptr(data), iptr(labels), C.c_int(N), C.byref(opt),
                           C.c_int(aggregate), C.c_int(nthreads), dptr(p), dptr(q), dptr(w), C.byref(st), dptr(tc),
                           iptr(ta), iptr(corr), dptr(cloud))
        return dict(p=p, q=q.reshape(self.J, 4), w=w, stats=st, trace_cost=tc, trace_acc=ta, corr=corr,
                    cloud=cloud.reshape(self.V, 3))

    def batch_runner(self, part_map, num_parts, data, labels, copies, opt: Options, p, q, w, aggregate=1, nworkers=1):
        """`copies` copies of one frame marshalled ONCE; returns a zero-argument callable that runs one optimize() per copy,
        one per core (the timed CPU baseline of the frame-batch configurations must not time numpy concatenations)."""
        pm = np.ascontiguousarray(part_map, np.int32)
        N = len(labels)
        offs = (np.arange(copies + 1, dtype=np.int64) * N).astype(np.int32)
        dat = np.ascontiguousarray(np.tile(np.asarray(data, np.float64).reshape(-1, 3), (copies, 1)))
        lab = np.ascontiguousarray(np.tile(np.asarray(labels, np.int32), copies))
        P0 = np.tile(np.asarray(p, np.float64).reshape(1, 3), (copies, 1)); Q0 = np.tile(np.asarray(q, np.float64).reshape(1, -1), (copies, 1))
        W0 = np.tile(np.asarray(w, np.float64).reshape(1, -1), (copies, 1))
        st = (Stats * copies)()
        keep = (pm, offs, dat, lab, P0, Q0, W0, st)

        def run():
            Pc, Qc, Wc = P0.copy(), Q0.copy(), W0.copy()
            lib().orc_optimize_batch(self.h, C.c_int(num_parts), iptr(pm), C.c_int(copies), dptr(dat), iptr(lab), iptr(offs), C.byref(opt),
                                     C.c_int(aggregate), C.c_int(nworkers), dptr(Pc), dptr(Qc), dptr(Wc), st)
            return Pc, Qc, Wc
        run._keep = keep
        return run

    def optimize_batch(self, part_map, num_parts, datas, labels, opt: Options, p, q, w, aggregate=1, nworkers=1):
        """Independent frames on independent cores (one single-threaded optimize() per worker)."""
        F = len(datas)
        pm = np.ascontiguousarray(part_map, np.int32)
        offs = np.zeros(F + 1, np.int32)
        for f in range(F):
            offs[f + 1] = offs[f] + len(labels[f])
        data = np.ascontiguousarray(np.concatenate([np.asarray(d, np.float64).reshape(-1, 3) for d in datas], 0))
        lab = np.ascontiguousarray(np.concatenate([np.asarray(l, np.int32) for l in labels]))
        p = np.array(p, np.float64).reshape(F, 3).copy(); q = np.array(q, np.float64).reshape(F, -1).copy()
        w = np.array(w, np.float64).reshape(F, -1).copy()
        st = (Stats * F)()
        lib().orc_optimize_batch(self.h, C.c_int(num_parts), iptr(pm), C.c_int(F), dptr(data), iptr(lab), iptr(offs), C.byref(opt),
                                 C.c_int(aggregate), C.c_int(nworkers), dptr(p), dptr(q), dptr(w), st)
        return p, q.reshape(F, self.J, 4), w, list(st)


def set_threading(persistent_pool=False, parallel_nn=False):
    """Threading style of the timed CPU baseline: spawn/join per evaluation like the reference (AvatarOptimizer.cpp:
    327-343) or a persistent pool; nearest-neighbour queries serial like the reference (:896-904) or split over threads."""
    lib().orc_set_threading(C.c_int(int(persistent_pool)), C.c_int(int(parallel_nn)))


def set_nn_implementation(kind="bruteforce"):
    """"bruteforce": the oracle's ordered exhaustive scan; "nanoflann": the reference's own KD-tree search from
    oracle/_ref (prebuilt in the build container; bit-identical results).  Returns the kind actually in effect."""
    if kind == "nanoflann" and have_reference_nn():
        ref = C.CDLL(_REF_SO)
        lib().orc_set_nn_override(C.cast(ref.ref_find_nn_inverted_mt, C.c_void_p))
        set_nn_implementation._keep = ref
        return "nanoflann"
    lib().orc_set_nn_override(C.c_void_p(0))
    return "bruteforce"


def hardware_concurrency():
    return int(lib().orc_hardware_concurrency())


def rot_to_quat(R):
    """(J,3,3) -> (J,4) xyzw, as optimize() converts ava.r (AvatarOptimizer.cpp:1250-1254)."""
    R = np.asarray(R, np.float64)
    out = np.empty((R.shape[0], 4))
    for j in range(R.shape[0]):
        rc = np.ascontiguousarray(R[j].T).reshape(-1)
        q = np.empty(4)
        lib().orc_rot_to_quat(dptr(rc), dptr(q))
        out[j] = q
    return out


def quat_to_rot(q):
    q = np.asarray(q, np.float64).reshape(-1, 4)
    out = np.empty((q.shape[0], 3, 3))
    for j in range(q.shape[0]):
        qq = np.ascontiguousarray(q[j]); rc = np.empty(9)
        lib().orc_quat_to_rot(dptr(qq), dptr(rc))
        out[j] = rc.reshape(3, 3).T
    return out
