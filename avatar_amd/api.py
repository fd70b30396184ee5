"""Python mirror of the reference's class surface for the hot path, on top of the C ABI (include/avt.h).

Same names, members, defaults and call protocol as `ark::AvatarModel`, `ark::Avatar` (include/Avatar.h:64-220)
and `ark::AvatarOptimizer` (include/AvatarOptimizer.h:11-61), so that the parity tests read like the callers in
demo.cpp:137-143,251-268.  numpy arrays stand in for Eigen types: clouds are (N,3) float64 (each row one
column of the reference's 3xN matrix), rotations (J,3,3).  All compute goes through libavatar_hip.so; there is
no CPU path here.
"""
from __future__ import annotations

import ctypes as C
import math
import os

import numpy as np

from . import capi
from .capi import ModelArrays, Options, Profile, Stats, bptr, dptr, iptr


class AvtError(RuntimeError):
    pass


def _check(rc):
    if rc != 0:
        e = AvtError(capi.load_library().avt_last_error().decode())
        e.status = rc            # avt.h: AVT_STATUS_NO_DEVICE 2, AVT_STATUS_DEVICE_FAULT 3
        raise e


# ---- rotation <-> quaternion exactly as optimize() converts (AvatarOptimizer.cpp:1250-1254, :1494-1496):
# Matrix3 -> Quaternion -> AngleAxis -> Quaternion on the way in, Quaternion::toRotationMatrix on the way out
# (Eigen 3.3 closed forms).
def rot_to_quat(R):
    """(N,3,3) rotations -> (N,4) quaternions (x,y,z,w).  Vectorised over N; every element goes through exactly the scalar
    operations of the Eigen closed forms (same order, same roundings as oracle.rot_to_quat)."""
    m = np.asarray(R, np.float64).reshape(-1, 3, 3)
    N = m.shape[0]
    q = np.empty((N, 4))
    tr = m[:, 0, 0] + m[:, 1, 1] + m[:, 2, 2]
    pos = tr > 0
    with np.errstate(invalid="ignore", divide="ignore"):
        # trace > 0 branch
        t = np.sqrt(np.where(pos, tr, 0.0) + 1.0)
        ti = 0.5 / t
        qa = np.stack([(m[:, 2, 1] - m[:, 1, 2]) * ti, (m[:, 0, 2] - m[:, 2, 0]) * ti, (m[:, 1, 0] - m[:, 0, 1]) * ti, 0.5 * t], 1)
        # otherwise: largest diagonal entry i, then j, k cyclic
        i = np.where(m[:, 1, 1] > m[:, 0, 0], 1, 0)
        ar = np.arange(N)
        i = np.where(m[:, 2, 2] > m[ar, i, i], 2, i)
        j = (i + 1) % 3
        k = (j + 1) % 3
        t2 = np.sqrt(np.where(pos, 1.0, m[ar, i, i] - m[ar, j, j] - m[ar, k, k] + 1.0))
        t2i = 0.5 / t2
        qb = np.empty((N, 4))
        qb[ar, i] = 0.5 * t2
        qb[:, 3] = (m[ar, k, j] - m[ar, j, k]) * t2i
        qb[ar, j] = (m[ar, j, i] + m[ar, i, j]) * t2i
        qb[ar, k] = (m[ar, k, i] + m[ar, i, k]) * t2i
        q = np.where(pos[:, None], qa, qb)
        # Quaternion -> AngleAxis -> Quaternion
        nrm = np.sqrt(q[:, 0] * q[:, 0] + q[:, 1] * q[:, 1] + q[:, 2] * q[:, 2])
        tiny = nrm < np.finfo(float).eps
        if tiny.any():
            mx = np.abs(q[:, :3]).max(1)
            safe = np.where(mx > 0, mx, 1.0)
            alt = np.where(mx > 0, mx * np.sqrt(((q[:, :3] / safe[:, None]) ** 2).sum(1)), 0.0)
            nrm = np.where(tiny, alt, nrm)
        nz = nrm != 0.0
        # transcendental functions through libm, one element at a time: numpy's vector loops may differ from it by an ulp
        ang = np.array([2.0 * math.atan2(a, b) if z else 0.0 for a, b, z in zip(nrm, np.abs(q[:, 3]), nz)])
        sgn = np.where(q[:, 3] < 0, -nrm, nrm)
        axis = np.where(nz[:, None], q[:, :3] / np.where(nz, sgn, 1.0)[:, None], np.array([1.0, 0.0, 0.0])[None, :])
        ha = 0.5 * ang
        out = np.empty((N, 4))
        out[:, 3] = [math.cos(h) for h in ha]
        out[:, :3] = np.array([math.sin(h) for h in ha])[:, None] * axis
    return out


def quat_to_rot(q):
    """(N,4) quaternions (x,y,z,w) -> (N,3,3) (Eigen Quaternion::toRotationMatrix), vectorised over N."""
    q = np.asarray(q, np.float64).reshape(-1, 4)
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    out = np.empty((q.shape[0], 3, 3))
    out[:, 0, 0] = 1 - (tyy + tzz); out[:, 0, 1] = txy - twz; out[:, 0, 2] = txz + twy
    out[:, 1, 0] = txy + twz; out[:, 1, 1] = 1 - (txx + tzz); out[:, 1, 2] = tyz - twx
    out[:, 2, 0] = txz - twy; out[:, 2, 1] = tyz + twx; out[:, 2, 2] = 1 - (txx + tyy)
    return out


class AvatarModel:
    """`struct AvatarModel` (Avatar.h:64-151).  Built from SMPL-npz-style arrays (dict) or from a directory
    holding `model.npz` (+ optional `pose_prior.txt`), the reference's model_dir convention (AvatarModel.cpp:18-23)."""

    def __init__(self, model=None, limit_one_joint_per_point=False, handle=None):
        """limit_one_joint_per_point (Avatar.h:76-77): the optimiser's forward model binds every point to its largest-weight joint
        only (AvatarModel.cpp:190-196 - the reference honours it in the legacy text format; here it is a property of the model
        description, whatever the source); Avatar::update() keeps all weights."""
        if isinstance(model, (str, os.PathLike)):
            model = load_model_dir(model)
        if model is None:
            raise AvtError("AvatarModel: no model data (the reference's data/avatar-model download is not bundled)")
        self.smpl = model
        self.arrays = ModelArrays(model, limit_one_joint_per_point)
        self._desc = self.arrays.desc()
        self._lib = capi.load_library()
        if handle is not None:      # an avt_model* built elsewhere (avt_model_unpack / avt_shard_broadcast_model): adopt it
            self.h = handle
            V, J, K, F, P = (C.c_int() for _ in range(5))
            _check(self._lib.avt_model_dims(self.h, C.byref(V), C.byref(J), C.byref(K), C.byref(F), C.byref(P)))
            a = self.arrays
            if (V.value, J.value, K.value, F.value, P.value) != (a.V, a.J, a.K, a.F, a.P):
                raise AvtError("AvatarModel: adopted handle and host arrays disagree on the model dimensions")
        else:
            self.h = C.c_void_p()
            _check(self._lib.avt_model_create(C.byref(self._desc), C.byref(self.h)))
        self.parent = self.arrays.parent
        self.mesh = self.arrays.mesh
        ijp = np.empty(3 * self.numJoints()); jsr = np.empty(3 * self.numJoints() * self.numShapeKeys())
        _check(self._lib.avt_model_joint_regression(self.h, dptr(ijp), dptr(jsr)))
        self.initialJointPos = ijp.reshape(-1, 3)
        self.jointShapeReg = jsr.reshape(self.numShapeKeys(), -1).T
        mj = np.empty(self.numPoints(), np.int32)
        _check(self._lib.avt_model_main_joint(self.h, iptr(mj)))
        self.mainJoint = mj
        self._default_ctx = None

    def numJoints(self): return self.arrays.J
    def numPoints(self): return self.arrays.V
    def numShapeKeys(self): return self.arrays.K
    def numFaces(self): return self.arrays.F
    def hasPosePrior(self): return self.arrays.ncomps > 0

    def default_ctx(self):
        if self._default_ctx is None:
            self._default_ctx = Context(self, self.numJoints(), np.arange(self.numJoints(), dtype=np.int32), 1024, 8)
        return self._default_ctx

    def __del__(self):
        try:
            self._default_ctx = None
            self._lib.avt_model_destroy(self.h)
        except Exception:
            pass


def _load_pcd_ascii(path):
    """loadPCDToPointVectorFast (AvatarHelpers.cpp:13-52): WIDTH gives the point count, DATA must be ascii, then 3 numbers per point."""
    tok = open(path).read().split("\n")
    n, body = -1, None
    for i, line in enumerate(tok):
        parts = line.split()
        if not parts:
            continue
        if parts[0] == "WIDTH":
            n = int(parts[1])
        elif parts[0] == "DATA":
            if n < 0:
                raise AvtError(f"invalid PCD file at {path}: no WIDTH field before data")
            if len(parts) < 2 or parts[1] != "ascii":
                raise AvtError(f"non-ascii PCD not supported: {path}")
            body = " ".join(tok[i + 1:]).split()
            break
    if body is None or len(body) < 3 * n:
        raise AvtError(f"invalid PCD file at {path}: unexpected EOF")
    return np.array(body[:3 * n], dtype=np.float64)


def load_legacy_model_dir(path):
    """The reference's deprecated ad-hoc model format (AvatarModel.cpp:128-288): skeleton.txt (joints with parents and rest
    positions, then every point's (joint, weight) list), model.pcd (base cloud), shapekey/*.pcd (one key cloud each; the reference
    takes them in directory order, which is unspecified - here sorted by file name), joint_shape_regressor.txt or
    joint_regressor.txt, mesh.txt.  Returns the same dict layout as model.npz."""
    tok = iter(open(os.path.join(path, "skeleton.txt")).read().split())
    J, V = int(next(tok)), int(next(tok))
    parent = np.zeros(J, np.int64)
    for i in range(J):
        jid = int(next(tok)); parent[jid] = int(next(tok)); next(tok); [next(tok) for _ in range(3)]      # name and rest position: not needed
    parent[0] = -1
    W = np.zeros((V, J))
    for v in range(V):
        for _ in range(int(next(tok))):
            j = int(next(tok)); W[v, j] = float(next(tok))
    m = {"v_template": _load_pcd_ascii(os.path.join(path, "model.pcd")).reshape(V, 3), "weights": W,
         "kintree_table": np.stack([parent, np.arange(J)])}
    kdir = os.path.join(path, "shapekey")
    keys = [_load_pcd_ascii(os.path.join(kdir, f)) for f in sorted(os.listdir(kdir))] if os.path.isdir(kdir) else []
    K = len(keys)
    m["shapedirs"] = np.stack(keys, 1).reshape(V, 3, K) if K else np.zeros((V, 3, 0))
    jsr_path, jr_path = os.path.join(path, "joint_shape_regressor.txt"), os.path.join(path, "joint_regressor.txt")
    m["J_regressor"] = np.zeros((J, V))
    if os.path.exists(jsr_path):
        t = open(jsr_path).read().split()
        nk = int(t[0]); vals = np.array(t[1:], dtype=np.float64)
        m["joint_shape_reg_base"] = vals[:3 * J]
        m["joint_shape_reg"] = vals[3 * J:3 * J + 3 * J * nk].reshape(3 * J, nk)       # row by row in the file (AvatarModel.cpp:238-242)
    elif os.path.exists(jr_path):
        t = iter(open(jr_path).read().split())
        for j in range(int(next(t))):
            for _ in range(int(next(t))):
                v = int(next(t)); m["J_regressor"][j, v] = float(next(t))
    mesh_path = os.path.join(path, "mesh.txt")
    if os.path.exists(mesh_path):
        t = open(mesh_path).read().split()
        m["f"] = np.array(t[1:1 + 3 * int(t[0])], dtype=np.int64).reshape(-1, 3)
    else:
        m["f"] = np.zeros((0, 3), np.int64)
    return m


def load_model_dir(path):
    """model.npz in SMPL layout (AvatarModel.cpp:26-104) + pose_prior.txt (GaussianMixture.cpp:12-58); without a model.npz the
    reference's legacy text format (AvatarModel.cpp:128-288)."""
    if os.path.exists(os.path.join(path, "model.npz")):
        with np.load(os.path.join(path, "model.npz")) as z:
            m = {k: z[k] for k in z.files}
    else:
        m = load_legacy_model_dir(path)
    pp = os.path.join(path, "pose_prior.txt")
    if os.path.exists(pp) and "prior_weight" not in m:
        tok = open(pp).read().split()
        nc, nd = int(tok[0]), int(tok[1])
        vals = np.array(tok[2:], dtype=np.float64)
        m["prior_weight"] = vals[:nc]
        m["prior_mean"] = vals[nc:nc + nc * nd].reshape(nc, nd)
        m["prior_cov"] = vals[nc + nc * nd:nc + nc * nd + nc * nd * nd].reshape(nc, nd, nd)
    return m


class Context:
    """One HIP device + stream + persistent buffers (avt_ctx)."""

    def __init__(self, model: AvatarModel, num_parts, part_map, max_points, max_frames, device=None):
        self.model = model
        self._lib = capi.load_library()
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        pm = np.ascontiguousarray(part_map, np.int32)
        if len(pm) < model.numJoints():
            raise AvtError("part_map must have at least numJoints entries (AvatarOptimizer.cpp:1229)")
        self.h = C.c_void_p()
        _check(self._lib.avt_ctx_create(C.c_int(device), model.h, C.c_int(num_parts), iptr(pm), C.c_int(max_points),
                                        C.c_int(max_frames), C.byref(self.h)))
        self.max_points, self.max_frames, self.num_parts = max_points, max_frames, num_parts

    def __del__(self):
        try:
            self._lib.avt_ctx_destroy(self.h)
        except Exception:
            pass

    # ---- thin wrappers -------------------------------------------------------------------------
    def lbs_update(self, w, p, R):
        """Batched Avatar::update. w (F,K), p (F,3), R (F,J,3,3). Returns cloud (F,V,3), jointPos (F,J,3),
        jointTrans (F,J,12)."""
        m = self.model
        w = np.ascontiguousarray(np.atleast_2d(w), np.float64); p = np.ascontiguousarray(np.atleast_2d(p), np.float64)
        R = np.asarray(R, np.float64).reshape(-1, m.numJoints(), 3, 3)
        F = w.shape[0]
        Rcm = np.ascontiguousarray(np.transpose(R, (0, 1, 3, 2)))  # column-major 3x3 blocks
        cloud = np.empty((F, m.numPoints(), 3)); jp = np.empty((F, m.numJoints(), 3)); jt = np.empty((F, m.numJoints(), 12))
        _check(self._lib.avt_lbs_update(self.h, C.c_int(F), dptr(w), dptr(p), dptr(Rcm), dptr(cloud), dptr(jp), dptr(jt)))
        return cloud, jp, jt

    def visibility(self, cloud, enable=True):
        cloud = np.ascontiguousarray(cloud, np.float64)
        vis = np.empty(self.model.numPoints(), np.uint8)
        _check(self._lib.avt_visibility(self.h, dptr(cloud), C.c_int(int(enable)), bptr(vis)))
        return vis

    def nn(self, model_cloud, visible, data, labels):
        mc = np.ascontiguousarray(model_cloud, np.float64); vis = np.ascontiguousarray(visible, np.uint8)
        data = np.ascontiguousarray(data, np.float64); labels = np.ascontiguousarray(labels, np.int32)
        out = np.empty(len(labels), np.int32)
        _check(self._lib.avt_nn(self.h, dptr(mc), bptr(vis), dptr(data), iptr(labels), C.c_int(len(labels)), iptr(out)))
        return out

    def optimize_batch(self, datas, labels, opt: Options, p, q, w):
        """datas/labels: lists of per-frame arrays. p (F,3), q (F,J,4), w (F,K) start states. Returns p,q,w,stats."""
        F = len(datas)
        offs = np.zeros(F + 1, np.int32)
        for f in range(F):
            offs[f + 1] = offs[f] + len(labels[f])
        data = np.ascontiguousarray(np.concatenate([np.asarray(d, np.float64).reshape(-1, 3) for d in datas], 0))
        lab = np.ascontiguousarray(np.concatenate([np.asarray(l, np.int32) for l in labels]))
        p = np.array(p, np.float64).reshape(F, 3).copy(); q = np.array(q, np.float64).reshape(F, -1).copy()
        w = np.array(w, np.float64).reshape(F, -1).copy()
        st = (Stats * F)()
        _check(self._lib.avt_optimize_batch(self.h, C.c_int(F), dptr(data), iptr(lab), iptr(offs), C.byref(opt), dptr(p),
                                            dptr(q), dptr(w), st))
        return p, q.reshape(F, -1, 4), w, list(st)

    def optimize_posed(self, data, labels, opt: Options, p, q, w):
        """One frame on host memory with the posed outputs of the closing update() in the same synchronisation (avt_optimize_posed): returns
        p, q (J,4), w, stats, cloud (V,3), jointPos (J,3), jointTrans (J,12)."""
        m = self.model
        data = np.ascontiguousarray(np.asarray(data, np.float64).reshape(-1, 3)); lab = np.ascontiguousarray(np.asarray(labels, np.int32))
        p = np.array(p, np.float64).reshape(3).copy(); q = np.array(q, np.float64).reshape(-1).copy(); w = np.array(w, np.float64).reshape(-1).copy()
        st = Stats()
        cloud = np.empty((m.numPoints(), 3)); jp = np.empty((m.numJoints(), 3)); jt = np.empty((m.numJoints(), 12))
        _check(self._lib.avt_optimize_posed(self.h, dptr(data), iptr(lab), C.c_int(len(lab)), C.byref(opt), dptr(p), dptr(q), dptr(w), C.byref(st),
                                            dptr(cloud), dptr(jp), dptr(jt)))
        self._N = np.array([len(lab)], np.int32)
        return p, q.reshape(-1, 4), w, st, cloud, jp, jt

    def host_optimize_call(self, data, labels, opt: Options, p0, q0, w0):
        """The reference's call shape - optimize(const CloudType&, const VectorXi&, ...) on HOST memory (AvatarOptimizer.h:17-19) - as a
        closure over prepared contiguous arrays: every call is ONE avt_optimize (H2D of the cloud and the start state, the fit, D2H of
        p / q / w / stats) with no Python-side array work beyond re-installing the 109 start values.  Returns (call, p, q, w, stats)."""
        data = np.ascontiguousarray(np.asarray(data, np.float64).reshape(-1, 3))
        lab = np.ascontiguousarray(np.asarray(labels, np.int32))
        ps, qs, ws = (np.array(a, np.float64).ravel().copy() for a in (p0, q0, w0))
        p, q, w = ps.copy(), qs.copy(), ws.copy()
        st = Stats()
        fn, h, n = self._lib.avt_optimize, self.h, C.c_int(len(lab))
        a_d, a_l, a_o, a_p, a_q, a_w, a_s = dptr(data), iptr(lab), C.byref(opt), dptr(p), dptr(q), dptr(w), C.byref(st)

        def call():
            p[:] = ps; q[:] = qs; w[:] = ws
            _check(fn(h, a_d, a_l, n, a_o, a_p, a_q, a_w, a_s))
        call._keep = (data, lab, opt)
        return call, p, q, w, st

    def frames_upload(self, datas, labels):
        F = len(datas)
        offs = np.zeros(F + 1, np.int32)
        for f in range(F):
            offs[f + 1] = offs[f] + len(labels[f])
        data = np.ascontiguousarray(np.concatenate([np.asarray(d, np.float64).reshape(-1, 3) for d in datas], 0))
        lab = np.ascontiguousarray(np.concatenate([np.asarray(l, np.int32) for l in labels]))
        _check(self._lib.avt_frames_upload(self.h, C.c_int(F), dptr(data), iptr(lab), iptr(offs)))
        self._F = F

    def render_frames(self, w, p, R, intrin=None, res_scale=1, painter=False):
        """Synthesise depth frames of posed avatars on the GPU, resident in this context.
        w (F,K), p (F,3), R (F,J,3,3). Returns the number of points of every frame.  painter=True: the reference's
        painter's-order renderer pixel for pixel (AVT_RENDER_PAINTER) instead of the z-buffer."""
        from . import synth
        k = dict(synth.K4A_INTRIN) if intrin is None else dict(intrin)
        m = self.model
        w = np.ascontiguousarray(np.atleast_2d(w), np.float64); p = np.ascontiguousarray(np.atleast_2d(p), np.float64)
        R = np.asarray(R, np.float64).reshape(-1, m.numJoints(), 3, 3)
        F = w.shape[0]
        Rcm = np.ascontiguousarray(np.transpose(R, (0, 1, 3, 2)))
        n = np.zeros(F, np.int32)
        _check(self._lib.avt_synth_render_frames_mode(self.h, C.c_int(F), dptr(w), dptr(p), dptr(Rcm), C.c_double(k["fx"] * res_scale),
                                                      C.c_double(k["fy"] * res_scale), C.c_double(k["cx"] * res_scale),
                                                      C.c_double(k["cy"] * res_scale), C.c_int(k["width"] * res_scale),
                                                      C.c_int(k["height"] * res_scale), C.c_int(1 if painter else 0), iptr(n)))
        self._F = F
        self._N = n
        self._img_shape = (k["height"] * res_scale, k["width"] * res_scale)
        return n

    def render_images(self, frame):
        """(depth (H,W) float32, part mask (H,W) uint8) of frame `frame` of the last painter=True render_frames call:
        what AvatarRenderer::renderDepth / renderPartMask return."""
        H, W = self._img_shape
        depth = np.empty((H, W), np.float32); mask = np.empty((H, W), np.uint8)
        _check(self._lib.avt_synth_render_images(self.h, C.c_int(frame), depth.ctypes.data_as(C.POINTER(C.c_float)),
                                                 mask.ctypes.data_as(C.POINTER(C.c_ubyte))))
        return depth, mask

    def frame_download(self, frame):
        n = int(self._N[frame])
        data = np.empty((n, 3)); lab = np.empty(n, np.int32)
        _check(self._lib.avt_frames_download(self.h, C.c_int(frame), dptr(data), iptr(lab)))
        return data, lab

    def state_upload(self, p, q, w):
        F = self._F
        p = np.ascontiguousarray(np.asarray(p, np.float64).reshape(F, 3)); q = np.ascontiguousarray(np.asarray(q, np.float64).reshape(F, -1))
        w = np.ascontiguousarray(np.asarray(w, np.float64).reshape(F, -1))
        _check(self._lib.avt_state_upload(self.h, C.c_int(F), dptr(p), dptr(q), dptr(w)))

    def optimize_resident(self, opt: Options):
        _check(self._lib.avt_optimize_resident(self.h, C.byref(opt)))

    def state_reset(self):
        """Asynchronous device-side reinstall of the last uploaded start state (no host transfer, no synchronisation)."""
        _check(self._lib.avt_state_reset(self.h))

    def sync(self):
        _check(self._lib.avt_sync(self.h))

    def state_download(self):
        F = self._F; m = self.model
        p = np.empty((F, 3)); q = np.empty((F, m.numJoints() * 4)); w = np.empty((F, m.numShapeKeys()))
        st = (Stats * F)()
        _check(self._lib.avt_state_download(self.h, dptr(p), dptr(q), dptr(w), st))
        return p, q.reshape(F, -1, 4), w, list(st)

    def correspondences(self, frame, n):
        out = np.empty(n, np.int32)
        _check(self._lib.avt_get_correspondences(self.h, C.c_int(frame), iptr(out)))
        return out

    def cloud(self, frame=0):
        out = np.empty((self.model.numPoints(), 3))
        _check(self._lib.avt_get_cloud(self.h, C.c_int(frame), dptr(out)))
        return out

    def posed(self, frame=0):
        """(cloud (V,3), jointPos (J,3), jointTrans (J,12)) left by the update() that ends optimize()."""
        m = self.model
        cloud = np.empty((m.numPoints(), 3)); jp = np.empty((m.numJoints(), 3)); jt = np.empty((m.numJoints(), 12))
        _check(self._lib.avt_get_posed(self.h, C.c_int(frame), dptr(cloud), dptr(jp), dptr(jt)))
        return cloud, jp, jt

    def normal_equations(self, frame=0):
        P = self.model.arrays.P
        H = np.empty((P, P)); g = np.empty(P); cost = C.c_double()
        _check(self._lib.avt_get_normal_equations(self.h, C.c_int(frame), dptr(H), dptr(g), C.byref(cost)))
        return H, g, cost.value

    def cost_trace(self, frame=0, n=11):
        """Objective at entry of the last ICP iteration and after each of its GN iterations (avt_debug_trace): entry i + 1 < entry i
        means iteration i + 1 accepted its trial point."""
        buf = np.zeros(64)
        _check(self._lib.avt_debug_trace(self.h, C.c_int(frame), dptr(buf)))
        return buf[:n].copy()

    def mfma_count(self, frame=0):
        """fp64 matrix instructions (2048 flop each) the kernels execute for `frame` with the last optimize()'s correspondences, from their
        own trip counts (avt_debug_mfma_count): dict eval_rows / moments / solve, None where the form did not run."""
        a, b, c = C.c_longlong(), C.c_longlong(), C.c_longlong()
        _check(self._lib.avt_debug_mfma_count(self.h, C.c_int(frame), C.byref(a), C.byref(b), C.byref(c)))
        return {k: (None if v.value < 0 else int(v.value)) for k, v in (("eval_rows", a), ("moments", b), ("solve", c))}

    DATA_TERM_ROWS, DATA_TERM_MOMENTS, DATA_TERM_AUTO = 0, 1, 2

    def tuning(self):
        """The knobs this context runs with (include/avt.h avt_tuning) as a capi.Tuning."""
        t = capi.Tuning()
        _check(self._lib.avt_ctx_get_tuning(self.h, C.byref(t)))
        return t

    def set_tuning(self, **kw):
        """Changes knobs of this context (avt_ctx_set_tuning): e.g. ctx.set_tuning(nspec=0, nn_slab=0)."""
        t = self.tuning()
        for k, v in kw.items():
            if k not in dict(t._fields_):
                raise KeyError(k)
            setattr(t, k, int(v))
        _check(self._lib.avt_ctx_set_tuning(self.h, C.byref(t)))
        return self

    def set_data_term(self, form):
        """How the ICP data term of a GN iteration is evaluated (include/avt.h: AVT_DATA_TERM_ROWS / AVT_DATA_TERM_MOMENTS)."""
        _check(self._lib.avt_set_data_term(self.h, C.c_int(int(form))))

    def data_term(self):
        return int(self._lib.avt_get_data_term(self.h))

    def launch_shape(self):
        """(groups, frames per group, k_eval workgroups per frame) of optimize() over the resident frames."""
        g, n, G = C.c_int(), C.c_int(), C.c_int()
        _check(self._lib.avt_launch_shape(self.h, C.byref(g), C.byref(n), C.byref(G)))
        return g.value, n.value, G.value

    def profile_begin(self, classes=None):
        mask = 0xffffffff if classes is None else sum(1 << capi.AVT_K_NAMES.index(c) for c in classes)
        _check(self._lib.avt_profile_select(self.h, C.c_uint(mask)))
        _check(self._lib.avt_profile_begin(self.h))

    def profile_end(self):
        pr = Profile()
        _check(self._lib.avt_profile_end(self.h, C.byref(pr)))
        return {capi.AVT_K_NAMES[i]: (pr.ms[i], pr.launches[i]) for i in range(capi.AVT_K_COUNT)}


class Avatar:
    """`class Avatar` (Avatar.h:155-220): state w, p, r -> update() -> cloud, jointPos, jointTrans."""

    def __init__(self, model: AvatarModel):
        self.model = model
        self.w = np.zeros(model.numShapeKeys())
        self.p = np.zeros(3)
        self.r = np.tile(np.eye(3), (model.numJoints(), 1, 1))   # Avatar.cpp:12-20
        self.cloud = np.zeros((0, 3)); self.jointPos = np.zeros((0, 3)); self.jointTrans = np.zeros((0, 12))

    def update(self):
        c, jp, jt = self.model.default_ctx().lbs_update(self.w[None], self.p[None], self.r[None])
        self.cloud, self.jointPos, self.jointTrans = c[0], jp[0], jt[0]

    # ---- the host-side members beside the tracker path (Avatar.cpp:77-193); include/ark/Avatar.h is the reference-faithful C++ side (std::mt19937
    # draws in the reference's order, Eigen's closed forms) and tests/test_avatar_methods_cpu.py checks it; these mirror the behaviour with numpy
    def smplParams(self):
        """Axis-angle of every joint but the root, 3 (J - 1) numbers (Avatar.cpp:128-137)."""
        from scipy.spatial.transform import Rotation
        return Rotation.from_matrix(self.r[1:]).as_rotvec().reshape(-1)

    def pdf(self):
        """GMM likelihood of the joint rotations as the reference evaluates it (Avatar.cpp:139, GaussianMixture.cpp:22-93: constants normalised by
        the smallest determinant; exponent |L (x - mu)|^2 with L = chol(cov^-1), not its transpose)."""
        a = self.model.arrays
        wt, mu, cov = a.prior_weight, a.prior_mean, a.prior_cov
        n = mu.shape[1]
        dets = np.array([np.prod(np.diag(np.linalg.cholesky(c))) for c in cov])
        consts = wt / (2 * np.pi) ** (n * 0.5) / dets * dets.min()
        x = self.smplParams()
        return float(sum(consts[i] * np.exp(-0.5 * np.sum((np.linalg.cholesky(np.linalg.inv(cov[i])) @ (x - mu[i])) ** 2)) for i in range(len(wt))))

    def randomize(self, randomize_pose=True, randomize_shape=True, randomize_root_pos_rot=True, seed=None):
        """Avatar::randomize (Avatar.cpp:77-126) with numpy's generator (NOT std::mt19937's stream: the C++ facade has that): normal shape
        coefficients, the pose from the prior's LAST component (the reference's component loop has no break), root position and rotation in
        the reference's ranges."""
        from scipy.spatial.transform import Rotation
        rng = np.random.default_rng(seed)
        J = self.model.numJoints()
        if randomize_shape:
            self.w = rng.normal(size=self.model.numShapeKeys())
        if randomize_pose:
            a = self.model.arrays
            x = a.prior_mean[-1] + np.linalg.cholesky(a.prior_cov[-1]) @ rng.normal(size=a.prior_mean.shape[1])
            self.r[1:] = Rotation.from_rotvec(x.reshape(J - 1, 3)).as_matrix()
        if randomize_root_pos_rot:
            self.p = np.array([rng.uniform(-1, 1), rng.uniform(-0.5, 0.5), rng.uniform(2.2, 4.5)])
            up = rng.uniform(-np.pi / 3, np.pi / 3) + np.pi
            th, ph = rng.uniform(0, 2 * np.pi), rng.uniform(-np.pi / 2, np.pi / 2)
            axis = np.array([np.sin(ph) * np.cos(th), np.cos(ph), np.sin(ph) * np.sin(th)])
            self.r[0] = Rotation.from_rotvec(axis * rng.normal(0, 0.2)).as_matrix() @ Rotation.from_rotvec([0, up, 0]).as_matrix()

    def alignToJoints(self, joint_pos):
        """Pose (and w[0]) from 24 target joint positions, NaN rows = not seen (Avatar.cpp:141-193, quirks included: see include/ark/Avatar.h)."""
        pos = np.asarray(joint_pos, float).reshape(-1, 3)
        ij, parent = self.model.initialJointPos, self.model.parent
        assert len(pos) == 24 == self.model.numJoints()

        def two(a, b):      # Quaternion::FromTwoVectors(a, b).toRotationMatrix()
            a, b = a / np.linalg.norm(a), b / np.linalg.norm(b)
            v, c = np.cross(a, b), float(a @ b)
            if c < -1 + 1e-12:
                e = np.eye(3)[int(np.argmin(np.abs(a)))]
                ax = np.cross(e, a); ax /= np.linalg.norm(ax)
                return 2 * np.outer(ax, ax) - np.eye(3)
            K = np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
            return np.eye(3) + K + K @ K / (1 + c)
        vr, vrt = ij[3] - ij[0], pos[3] - pos[0]
        if not np.isnan(pos[0, 0]):
            self.p = pos[0].copy()
        self.r[0] = two(vr, vrt) if not (np.isnan(vr[0]) or np.isnan(vrt[0])) else np.eye(3)
        rt = [None] * 24
        rt[0] = self.r[0].copy()
        scale = np.mean([np.linalg.norm(pos[i] - pos[parent[i]]) / np.linalg.norm(ij[i] - ij[parent[i]]) for i in range(1, 24)])
        w0 = np.linalg.norm(ij[6] - ij[0]) * (scale - 1.0) * 32.0
        self.w[0] = 1.5 if np.isnan(w0) else w0
        for i in range(1, 24):
            rt[i] = rt[parent[i]]
            if not np.isnan(pos[i, 0]):
                rt[i] = two(ij[i] - ij[parent[i]], pos[i] - pos[parent[i]])
                self.r[i] = rt[parent[i]].T @ rt[i]
            else:
                self.r[i] = np.eye(3)


class AvatarOptimizer:
    """`class AvatarOptimizer` (AvatarOptimizer.h:11-61).  `intrin` and `image_size` are accepted for signature
    parity; they do not influence optimize() in the reference either (renderer unused, AvatarOptimizer.cpp:1271,
    :1369-1385)."""

    ROT_SIZE = 4

    def __init__(self, ava: Avatar, intrin=None, image_size=None, num_parts=None, part_map=None, max_points=200000):
        self.ava = ava
        self.intrin, self.imageSize = intrin, image_size
        J = ava.model.numJoints()
        self.partMap = np.arange(J, dtype=np.int32) if part_map is None else np.asarray(part_map, np.int32)
        self.numParts = J if num_parts is None else num_parts
        self.betaPose, self.betaShape = 0.1, 1.0          # AvatarOptimizer.h:28
        self.nnStep = 20                                   # :33
        self.maxItersPerICP = 10                           # :36
        self.enableOcclusion = True                        # :39
        self.functionTolerance = 1e-4                      # not a member of the reference's class: what its optimize() hard-codes at AvatarOptimizer.cpp:1333 (0 = no early exit)
        self.r = np.zeros((J, 4)); self.r[:, 3] = 1.0      # quaternions (x,y,z,w), :25
        self.ctx = Context(ava.model, self.numParts, self.partMap, max_points, 1)
        self.last_stats = None

    def options(self, icp_iters=1, num_threads=4) -> Options:
        o = Options.reference_defaults()
        o.beta_pose, o.beta_shape = self.betaPose, self.betaShape
        o.nn_step, o.max_iters_per_icp = self.nnStep, self.maxItersPerICP
        o.enable_occlusion, o.icp_iters, o.num_threads = int(self.enableOcclusion), icp_iters, num_threads
        o.function_tolerance = self.functionTolerance
        return o

    def optimize(self, data_cloud, data_part_labels, icp_iters=1, num_threads=4):
        ava = self.ava
        self.r = rot_to_quat(ava.r)                                              # :1250-1254
        if icp_iters >= 1:      # one call, one synchronisation: the fit and the outputs of the update() the launch sequence ends with (:1497)
            ava.p, self.r, ava.w, self.last_stats, ava.cloud, ava.jointPos, ava.jointTrans = self.ctx.optimize_posed(
                data_cloud, data_part_labels, self.options(icp_iters, num_threads), ava.p, self.r, ava.w)
        else:
            p, q, w, st = self.ctx.optimize_batch([data_cloud], [data_part_labels], self.options(icp_iters, num_threads),
                                                  ava.p[None], self.r[None], ava.w[None])
            ava.p, self.r, ava.w, self.last_stats = p[0], q[0], w[0], st[0]
            ava.cloud, ava.jointPos, ava.jointTrans = self.ctx.posed(0)
        ava.r = quat_to_rot(self.r)                                              # :1494-1496
