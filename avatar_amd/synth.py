"""Deterministic synthetic SMPL-shaped body model + GMM pose prior (workload harness).

The reference ships no model data (SMPL is licence-restricted; `README.md:75-78`), so every test and
benchmark in this repo runs on a seeded stand-in with SMPL's exact sizes: V=6890 vertices, F=13776
faces (closed genus-0 triangle mesh, outward oriented), J=24 joints on the SMPL tree
(`include/Avatar.h:27-58`), K=10 shape keys, <=4 skinning weights per vertex, a sparse joint
regressor, and an 8x69 GMM in the `pose_prior.txt` layout (`GaussianMixture.cpp:20-58`).

The arrays are emitted under SMPL's own npz key names (`AvatarModel.cpp:26-104`: v_template, f,
kintree_table, J_regressor, weights, shapedirs) so the same ingest path serves a real SMPL file.

This module is harness only: it is neither the product (avatar_amd/csrc) nor the oracle (oracle/).
It also carries a small numpy linear-blend-skinning routine used ONLY to pose ground-truth bodies
when synthesising depth clouds, so that workload generation depends on neither of the two.
"""
from __future__ import annotations

import os
import numpy as np

NUM_VERTS = 6890
NUM_FACES = 13776
NUM_JOINTS = 24
NUM_SHAPE = 10

# SMPL kinematic tree, BFS order (include/Avatar.h:27-58)
PARENT = np.array([-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21],
                  dtype=np.int32)

# planned rest joint locations (metres, y up, facing +z, pelvis at origin; +x is the body's left)
_JOINTS = np.array([
    [0.00, 0.00, 0.00],     # 0 pelvis
    [0.075, -0.09, 0.00],   # 1 L_hip
    [-0.075, -0.09, 0.00],  # 2 R_hip
    [0.00, 0.11, -0.02],    # 3 spine1
    [0.10, -0.47, 0.00],    # 4 L_knee
    [-0.10, -0.47, 0.00],   # 5 R_knee
    [0.00, 0.25, 0.00],     # 6 spine2
    [0.09, -0.87, -0.03],   # 7 L_ankle
    [-0.09, -0.87, -0.03],  # 8 R_ankle
    [0.00, 0.31, 0.02],     # 9 spine3
    [0.11, -0.93, 0.09],    # 10 L_foot
    [-0.11, -0.93, 0.09],   # 11 R_foot
    [0.00, 0.52, -0.02],    # 12 neck
    [0.08, 0.43, -0.01],    # 13 L_collar
    [-0.08, 0.43, -0.01],   # 14 R_collar
    [0.00, 0.61, 0.03],     # 15 head
    [0.19, 0.46, -0.02],    # 16 L_shoulder
    [-0.19, 0.46, -0.02],   # 17 R_shoulder
    [0.45, 0.46, -0.03],    # 18 L_elbow
    [-0.45, 0.46, -0.03],   # 19 R_elbow
    [0.70, 0.46, -0.03],    # 20 L_wrist
    [-0.70, 0.46, -0.03],   # 21 R_wrist
    [0.79, 0.46, -0.03],    # 22 L_hand
    [-0.79, 0.46, -0.03],   # 23 R_hand
], dtype=np.float64)


def _capsules():
    """(a, b, ra, rb, scale) solid primitives making up the rest body."""
    caps = []

    def add(a, b, ra, rb=None, scale=(1.0, 1.0, 1.0), mirror=False):
        a = np.asarray(a, float); b = np.asarray(b, float)
        rb_ = ra if rb is None else rb
        caps.append((a, b, ra, rb_, np.asarray(scale, float)))
        if mirror:
            m = np.array([-1.0, 1.0, 1.0])
            caps.append((a * m, b * m, ra, rb_, np.asarray(scale, float)))

    add([0, -0.03, 0.0], [0, 0.22, 0.0], 0.105, 0.10, scale=(1.40, 1.0, 1.0))      # abdomen
    add([0, 0.25, 0.0], [0, 0.41, 0.0], 0.105, 0.10, scale=(1.50, 1.0, 1.0))       # chest
    add([0, 0.45, -0.01], [0, 0.57, 0.01], 0.05)                                    # neck
    add([0, 0.65, 0.03], [0, 0.67, 0.03], 0.088, scale=(0.92, 1.2, 1.05))           # head
    add([0.05, 0.44, -0.01], [0.19, 0.46, -0.02], 0.056, mirror=True)               # collar
    add([0.19, 0.46, -0.02], [0.45, 0.46, -0.03], 0.047, 0.040, mirror=True)        # upper arm
    add([0.45, 0.46, -0.03], [0.70, 0.46, -0.03], 0.040, 0.032, mirror=True)        # forearm
    add([0.70, 0.46, -0.03], [0.85, 0.46, -0.03], 0.034, 0.030, scale=(1.0, 0.7, 1.25), mirror=True)  # hand
    add([0.075, -0.08, 0.0], [0.10, -0.47, 0.0], 0.078, 0.055, mirror=True)         # thigh
    add([0.10, -0.47, 0.0], [0.09, -0.87, -0.03], 0.052, 0.038, mirror=True)        # shin
    add([0.09, -0.905, -0.05], [0.11, -0.925, 0.13], 0.042, 0.036, mirror=True)     # foot
    return caps


def _implicit(p, caps, blend=0.0):
    """Signed pseudo-distance to the union of (anisotropically scaled, tapered) capsules; <0 inside."""
    p = np.asarray(p, float)
    best = None
    acc = None
    for a, b, ra, rb, sc in caps:
        pa = (p - a) / sc
        ba = (b - a) / sc
        t = np.clip((pa @ ba) / (ba @ ba), 0.0, 1.0)
        d = np.linalg.norm(pa - t[:, None] * ba, axis=1) - (ra + (rb - ra) * t)
        if blend > 0.0:
            e = np.exp(-d / blend)
            acc = e if acc is None else acc + e
        else:
            best = d if best is None else np.minimum(best, d)
    if blend > 0.0:
        return -blend * np.log(acc)
    return best


def _voxel_body(h, caps):
    lo = np.array([-0.95, -1.05, -0.22]); hi = np.array([0.95, 0.85, 0.30])
    n = np.ceil((hi - lo) / h).astype(int)
    gx, gy, gz = [lo[i] + (np.arange(n[i]) + 0.5) * h for i in range(3)]
    P = np.stack(np.meshgrid(gx, gy, gz, indexing="ij"), -1).reshape(-1, 3)
    occ = (_implicit(P, caps) < 0.0).reshape(n)
    occ = np.pad(occ, 1)
    # repair non-manifold contacts: fill until no 2x2 diagonal-only edge / 2x2x2 corner-only contacts
    for _ in range(64):
        changed = False
        for ax in range(3):
            o = np.moveaxis(occ, ax, 0)
            a = o[:, :-1, :-1]; b = o[:, 1:, :-1]; c = o[:, :-1, 1:]; d = o[:, 1:, 1:]
            bad1 = a & d & ~b & ~c
            bad2 = b & c & ~a & ~d
            if bad1.any() or bad2.any():
                changed = True
                o[:, 1:, :-1] |= bad1
                o[:, :-1, :-1] |= bad2
        # vertex-only contacts in a 2x2x2 block
        blk = [occ[i:occ.shape[0] - 1 + i, j:occ.shape[1] - 1 + j, k:occ.shape[2] - 1 + k]
               for i in (0, 1) for j in (0, 1) for k in (0, 1)]
        cnt = sum(b.astype(np.int32) for b in blk)
        for q in range(4):
            bad = blk[q] & blk[7 - q] & (cnt == 2)
            if bad.any():
                changed = True
                i, j, k = (q >> 2) & 1, (q >> 1) & 1, q & 1
                # fill the face-neighbour of blk[q] toward blk[7-q] along x
                occ[(1 - i):occ.shape[0] - i, j:occ.shape[1] - 1 + j, k:occ.shape[2] - 1 + k] |= bad
        if not changed:
            break
    return occ, lo - h, h


def _boundary_mesh(occ, origin, h):
    """Quads of the voxel boundary (outward CCW) -> vertices (grid corners) + triangles."""
    quads = []
    idx = np.argwhere(occ)
    # for axis ax and sign s: corners of the face, ordered CCW seen from outside
    corner = {
        (0, 1): [(1, 0, 0), (1, 1, 0), (1, 1, 1), (1, 0, 1)],
        (0, -1): [(0, 0, 0), (0, 0, 1), (0, 1, 1), (0, 1, 0)],
        (1, 1): [(0, 1, 0), (0, 1, 1), (1, 1, 1), (1, 1, 0)],
        (1, -1): [(0, 0, 0), (1, 0, 0), (1, 0, 1), (0, 0, 1)],
        (2, 1): [(0, 0, 1), (1, 0, 1), (1, 1, 1), (0, 1, 1)],
        (2, -1): [(0, 0, 0), (0, 1, 0), (1, 1, 0), (1, 0, 0)],
    }
    for (ax, s), cs in corner.items():
        nb = idx.copy(); nb[:, ax] += s
        empty = ~occ[nb[:, 0], nb[:, 1], nb[:, 2]]
        base = idx[empty]
        q = np.stack([base + np.array(c) for c in cs], 1)  # (n,4,3) integer corner coords
        quads.append(q)
    quads = np.concatenate(quads, 0)
    flat = quads.reshape(-1, 3)
    key = (flat[:, 0].astype(np.int64) << 40) | (flat[:, 1].astype(np.int64) << 20) | flat[:, 2].astype(np.int64)
    uniq, inv = np.unique(key, return_inverse=True)
    verts_i = np.stack([(uniq >> 40) & 0xFFFFF, (uniq >> 20) & 0xFFFFF, uniq & 0xFFFFF], 1).astype(np.float64)
    verts = origin + verts_i * h
    q = inv.reshape(-1, 4)
    # alternate the split diagonal by parity to avoid directional bias
    par = (quads[:, 0, :].sum(1) & 1).astype(bool)
    t1 = np.where(par[:, None], q[:, [0, 1, 2]], q[:, [0, 1, 3]])
    t2 = np.where(par[:, None], q[:, [0, 2, 3]], q[:, [1, 2, 3]])
    tris = np.concatenate([t1, t2], 0)
    return verts, tris, len(q)


def _edges(tris):
    e = np.concatenate([tris[:, [0, 1]], tris[:, [1, 2]], tris[:, [2, 0]]], 0)
    e = np.sort(e, 1)
    return np.unique(e, axis=0)


def _project(verts, caps, iters=6, blend=0.012):
    v = verts.copy()
    eps = 1e-4
    for _ in range(iters):
        f = _implicit(v, caps, blend)
        g = np.zeros_like(v)
        for a in range(3):
            d = np.zeros(3); d[a] = eps
            g[:, a] = (_implicit(v + d, caps, blend) - _implicit(v - d, caps, blend)) / (2 * eps)
        gn = (g * g).sum(1, keepdims=True) + 1e-12
        step = f[:, None] * g / gn
        n = np.linalg.norm(step, axis=1, keepdims=True)
        step *= np.minimum(1.0, 0.02 / np.maximum(n, 1e-12))
        v -= step
    return v


def _laplacian(verts, tris, lam=0.5, iters=2):
    v = verts.copy()
    e = _edges(tris)
    for _ in range(iters):
        acc = np.zeros_like(v); cnt = np.zeros(len(v))
        np.add.at(acc, e[:, 0], v[e[:, 1]]); np.add.at(acc, e[:, 1], v[e[:, 0]])
        np.add.at(cnt, e[:, 0], 1.0); np.add.at(cnt, e[:, 1], 1.0)
        v = v + lam * (acc / cnt[:, None] - v)
    return v


def _split_longest_edges(verts, tris, nsplit):
    """Each split adds 1 vertex and 2 faces, preserving F = 2V - 4 on the closed mesh."""
    verts = [tuple(x) for x in verts]
    tris = [list(t) for t in tris]
    for _ in range(nsplit):
        V = np.asarray(verts); T = np.asarray(tris)
        e = _edges(T)
        ln = np.linalg.norm(V[e[:, 0]] - V[e[:, 1]], axis=1)
        k = int(np.lexsort((e[:, 1], e[:, 0], -np.round(ln, 9)))[0])
        a, b = int(e[k, 0]), int(e[k, 1])
        m = len(verts)
        verts.append(tuple(0.5 * (V[a] + V[b])))
        new = []
        for ti, t in enumerate(tris):
            if a in t and b in t:
                ia, ib = t.index(a), t.index(b)
                t1 = list(t); t1[ib] = m
                t2 = list(t); t2[ia] = m
                tris[ti] = t1
                new.append(t2)
        tris.extend(new)
    return np.asarray(verts, float), np.asarray(tris, np.int64)


def _bone_segments(joints):
    """Per joint: list of segments the joint 'owns' for skinning-weight distances."""
    children = [[] for _ in range(NUM_JOINTS)]
    for j in range(1, NUM_JOINTS):
        children[PARENT[j]].append(j)
    segs = []
    for j in range(NUM_JOINTS):
        s = []
        if children[j]:
            for c in children[j]:
                s.append((joints[j], joints[c]))
        else:
            d = joints[j] - joints[PARENT[j]]
            d = d / np.linalg.norm(d)
            ext = {10: 0.10, 11: 0.10, 15: 0.12, 22: 0.07, 23: 0.07}.get(j, 0.08)
            s.append((joints[j], joints[j] + ext * d))
        segs.append(s)
    return segs


def _seg_dist(p, a, b):
    ba = b - a
    t = np.clip(((p - a) @ ba) / (ba @ ba), 0.0, 1.0)
    c = a + t[:, None] * ba
    return np.linalg.norm(p - c, axis=1), c


def build_model(seed: int = 0):
    """Build the synthetic model. Returns a dict of numpy arrays under SMPL npz key names plus the
    GMM prior under `prior_*`. Deterministic for a given seed."""
    rng = np.random.default_rng(seed)
    caps = _capsules()

    # --- closed genus-0 mesh with exactly NUM_VERTS vertices ---------------------------------
    target_q = NUM_VERTS - 2
    best = None
    # voxel sizes pre-scanned so the boundary has (close to, never more than) V-2 quads; the first one hits
    # 6888 exactly on this body, the rest are fall-backs finished by edge splits.
    for h in (0.01778, 0.01786, 0.01788, 0.01790, 0.01792, 0.01800):
        occ, origin, hh = _voxel_body(float(h), caps)
        verts, tris, nq = _boundary_mesh(occ, origin, hh)
        if nq <= target_q and (best is None or nq > best[3]):
            best = (verts, tris, hh, nq)
        if nq == target_q:
            break
    verts, tris, h, nq = best
    assert len(verts) == nq + 2, "voxel boundary is not a genus-0 closed surface"
    verts = _project(verts, caps)
    verts = _laplacian(verts, tris, 0.5, 2)
    verts = _project(verts, caps, iters=3)
    verts, tris = _split_longest_edges(verts, tris, target_q - nq)
    assert verts.shape == (NUM_VERTS, 3) and tris.shape == (NUM_FACES, 3)
    e = _edges(tris)
    assert len(verts) - len(e) + len(tris) == 2
    # orientation: signed volume must be positive (outward CCW)
    vol = np.einsum("ij,ij->i", verts[tris[:, 0]], np.cross(verts[tris[:, 1]], verts[tris[:, 2]])).sum() / 6.0
    assert vol > 0.0

    shift = np.array([0.0, -0.22, 0.0])          # SMPL's origin sits near the chest, pelvis below it
    verts = verts + shift
    joints_plan = _JOINTS + shift

    # --- skinning weights: <=4 joints per vertex ----------------------------------------------
    segs = _bone_segments(joints_plan)
    dist = np.empty((NUM_VERTS, NUM_JOINTS))
    for j in range(NUM_JOINTS):
        dj = None
        for a, b in segs[j]:
            d, _ = _seg_dist(verts, a, b)
            dj = d if dj is None else np.minimum(dj, d)
        dist[:, j] = dj
    # keep limbs from grabbing the opposite side / torso from grabbing far limbs: sharp falloff
    raw = 1.0 / (dist ** 4 + 1e-7)
    order = np.argsort(-raw, axis=1, kind="stable")[:, :4]
    wsel = np.take_along_axis(raw, order, 1)
    wsel = wsel / wsel.sum(1, keepdims=True)
    wsel[wsel < 0.02] = 0.0
    wsel = wsel / wsel.sum(1, keepdims=True)
    weights = np.zeros((NUM_VERTS, NUM_JOINTS))
    np.put_along_axis(weights, order, wsel, 1)

    # --- sparse joint regressor: centroid of the surface ring around each planned joint --------
    jreg = np.zeros((NUM_JOINTS, NUM_VERTS))
    for j in range(NUM_JOINTS):
        a, b = segs[j][0]
        u = (b - a) / np.linalg.norm(b - a)
        rel = verts - joints_plan[j]
        d = np.linalg.norm(rel, axis=1)
        slab = np.abs(rel @ u) < 0.016
        near_r = np.sort(d[slab])[0] if slab.any() else d.min()
        sel = np.nonzero(slab & (d < 2.2 * near_r + 0.02))[0]
        if len(sel) < 8:
            sel = np.argsort(d, kind="stable")[:32]
        sel = sel[np.argsort(d[sel], kind="stable")][:64]
        jreg[j, np.sort(sel)] = 1.0 / len(sel)

    # --- shape keys -----------------------------------------------------------------------------
    shapedirs = np.zeros((NUM_VERTS, 3, NUM_SHAPE))
    pel = joints_plan[0]
    shapedirs[:, :, 0] = 0.03 * (verts - pel)                        # overall size
    near = np.argmin(dist, axis=1)
    radial = np.zeros_like(verts)
    for j in range(NUM_JOINTS):
        m = near == j
        if not m.any():
            continue
        a, b = segs[j][0]
        _, c = _seg_dist(verts[m], a, b)
        radial[m] = verts[m] - c
    shapedirs[:, :, 1] = 0.12 * radial                               # girth
    shapedirs[:, 1, 2] = 0.025 * (verts[:, 1] - pel[1])              # height
    shapedirs[:, 0, 3] = 0.03 * verts[:, 0] * (verts[:, 1] > joints_plan[9][1] - 0.1)  # shoulder width
    for k in range(4, NUM_SHAPE):
        B = rng.normal(size=(3, 3)) * (2.0 + 0.8 * k)
        ph = rng.uniform(0, 2 * np.pi, size=3)
        amp = 0.012 / (1.0 + 0.35 * (k - 4))
        shapedirs[:, :, k] = amp * np.sin(verts @ B.T + ph)

    # --- GMM pose prior (pose_prior.txt layout) ---------------------------------------------------
    nc, nd = 8, 3 * (NUM_JOINTS - 1)
    pw = rng.dirichlet(np.full(nc, 4.0))
    pmean = rng.normal(size=(nc, nd)) * 0.12
    pcov = np.empty((nc, nd, nd))
    for c in range(nc):
        A = rng.normal(size=(nd, nd)) * 0.025
        sig = rng.uniform(0.08, 0.25, size=nd)
        pcov[c] = A @ A.T + np.diag(sig ** 2)
        pcov[c] = 0.5 * (pcov[c] + pcov[c].T)

    kintree = np.zeros((2, NUM_JOINTS), dtype=np.int64)
    kintree[0] = PARENT
    kintree[0, 0] = -1
    kintree[1] = np.arange(NUM_JOINTS)
    return {
        "v_template": np.ascontiguousarray(verts),                    # (V,3)
        "f": np.ascontiguousarray(tris.astype(np.int32)),             # (F,3)
        "kintree_table": kintree,                                     # (2,J) row 0 = parent
        "J_regressor": jreg,                                          # (J,V)
        "weights": weights,                                           # (V,J)
        "shapedirs": shapedirs,                                       # (V,3,K)
        "prior_weight": pw, "prior_mean": pmean, "prior_cov": pcov,
    }


_CACHE = {}


def load_model(seed: int = 0):
    """Cached build_model(); also memoised on disk under $AVT_CACHE_DIR (default /tmp/avt_cache)."""
    if seed in _CACHE:
        return _CACHE[seed]
    cdir = os.environ.get("AVT_CACHE_DIR", "/tmp/avt_cache")
    path = os.path.join(cdir, f"synth_model_seed{seed}_v1.npz")
    m = None
    if os.path.exists(path):
        try:
            with np.load(path) as z:
                m = {k: z[k] for k in z.files}
        except Exception:
            m = None
    if m is None:
        m = build_model(seed)
        try:
            os.makedirs(cdir, exist_ok=True)
            tmp = path + f".{os.getpid()}.tmp.npz"
            np.savez(tmp, **m)
            os.replace(tmp, path)
        except OSError:
            pass
    _CACHE[seed] = m
    return m


def write_pose_prior_txt(model, path):
    """Emit the GMM in the reference's text layout (GaussianMixture.cpp:20-58)."""
    nc, nd = model["prior_mean"].shape
    with open(path, "w") as f:
        f.write(f"{nc} {nd}\n")
        f.write(" ".join(repr(float(x)) for x in model["prior_weight"]) + "\n")
        for c in range(nc):
            f.write(" ".join(repr(float(x)) for x in model["prior_mean"][c]) + "\n")
        for c in range(nc):
            for r in range(nd):
                f.write(" ".join(repr(float(x)) for x in model["prior_cov"][c, r]) + "\n")


# ------------------------------------------------------------------------------------------------
# ground-truth posing (harness-only numpy LBS, used to synthesise depth clouds)
# ------------------------------------------------------------------------------------------------
def rodrigues(aa):
    aa = np.asarray(aa, float)
    th = np.linalg.norm(aa)
    if th < 1e-12:
        return np.eye(3)
    k = aa / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


def pose_vertices(model, w, p, R):
    """Numpy LBS with the reference's conventions (root placed at p). R: (J,3,3). Returns (V,3)."""
    v = model["v_template"] + model["shapedirs"] @ w
    Jr = model["J_regressor"]
    jp = Jr @ model["v_template"] + np.einsum("jv,vck,k->jc", Jr, model["shapedirs"], w)
    J = len(PARENT)
    A = np.zeros((J, 3, 4))
    A[0, :, :3] = R[0]; A[0, :, 3] = p
    for i in range(1, J):
        pa = PARENT[i]
        A[i, :, :3] = A[pa, :, :3] @ R[i]
        A[i, :, 3] = A[pa, :, 3] + A[pa, :, :3] @ (jp[i] - jp[pa])
    T = A.copy()
    for i in range(J):
        T[i, :, 3] = A[i, :, 3] - A[i, :, :3] @ jp[i]
    PT = np.einsum("vj,jab->vab", model["weights"], T)
    return np.einsum("vab,vb->va", PT[:, :, :3], v) + PT[:, :, 3]


def sample_ground_truth(model, seed, z_range=(2.3, 2.5), use_gmm=True):
    """Ground-truth (w, p, R[J,3,3]) for frame `seed`, following Avatar::randomize (Avatar.cpp:77-126)."""
    rng = np.random.default_rng(1000003 * (seed + 1))
    w = rng.normal(size=NUM_SHAPE)
    R = np.tile(np.eye(3), (NUM_JOINTS, 1, 1))
    if use_gmm:
        c = int(rng.choice(len(model["prior_weight"]), p=model["prior_weight"]))
        L = np.linalg.cholesky(model["prior_cov"][c])
        x = model["prior_mean"][c] + L @ rng.normal(size=L.shape[0])
    else:
        x = rng.normal(size=3 * (NUM_JOINTS - 1)) * 0.2
    for i in range(1, NUM_JOINTS):
        R[i] = rodrigues(x[3 * (i - 1):3 * i])
    p = np.array([rng.uniform(-0.6, 0.6), rng.uniform(-0.25, 0.25), rng.uniform(*z_range)])
    ang_up = np.pi + rng.uniform(-np.pi / 3, np.pi / 3)
    th = rng.uniform(0, 2 * np.pi); ph = rng.uniform(-np.pi / 2, np.pi / 2)
    axis = np.array([np.sin(ph) * np.cos(th), np.cos(ph), np.sin(ph) * np.sin(th)])
    R[0] = rodrigues(axis * rng.normal(0, 0.2)) @ rodrigues(np.array([0, ang_up, 0]))
    return w, p, R


def perturb_start(w, p, R, seed):
    """Tracking start state of the disabled validator (optim.cpp:131-142): N(0,0.1) rad per joint about a
    random axis, shape reset with w0 = -2.5."""
    rng = np.random.default_rng(7919 * (seed + 1))
    R2 = R.copy()
    for i in range(NUM_JOINTS):
        th = rng.uniform(0, 2 * np.pi); ph = rng.uniform(-np.pi / 2, np.pi / 2)
        axis = np.array([np.sin(ph) * np.cos(th), np.cos(ph), np.sin(ph) * np.sin(th)])
        R2[i] = R[i] @ rodrigues(axis * rng.normal(0, 0.1))
    w2 = np.zeros_like(w); w2[0] = -2.5
    return w2, p.copy(), R2


# ------------------------------------------------------------------------------------------------
# synthetic depth frames (smplsynth / optim.cpp counterpart)
# ------------------------------------------------------------------------------------------------
K4A_INTRIN = dict(fx=606.438, fy=606.351, cx=637.294, cy=366.992, width=1280, height=720)  # smplsynth.cpp:245-249


def main_joint(model):
    """assignedJoints[v][0].second: the largest (weight, joint) pair, ties to the larger joint id
    (std::greater<pair<double,int>>, AvatarModel.cpp:92-94)."""
    W = np.asarray(model["weights"])
    J = W.shape[1]
    key = W + 0.0
    best = np.zeros(W.shape[0], np.int64)
    bw = np.full(W.shape[0], -1.0)
    for j in range(J):
        m = key[:, j] >= bw
        m &= key[:, j] > 1e-12
        best[m] = j
        bw[m] = key[m, j]
    return best.astype(np.int32)


_synth_lib = None


def _render_lib():
    global _synth_lib
    if _synth_lib is None:
        import ctypes as C
        import subprocess
        here = os.path.dirname(os.path.abspath(__file__))
        so = os.path.join(here, "csrc", "libavt_synth.so")
        src = os.path.join(here, "csrc", "synth_render.cpp")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, src])
        _synth_lib = C.CDLL(so)
        _synth_lib.avt_synth_render_cloud.restype = C.c_int
    return _synth_lib


def render_cloud(model, verts, part_map, res_scale=1):
    """Depth-render the posed vertices `verts` (V,3) with the K4A intrinsics (x res_scale) and back-project
    every foreground pixel.  Returns (data (N,3) float64, labels (N,) int32)."""
    import ctypes as C
    lib = _render_lib()
    V = verts.shape[0]
    mesh = np.ascontiguousarray(model["f"], np.int32)
    vp = np.ascontiguousarray(np.asarray(part_map, np.int32)[main_joint(model)])
    cloud = np.ascontiguousarray(verts, np.float64)
    k = K4A_INTRIN
    W, H = k["width"] * res_scale, k["height"] * res_scale
    cap = 400000 * res_scale * res_scale
    xyz = np.empty((cap, 3), np.float64); lab = np.empty(cap, np.int32)
    dp = C.POINTER(C.c_double); ip = C.POINTER(C.c_int)
    n = lib.avt_synth_render_cloud(
        C.c_int(V), C.c_int(mesh.shape[0]), cloud.ctypes.data_as(dp), mesh.ctypes.data_as(ip), vp.ctypes.data_as(ip),
        C.c_double(k["fx"] * res_scale), C.c_double(k["fy"] * res_scale), C.c_double(k["cx"] * res_scale),
        C.c_double(k["cy"] * res_scale), C.c_int(W), C.c_int(H), C.c_int(cap), xyz.ctypes.data_as(dp),
        lab.ctypes.data_as(ip))
    assert n <= cap
    return xyz[:n].copy(), lab[:n].copy()


def render_images(model, verts, part_map):
    """XYZ map (H,W,3) float32 (camera coordinates, y down) + part mask (H,W) uint8 (255 = background) of the posed
    vertices, the inputs of the reference's tracker loop (demo.cpp:215-250)."""
    import ctypes as C
    lib = _render_lib()
    mesh = np.ascontiguousarray(model["f"], np.int32)
    vp = np.ascontiguousarray(np.asarray(part_map, np.int32)[main_joint(model)])
    cloud = np.ascontiguousarray(verts, np.float64)
    k = K4A_INTRIN
    W, H = k["width"], k["height"]
    xyz = np.empty((H, W, 3), np.float32); mask = np.empty((H, W), np.uint8)
    dp = C.POINTER(C.c_double); ip = C.POINTER(C.c_int)
    lib.avt_synth_render_images.restype = C.c_int
    n = lib.avt_synth_render_images(
        C.c_int(verts.shape[0]), C.c_int(mesh.shape[0]), cloud.ctypes.data_as(dp), mesh.ctypes.data_as(ip), vp.ctypes.data_as(ip),
        C.c_double(k["fx"]), C.c_double(k["fy"]), C.c_double(k["cx"]), C.c_double(k["cy"]), C.c_int(W), C.c_int(H),
        xyz.ctypes.data_as(C.POINTER(C.c_float)), mask.ctypes.data_as(C.POINTER(C.c_ubyte)))
    return xyz, mask, n


def identity_part_map(J=NUM_JOINTS):
    """numParts = J, partMap = identity: the documented benchmark choice."""
    return np.arange(J, dtype=np.int32)


def make_frame(model, seed, dense=False, part_map=None):
    """One synthetic tracking frame: ground truth, its depth cloud + labels, and the perturbed start state."""
    part_map = identity_part_map() if part_map is None else part_map
    w, p, R = sample_ground_truth(model, seed)
    verts = pose_vertices(model, w, p, R)
    data, labels = render_cloud(model, verts, part_map, res_scale=2 if dense else 1)
    w0, p0, R0 = perturb_start(w, p, R, seed)
    return dict(data=data, labels=labels, gt=(w, p, R), start=(w0, p0, R0), gt_verts=verts)
