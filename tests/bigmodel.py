"""Synthetic skeletons beyond 24 joints, grown from the 24-joint synthetic model: five finger chains hang off each hand joint
(3,3,3,3,2 joints: 52 joints, SMPL-H sized, P = 3 + 3*52 + 10 = 169; 3,3,3,3,3 plus one jaw-like joint on the head: 55
joints, SMPL-X sized, P = 178); hand (head) vertices are re-weighted onto one new segment each (<= 4 weights per vertex,
<= 16 ancestors per vertex kept), joint regressor rows and the GMM prior are extended to match.  Test helper only."""
import numpy as np

from avatar_amd import synth

CHAINS = (3, 3, 3, 3, 2)


def extend_model(smpl, joints=52):
    assert joints in (52, 55)
    chains = CHAINS if joints == 52 else (3, 3, 3, 3, 3)
    J0 = 24
    parent = list(synth.PARENT)
    W0 = np.asarray(smpl["weights"], np.float64)
    V = W0.shape[0]
    mj = synth.main_joint(smpl)
    vt = np.asarray(smpl["v_template"], np.float64)
    jpos = np.asarray(smpl["J_regressor"]) @ vt
    newW = {}           # vertex -> new joint
    members = {}        # new joint -> vertices
    for hand in (22, 23):
        verts = np.flatnonzero(mj == hand)
        # fingers = five contiguous groups along the axis of largest extent; segments by distance from the hand joint
        ax = int(np.argmax(np.ptp(vt[verts], 0)))
        verts = verts[np.argsort(vt[verts, ax], kind="stable")]
        groups = np.array_split(verts, len(chains))
        for g, nseg in zip(groups, chains):
            g = g[np.argsort(np.linalg.norm(vt[g] - jpos[hand], axis=1), kind="stable")]
            prev = hand
            for seg in np.array_split(g, nseg):
                j = len(parent)
                parent.append(prev)
                prev = j
                members[j] = seg
                for v in seg:
                    newW[int(v)] = j
    if joints == 55:      # a jaw-like joint on the head: the lower third of the head's vertices
        verts = np.flatnonzero(mj == 15)
        verts = verts[np.argsort(vt[verts, 1], kind="stable")][: max(8, len(verts) // 3)]
        j = len(parent)
        parent.append(15)
        members[j] = verts
        for v in verts:
            newW[int(v)] = j
    J = len(parent)
    assert J == joints
    W = np.zeros((V, J))
    W[:, :J0] = W0
    for v, j in newW.items():
        row = W[v, :J0].copy()
        nz = np.flatnonzero(row > 1e-12)
        if len(nz) > 3:                              # keep the three largest: four weights with the new joint
            drop = nz[np.argsort(row[nz])[:len(nz) - 3]]
            row[drop] = 0.0
        row *= 0.5 / row.sum()
        W[v, :J0] = row
        W[v, j] = 0.5
    Jr = np.zeros((J, V))
    Jr[:J0] = np.asarray(smpl["J_regressor"], np.float64)
    for j, seg in members.items():
        Jr[j, seg] = 1.0 / len(seg)
    kin = np.zeros((2, J), np.int64)
    kin[0] = parent; kin[0, 0] = -1
    kin[1] = np.arange(J)
    nd0, nd = 3 * (J0 - 1), 3 * (J - 1)
    C = len(smpl["prior_weight"])
    mean = np.zeros((C, nd)); mean[:, :nd0] = smpl["prior_mean"]
    cov = np.zeros((C, nd, nd))
    for c in range(C):
        cov[c, :nd0, :nd0] = smpl["prior_cov"][c]
        cov[c, nd0:, nd0:] = 0.04 * np.eye(nd - nd0)
    m = dict(smpl)
    m.update(weights=W, J_regressor=Jr, kintree_table=kin, prior_mean=mean, prior_cov=cov)
    return m


def make_frame(model52, omodel52, smpl24, seed):
    """Ground truth, rendered depth cloud + labels (identity parts) and a perturbed start for the extended model."""
    J = np.asarray(model52["kintree_table"]).shape[1]
    rng = np.random.default_rng(4242 + seed)
    w, p, R24 = synth.sample_ground_truth(smpl24, seed)
    R = np.tile(np.eye(3), (J, 1, 1)); R[:24] = R24
    for j in range(24, J):
        R[j] = synth.rodrigues(rng.normal(0, 0.15, 3))
    verts, _, _ = omodel52.update(w, p, R)
    pm = np.arange(J, dtype=np.int32)
    data, labels = synth.render_cloud(model52, verts, pm)
    w0, p0, R0_24 = synth.perturb_start(w, p, R24, seed)
    R0 = R.copy(); R0[:24] = R0_24
    for j in range(24, J):
        R0[j] = R[j] @ synth.rodrigues(rng.normal(0, 0.1, 3))
    return dict(data=data, labels=labels, gt=(w, p, R), start=(w0, p0, R0), gt_verts=verts, part_map=pm)
