"""CPU tests (-m "not gpu"): the oracle against golden vectors and known answers, host logic, the C-ABI library's
exported symbols, and the N>1 sharding logic under gloo.  No GPU compute here."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from avatar_amd import capi, synth
from avatar_amd.capi import Options

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


# ------------------------------------------------------------------------------------------------ model harness
def test_synthetic_model_is_smpl_shaped(smpl):
    v, f, W = smpl["v_template"], smpl["f"], smpl["weights"]
    assert v.shape == (6890, 3) and f.shape == (13776, 3) and W.shape == (6890, 24)
    assert smpl["shapedirs"].shape == (6890, 3, 10)
    e = np.unique(np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]), 1), axis=0)
    assert len(v) - len(e) + len(f) == 2                       # closed genus-0 surface
    vol = np.einsum("ij,ij->i", v[f[:, 0]], np.cross(v[f[:, 1]], v[f[:, 2]])).sum() / 6
    assert vol > 0                                             # outward orientation
    assert ((W > 0).sum(1) <= 4).all() and np.allclose(W.sum(1), 1.0)
    assert list(np.asarray(smpl["kintree_table"])[0]) == [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19,
                                                          20, 21]
    assert np.linalg.eigvalsh(smpl["prior_cov"]).min() > 0
    m2 = synth.build_model(0)
    assert all(np.array_equal(smpl[k], m2[k]) for k in smpl)   # deterministic


# ------------------------------------------------------------------------------------------------ NN pinned by nanoflann
def test_oracle_nn_matches_nanoflann_golden(omodel):
    """tests/golden/nn_golden.npz holds the outputs of the reference's own nanoflann (make_nn_golden.py)."""
    z = np.load(os.path.join(HERE, "golden", "nn_golden.npz"))
    for k in range(int(z["ncase"])):
        got = omodel.nn(z[f"part_map_{k}"], int(z[f"num_parts_{k}"]), z[f"cloud_{k}"], z[f"vis_{k}"], z[f"data_{k}"], z[f"labels_{k}"])
        assert np.array_equal(got, z[f"idx_{k}"]), f"case {k}"
        idx = z[f"idx_{k}"]
        # goldens carry no exact ties: the winner is strictly closer than the runner-up of the same part
        d = z[f"dist_{k}"]
        assert (d[idx >= 0] >= 0).all()
    # case 1 has a part with data points but no visible model point -> dropped (-1), AvatarOptimizer.cpp:899
    assert (z["idx_1"] == -1).any()


def test_oracle_nn_live_against_reference_build(smpl, omodel):
    from oracle import oracle as orc
    if not orc.have_reference_nn():
        pytest.skip("oracle/_ref not built (reference tree absent)")
    fr = synth.make_frame(smpl, 11)
    w0, p0, R0 = fr["start"]
    cloud, _, _ = omodel.update(w0, p0, R0)
    vis = omodel.visibility(cloud)
    pm = synth.identity_part_map()
    ref, _ = orc.reference_nn(pm[omodel.main_joint()].astype(np.int32), cloud, vis, fr["data"], fr["labels"], 24)
    assert np.array_equal(ref, omodel.nn(pm, 24, cloud, vis, fr["data"], fr["labels"]))


def test_oracle_nn_full_size_against_nanoflann_goldens(smpl, omodel):
    """The oracle's ordered brute-force search against the reference's nanoflann KD-tree at FULL size (38 k, 125 k dense and
    a coarse 6-part map): inputs regenerated from seeds, reference index arrays committed (tests/golden/nn_golden_full.npz,
    made by tests/golden/make_nn_golden_full.py in the build container)."""
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_nn_golden_full as mk
    z = np.load(os.path.join(HERE, "golden", "nn_golden_full.npz"))
    assert int(z["ncase"]) == len(mk.CASES)
    for k, c in enumerate(mk.CASES):
        pm, npart, cloud, vis, data, labels = mk.case_inputs(smpl, omodel, c)
        assert len(labels) == int(z[f"n_{k}"])
        assert np.array_equal(omodel.nn(pm, npart, cloud, vis, data, labels), z[f"idx_{k}"]), k


# ------------------------------------------------------------------------------------------------ LBS known answers
def test_lbs_known_answers(smpl, omodel):
    J0 = omodel.joint_regression()[0][0]
    I = np.tile(np.eye(3), (24, 1, 1))
    p = np.array([0.1, -0.3, 2.0])
    # identity pose, zero shape: cloud = base - J0 + p (Avatar.cpp:47-49,59-64)
    c, jp, _ = omodel.update(np.zeros(10), p, I)
    assert np.abs(c - (smpl["v_template"] - J0 + p)).max() < 1e-14
    # root-only rotation: rigid transform about J0
    R = I.copy(); R[0] = synth.rodrigues([0.3, -1.1, 0.4])
    c2, _, _ = omodel.update(np.zeros(10), p, R)
    assert np.abs(c2 - ((smpl["v_template"] - J0) @ R[0].T + p)).max() < 1e-13
    # translation equivariance (weights sum to one)
    w = np.linspace(-1, 1, 10)
    Rr = np.array([synth.rodrigues(0.2 * np.sin(np.arange(3) + j)) for j in range(24)])
    a, _, _ = omodel.update(w, p, Rr)
    b, _, _ = omodel.update(w, p + [0.5, 0.25, -1.0], Rr)
    assert np.abs(b - a - [0.5, 0.25, -1.0]).max() < 1e-13
    # independent numpy LBS (harness) agrees
    assert np.abs(a - synth.pose_vertices(smpl, w, p, Rr)).max() < 1e-13


def test_optimiser_forward_model_equals_update(smpl, omodel):
    """updateData's x_m (AvatarOptimizer.cpp:507-514) must reproduce Avatar::update()'s cloud."""
    from oracle import oracle as orc
    w, p, R = synth.sample_ground_truth(smpl, 5)
    c, _, _ = omodel.update(w, p, R)
    q = orc.rot_to_quat(R)
    assert np.abs(orc.quat_to_rot(q) - R).max() < 1e-14
    assert np.abs(omodel.points(p, q, w) - c).max() < 1e-13


def test_jacobian_vs_finite_differences(smpl, omodel):
    """Analytic blocks (AvatarOptimizer.cpp:524-580) against central differences through the reference's own
    retraction (q <- dq*q with rotation angle 2|delta|, :123-143)."""
    from oracle import oracle as orc
    w, p, R = synth.sample_ground_truth(smpl, 2)
    q = orc.rot_to_quat(R)
    rng = np.random.default_rng(0)
    for pt in rng.choice(6890, 6, replace=False):
        x, Jd = omodel.point_jacobian(p, q, w, int(pt))
        eps = 1e-6
        Jfd = np.zeros_like(Jd)
        for a in range(omodel.P):
            dlt = np.zeros(omodel.P); dlt[a] = eps
            xp = omodel.points(*[omodel.retract(p, q, w, dlt)[i] for i in (0, 1, 2)])[pt]
            xm = omodel.points(*[omodel.retract(p, q, w, -dlt)[i] for i in (0, 1, 2)])[pt]
            Jfd[:, a] = (xp - xm) / (2 * eps)
        assert np.abs(Jd - Jfd).max() < 5e-9
        anc = omodel.ancestors(int(pt))
        cols = set(range(3)) | {3 + 3 * j + c for j in anc for c in range(3)} | set(range(75, 85))
        assert np.abs(Jd[:, [c for c in range(85) if c not in cols]]).max() == 0.0


def test_independent_forward_model(smpl, omodel):
    """The reference's own check design (TEST_COMPARE_AUTO_DIFF, AvatarOptimizer.cpp:742-818): rotate-and-translate
    along each assigned joint's chain to the root."""
    from oracle import oracle as orc
    w, p, R = synth.sample_ground_truth(smpl, 9)
    q = orc.rot_to_quat(R)
    v = smpl["v_template"] + smpl["shapedirs"] @ w
    ijp, jsr = omodel.joint_regression()
    jp = ijp + (jsr @ w).reshape(24, 3)
    off = jp[0].copy(); v = v - off; jp = jp - off
    Rq = orc.quat_to_rot(q)
    par = synth.PARENT
    mine = omodel.points(p, q, w)
    W = smpl["weights"]
    for pt in (0, 777, 3456, 6889):
        acc = np.zeros(3)
        for k in np.nonzero(W[pt] > 1e-12)[0]:
            vec = v[pt] - jp[k]
            j = k
            while j != -1:
                vec = Rq[j] @ vec
                if j:
                    vec = vec + jp[j] - jp[par[j]]
                j = par[j]
            acc += W[pt, k] * vec
        assert np.abs(acc + p - mine[pt]).max() < 1e-13


# ------------------------------------------------------------------------------------------------ priors
def test_pose_prior_vs_dense_numpy(smpl, omodel):
    from oracle import oracle as orc
    w, p, R = synth.sample_ground_truth(smpl, 4)
    q = orc.rot_to_quat(R)
    comp, x, res = omodel.pose_prior_residual(q)
    n = 69
    # smplParams: axis-angle of the non-root rotations
    for j in range(1, 24):
        assert np.abs(synth.rodrigues(x[3 * (j - 1):3 * j]) - R[j]).max() < 1e-12
    cov, mu, wt = smpl["prior_cov"], smpl["prior_mean"], smpl["prior_weight"]
    dets = np.array([np.sqrt(np.linalg.det(c)) for c in cov])
    clog = np.log(wt) - n * 0.5 * np.log(2 * np.pi) - np.log(dets) + np.log(dets.min())
    score = np.array([0.5 * (x - mu[c]) @ np.linalg.solve(cov[c], x - mu[c]) - clog[c] for c in range(8)])
    assert comp == int(np.argmin(score))
    assert abs(res[:n] @ res[:n] - 0.5 * (x - mu[comp]) @ np.linalg.solve(cov[comp], x - mu[comp])) < 1e-9
    assert abs(res[n] - np.sqrt(-clog[comp])) < 1e-12
    L, cl = omodel.prior_factors(comp)
    assert np.abs(L @ L.T - np.linalg.inv(cov[comp])).max() < 1e-7 * np.abs(np.linalg.inv(cov[comp])).max()
    assert abs(cl - clog[comp]) < 1e-10


def test_gradient_vs_finite_differences_of_cost(smpl, omodel, frame0):
    """J^T r of the data + shape terms vs finite differences of the cost through the retraction.  (The pose-prior
    Jacobian deliberately drops d(axis-angle)/d(delta) like the reference, AvatarOptimizer.cpp:675-687, so it is
    excluded here: beta_pose = 0.)"""
    from oracle import oracle as orc
    fr = frame0
    w0, p0, R0 = fr["start"]
    q0 = orc.rot_to_quat(R0)
    pm = synth.identity_part_map()
    cloud, _, _ = omodel.update(w0, p0, R0)
    corr = omodel.nn(pm, 24, cloud, omodel.visibility(cloud), fr["data"], fr["labels"])
    sel = np.arange(0, len(corr), 40)
    cost, g, H, _ = omodel.evaluate(p0, q0, w0, corr[sel], fr["data"][sel], 0.0, 0.12)
    cost_a, g_a, H_a, _ = omodel.evaluate(p0, q0, w0, corr[sel], fr["data"][sel], 0.0, 0.12, aggregate=1)
    assert abs(cost - cost_a) < 1e-10 * cost and np.abs(g - g_a).max() < 1e-8 * np.abs(g).max()
    assert np.abs(H - H_a).max() < 1e-9 * np.abs(H).max()
    eps = 1e-6
    for a in (0, 2, 3, 10, 41, 60, 74, 75, 84):
        d = np.zeros(85); d[a] = eps
        cp = omodel.evaluate(*omodel.retract(p0, q0, w0, d), corr[sel], fr["data"][sel], 0.0, 0.12)[0]
        cm = omodel.evaluate(*omodel.retract(p0, q0, w0, -d), corr[sel], fr["data"][sel], 0.0, 0.12)[0]
        assert abs((cp - cm) / (2 * eps) - g[a]) < 1e-5 * max(1.0, abs(g[a]))


# ------------------------------------------------------------------------------------------------ fit behaviour
def test_lm_schedule_descends_and_recovers_ground_truth(smpl, omodel, frame0):
    from oracle import oracle as orc
    fr = frame0
    w0, p0, R0 = fr["start"]
    opt = Options.demo(icp_iters=4)
    res = omodel.optimize(synth.identity_part_map(), 24, fr["data"], fr["labels"], opt, p0, orc.rot_to_quat(R0), w0, aggregate=1)
    tc = res["trace_cost"].reshape(4, 11)
    assert (np.diff(tc, axis=1) <= 1e-12).all()                       # monotone within an ICP iteration
    start = omodel.update(w0, p0, R0)[0]
    e0 = np.abs(start - fr["gt_verts"]).mean(); e1 = np.abs(res["cloud"] - fr["gt_verts"]).mean()
    assert e1 < 0.5 * e0                                              # moves toward the generating body
    # literal per-residual-block accumulation gives the same iterates as the aggregated algebra
    res2 = omodel.optimize(synth.identity_part_map(), 24, fr["data"][::8], fr["labels"][::8], Options.demo(), p0,
                           orc.rot_to_quat(R0), w0, aggregate=0)
    res3 = omodel.optimize(synth.identity_part_map(), 24, fr["data"][::8], fr["labels"][::8], Options.demo(), p0,
                           orc.rot_to_quat(R0), w0, aggregate=1)
    assert np.abs(res2["cloud"] - res3["cloud"]).max() < 1e-8


def test_gain_ratio_schedule_accepts_more_and_ends_lower_on_the_bench_seeds(smpl, omodel):
    """avt_options.lm_policy = 1 with the demo constants (Options.GAIN_LM_UP: the first rejection's multiplier) on the 12 bench seeds: at
    least 0.8 of the GN iterations move the estimate (the fixed factors: under 0.6) and no seed ends at a higher objective."""
    from oracle import oracle as orc
    pm = synth.identity_part_map()
    acc = {0: [], 1: []}
    lower = 0
    for seed in range(12):
        fr = synth.make_frame(smpl, seed)
        w0, p0, R0 = fr["start"]
        r = [omodel.optimize(pm, 24, fr["data"], fr["labels"], Options.demo(lm_policy=pol), p0, orc.rot_to_quat(R0), w0, aggregate=1)["stats"] for pol in (0, 1)]
        for pol in (0, 1):
            acc[pol].append(r[pol].accepted_steps / r[pol].gn_iterations)
        lower += r[1].final_cost < r[0].final_cost
    # the gain-ratio schedule is the default (round 6); lm_policy=0 without an lm_up selects the fixed factors with the constant they were tuned with
    assert Options.demo().lm_policy == 1 and Options.demo().lm_up == Options.GAIN_LM_UP and Options.demo(lm_policy=1, lm_up=2.0).lm_up == 2.0
    assert Options.demo(lm_policy=0).lm_up == Options.FIXED_LM_UP == 4.0 and Options.demo(lm_policy=0, lm_up=8.0).lm_up == 8.0
    assert np.mean(acc[1]) >= 0.8 and np.mean(acc[0]) < 0.6, (np.mean(acc[0]), np.mean(acc[1]))
    assert lower == 12


def test_stopping_rule_ends_the_inner_iterations_of_an_icp_iteration(smpl, omodel):
    """avt_options.function_tolerance (the reference's options.function_tolerance = 1e-4, AvatarOptimizer.cpp:1333): an accepted step whose decrease
    is at most that fraction of the objective it started from is the last GN iteration of its ICP iteration - the next ICP iteration starts over.
    0 switches the rule off; a run with the rule on is the prefix of the run without it up to the first stop."""
    from oracle import oracle as orc
    pm = synth.identity_part_map()
    fr = synth.make_frame(smpl, 0)
    sel = np.arange(0, len(fr["labels"]), 6)
    data, labels = fr["data"][sel], fr["labels"][sel]
    w0, p0, R0 = fr["start"]
    q0 = orc.rot_to_quat(R0)
    off = omodel.optimize(pm, 24, data, labels, Options.counted(icp_iters=3), p0, q0, w0, aggregate=1)
    on = omodel.optimize(pm, 24, data, labels, Options.demo(icp_iters=3), p0, q0, w0, aggregate=1)
    assert Options.demo().function_tolerance == 1e-4 and Options.counted().function_tolerance == 0.0
    assert off["stats"].gn_iterations == 30 and (off["trace_acc"] != -2).all()
    acc, tc = on["trace_acc"].reshape(3, 10), on["trace_cost"].reshape(3, 11)
    assert on["stats"].gn_iterations == int((acc != -2).sum()) < 30      # this frame stops early in a later ICP iteration
    first = int(np.argmax((acc == -2).any(axis=1)))
    k = int(np.argmax(acc[first] == -2))      # iterations k .. 9 of ICP iteration `first` did not run; k - 1 is the step that met the rule
    assert k >= 1 and acc[first, k - 1] == 1 and (acc[first, k:] == -2).all()
    dec = tc[first, k - 1] - tc[first, k]
    assert 0.0 < dec <= 1e-4 * tc[first, k - 1] and (tc[first, k:] == tc[first, k]).all()
    # no accepted step in front of it met the rule
    for i in range(k - 1):
        if acc[first, i] == 1:
            assert tc[first, i] - tc[first, i + 1] > 1e-4 * tc[first, i]
    # identical to the run without the rule up to there
    n = first * 10 + k
    assert np.array_equal(on["trace_acc"][:n], off["trace_acc"][:n]) and np.array_equal(tc.reshape(-1)[:first * 11 + k + 1], off["trace_cost"][:first * 11 + k + 1])
    # a tolerance that every accepted step meets: one iteration per ICP iteration when the first step is accepted
    one = omodel.optimize(pm, 24, data, labels, Options.demo(icp_iters=2, function_tolerance=0.999), p0, q0, w0, aggregate=1)
    a1 = one["trace_acc"].reshape(2, 10)
    for i in range(2):
        kk = int(np.argmax(a1[i] == 1))
        assert (a1[i, :kk] != 1).all() and (a1[i, kk + 1:] == -2).all()


def test_lm_not_worse_than_scipy_bfgs(smpl, omodel, frame0):
    """Closest available stand-in for the reference's Ceres BFGS line search (AvatarOptimizer.cpp:1322-1326): scipy
    BFGS on the same cost/gradient with the same correspondences and 10 iterations must not reach a lower objective."""
    from scipy.optimize import minimize
    from oracle import oracle as orc
    fr = frame0
    w0, p0, R0 = fr["start"]
    q0 = orc.rot_to_quat(R0)
    pm = synth.identity_part_map()
    sel = np.arange(0, len(fr["labels"]), 10)
    data, labels = fr["data"][sel], fr["labels"][sel]
    opt = Options.demo()
    res = omodel.optimize(pm, 24, data, labels, opt, p0, q0, w0, aggregate=1)
    corr = res["corr"]

    def fun(dl):
        pp, qq, ww = omodel.retract(p0, q0, w0, dl)
        c, g0, _, _ = omodel.evaluate(pp, qq, ww, corr, data, opt.beta_pose, opt.beta_shape, aggregate=1)
        return c
    r = minimize(fun, np.zeros(85), method="BFGS", options={"maxiter": 10, "gtol": 1e-12})
    assert res["stats"].final_cost <= r.fun * (1 + 1e-6)


# ------------------------------------------------------------------------------------------------ C ABI
def test_abi_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "avt.h")).read() + open(os.path.join(ROOT, "include", "avt_shard.h")).read()
    declared = set(re.findall(r"\b(avt_[a-z_]+)\s*\(", hdr))
    assert declared == set(capi.EXPORTED_SYMBOLS), declared ^ set(capi.EXPORTED_SYMBOLS)
    assert os.path.exists(capi.LIB_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(capi.LIB_PATH)
    for s in declared:
        assert hasattr(lib, s), s


def test_struct_layouts_match_header():
    assert ctypes.sizeof(capi.Options) == 8 * 2 + 4 * 6 + 8 * 6
    assert ctypes.sizeof(capi.Stats) == 8 * 3 + 4 * 4
    assert ctypes.sizeof(capi.Profile) == (12 * capi.AVT_K_COUNT + 7) // 8 * 8          # doubles, ints, tail padding to 8


def test_python_option_defaults_are_the_c_defaults():
    """avt_options_default (avt_model.cpp) is the one place the defaults are set: the Python mirror agrees field by field, and
    avt_options_fixed_factors is Options.demo(lm_policy=0) with the reference's member defaults for the two prior weights."""
    lib = ctypes.CDLL(capi.LIB_PATH)
    c = Options()
    lib.avt_options_default(ctypes.byref(c))
    py = Options.reference_defaults()
    for name, _ in Options._fields_:
        assert getattr(c, name) == getattr(py, name), name
    assert c.lm_policy == Options.LM_GAIN_RATIO and c.lm_up == 16.0 and c.function_tolerance == 1e-4      # AvatarOptimizer.cpp:1333
    lib.avt_options_fixed_factors(ctypes.byref(c))
    fx = Options.reference_defaults(); fx.lm_policy = 0; fx.lm_up = Options.FIXED_LM_UP
    for name, _ in Options._fields_:
        assert getattr(c, name) == getattr(fx, name), name


def test_model_create_host_side(smpl, omodel):
    """avt_model_create is pure host code: derived data must equal the oracle's."""
    lib = capi.load_library()
    arr = capi.ModelArrays(smpl)
    desc = arr.desc()
    h = ctypes.c_void_p()
    assert lib.avt_model_create(ctypes.byref(desc), ctypes.byref(h)) == 0
    mj = np.empty(arr.V, np.int32)
    assert lib.avt_model_main_joint(h, capi.iptr(mj)) == 0
    assert np.array_equal(mj, omodel.main_joint())
    assert np.array_equal(mj, synth.main_joint(smpl))
    ijp = np.empty(72); jsr = np.empty(720)
    assert lib.avt_model_joint_regression(h, capi.dptr(ijp), capi.dptr(jsr)) == 0
    oi, oj = omodel.joint_regression()
    assert np.abs(ijp.reshape(24, 3) - oi).max() < 1e-15 and np.abs(jsr.reshape(10, 72).T - oj).max() < 1e-15
    lib.avt_model_destroy(h)
    # error behaviour: a malformed model is refused with a message, never a crash
    bad = capi.ModelArrays(smpl); bad.parent = bad.parent.copy(); bad.parent[0] = 0
    d2 = bad.desc()
    assert lib.avt_model_create(ctypes.byref(d2), ctypes.byref(h)) != 0
    assert b"parent[0]" in lib.avt_last_error()


def test_tile_layout_of_the_evaluation_kernel(smpl):
    """Host-side column layout behind the block-sparse J^T J (avt_model.cpp::build_tile_layout): a permutation of the
    parameters into 16-column tiles, every vertex's tile set covers the parameter blocks its residual depends on
    (AvatarOptimizer.cpp:620-629: root translation, ancestors' rotations, shape keys), vertices ordered by tile set."""
    lib = capi.load_library()
    arr = capi.ModelArrays(smpl)
    desc = arr.desc()
    h = ctypes.c_void_p()
    assert lib.avt_model_create(ctypes.byref(desc), ctypes.byref(h)) == 0
    V, J, K = arr.V, 24, 10
    P = 3 + 3 * J + K
    nt = ctypes.c_int()
    tp = np.full(16 * 11, -7, np.int32); vt = np.zeros(V, np.uint16); vo = np.zeros(V, np.int32)
    assert lib.avt_model_tile_layout(h, ctypes.byref(nt), capi.iptr(tp), vt.ctypes.data_as(ctypes.POINTER(ctypes.c_ushort)), capi.iptr(vo)) == 0
    NT = nt.value
    assert NT == (P + 1 + 15) // 16 == 6
    tp = tp[:16 * NT]
    real = tp[tp >= 0]
    assert sorted(real.tolist()) == list(range(P + 1)), "every parameter and the residual column exactly once"
    tile_of_param = np.empty(P + 1, int); tile_of_param[tp[tp >= 0]] = np.nonzero(tp >= 0)[0] // 16
    # the three parameters of a joint, the translation and the shape keys each stay inside one tile
    for j in range(J):
        assert len(set(tile_of_param[3 + 3 * j: 6 + 3 * j])) == 1
    assert len(set(tile_of_param[0:3])) == 1 and len(set(tile_of_param[3 + 3 * J: P])) == 1
    parent = np.asarray(arr.parent)
    W = np.asarray(smpl["weights"])
    pairs = 0
    for v in range(V):
        need = {tile_of_param[0], tile_of_param[3 + 3 * J], tile_of_param[P]}
        asg = np.nonzero(W[v] > 1e-12)[0]
        asg = asg[np.argsort(-W[v, asg], kind="stable")][:4]
        for k in asg:
            j = int(k)
            while j >= 0:
                need.add(tile_of_param[3 + 3 * j]); j = int(parent[j])
        got = {t for t in range(NT) if (vt[v] >> t) & 1}
        assert need <= got, (v, need, got)
        pairs += len(got) * (len(got) + 1) // 2
    assert pairs / V < 8.0, "the synthetic SMPL-shaped skeleton packs: few of the 21 tile pairs per vertex"
    assert sorted(vo.tolist()) == list(range(V))
    m = vt[vo]
    assert np.all(np.diff(m.astype(int)) >= 0), "vertices ordered by tile set"
    same = np.diff(m.astype(int)) == 0
    assert np.all(np.diff(vo)[same] > 0), "ascending vertex id inside a tile set"
    lib.avt_model_destroy(h)


def test_no_gpu_means_loud_failure(smpl):
    """Without a HIP device the product path must fail loudly, not fall back."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from avatar_amd import api
    gm = api.AvatarModel(smpl)
    with pytest.raises(api.AvtError):
        api.Context(gm, 24, synth.identity_part_map(), 1000, 1)


def test_product_never_imports_oracle():
    """The product path (avatar_amd/) must not import, include or link anything under oracle/."""
    pat_py = re.compile(r"^\s*(from|import)\s+oracle\b", re.M)
    pat_c = re.compile(r"#include\s*[<\"][^>\"]*oracle")
    for root, _, files in os.walk(os.path.join(ROOT, "avatar_amd")):
        for fn in files:
            path = os.path.join(root, fn)
            if fn.endswith(".py"):
                assert not pat_py.search(open(path).read()), path
            elif fn.endswith((".cpp", ".hip", ".h")) or fn == "Makefile":
                txt = open(path).read()
                assert not pat_c.search(txt) and "liboracle" not in txt, path


# ------------------------------------------------------------------------------------------------ N>1 sharding (gloo)
def test_frame_sharding_world_size_2_gloo(tmp_path):
    script = os.path.join(HERE, "dist_shard_check.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29731")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29731", script], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "SHARD_OK" in r.stdout


def test_model_create_error_paths_report_and_release(smpl):
    """avt_model_create failures (every one of them frees the half-built model exactly once and leaves a message)."""
    lib = capi.load_library()
    h = ctypes.c_void_p()

    def create(m):
        arr = capi.ModelArrays(m)
        d = arr.desc()
        rc = lib.avt_model_create(ctypes.byref(d), ctypes.byref(h))
        return rc, lib.avt_last_error().decode()
    W = np.array(smpl["weights"], np.float64)
    W5 = W.copy(); W5[7] = 0.0; W5[7, :5] = 0.2                      # five skinning weights on one vertex
    rc, msg = create(dict(smpl, weights=W5))
    assert rc != 0 and "4 skinning weights" in msg
    W0 = W.copy(); W0[11] = 0.0                                       # a vertex without weights
    rc, msg = create(dict(smpl, weights=W0))
    assert rc != 0 and "without skinning weights" in msg
    f = np.array(smpl["f"]).copy(); f[3, 1] = 10 ** 6                 # mesh index out of range
    rc, msg = create(dict(smpl, f=f))
    assert rc != 0 and "mesh index" in msg
    cov = np.array(smpl["prior_cov"]).copy(); cov[2] = -cov[2]        # covariance not positive definite
    rc, msg = create(dict(smpl, prior_cov=cov))
    assert rc != 0 and "positive definite" in msg
    rc, msg = create(smpl)                                            # and the library is still fine afterwards
    assert rc == 0
    lib.avt_model_destroy(h)
