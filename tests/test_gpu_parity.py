"""GPU parity tests proper: the HIP path (through the C ABI) against the CPU oracle on identical seeded inputs.

Bars (BASELINE.json north_star): correspondence indices bit-exact; joint angles and vertex positions within 1e-4
(we assert far tighter where the arithmetic allows).
"""
import numpy as np
import pytest

from avatar_amd import synth
from avatar_amd.capi import Options

pytestmark = pytest.mark.gpu


def _api():
    from avatar_amd import api
    return api


def test_library_loaded_is_hip():
    from avatar_amd import capi
    lib = capi.load_library()
    assert lib.avt_kernel_name(6) == b"eval"


def test_lbs_matches_oracle(smpl, omodel, gmodel):
    ctx = gmodel.default_ctx()
    ws, ps, Rs = [], [], []
    for s in range(6):
        w, p, R = synth.sample_ground_truth(smpl, s)
        ws.append(w); ps.append(p); Rs.append(R)
    cloud, jp, jt = ctx.lbs_update(np.array(ws), np.array(ps), np.array(Rs))
    for s in range(6):
        c0, jp0, jt0 = omodel.update(ws[s], ps[s], Rs[s])
        assert np.abs(cloud[s] - c0).max() < 1e-12
        assert np.abs(jp[s] - jp0).max() < 1e-12
        assert np.abs(jt[s] - jt0).max() < 1e-12


def test_lbs_known_answers(smpl, gmodel):
    """Identity pose, zero shape => cloud_v = base_v - J_0 + p (Avatar.cpp:47-49,59-64)."""
    api = _api()
    ava = api.Avatar(gmodel)
    ava.p = np.array([0.3, -0.2, 2.5])
    ava.update()
    J0 = gmodel.initialJointPos[0]
    assert np.abs(ava.cloud - (smpl["v_template"] - J0 + ava.p)).max() < 1e-13
    assert np.abs(ava.jointPos - (gmodel.initialJointPos - J0 + ava.p)).max() < 1e-13


def test_visibility_matches_oracle(smpl, omodel, gmodel):
    ctx = gmodel.default_ctx()
    for s in range(3):
        w, p, R = synth.sample_ground_truth(smpl, s)
        c0, _, _ = omodel.update(w, p, R)
        assert np.array_equal(ctx.visibility(c0, True), omodel.visibility(c0, True))
        assert ctx.visibility(c0, False).all()


def test_nn_bit_exact_vs_oracle(smpl, omodel, gmodel, frame0):
    api = _api()
    pm = synth.identity_part_map()
    ctx = api.Context(gmodel, 24, pm, 60000, 1)
    for s in range(3):
        fr = synth.make_frame(smpl, s)
        w0, p0, R0 = fr["start"]
        c0, _, _ = omodel.update(w0, p0, R0)
        vis = omodel.visibility(c0, True)
        ref = omodel.nn(pm, 24, c0, vis, fr["data"], fr["labels"])
        got = ctx.nn(c0, vis, fr["data"], fr["labels"])
        assert np.array_equal(ref, got)


def test_nn_golden_nanoflann(gmodel):
    """Against the committed outputs of the reference's own nanoflann (tests/golden/make_nn_golden.py)."""
    import os
    api = _api()
    path = os.path.join(os.path.dirname(__file__), "golden", "nn_golden.npz")
    z = np.load(path)
    ncase = int(z["ncase"])
    for k in range(ncase):
        pm = z[f"part_map_{k}"]; npart = int(z[f"num_parts_{k}"])
        ctx = api.Context(gmodel, npart, pm, 60000, 1)
        got = ctx.nn(z[f"cloud_{k}"], z[f"vis_{k}"], z[f"data_{k}"], z[f"labels_{k}"])
        assert np.array_equal(got, z[f"idx_{k}"]), f"case {k}"


def _start_state(fr):
    from avatar_amd import api
    w0, p0, R0 = fr["start"]
    return p0, api.rot_to_quat(R0), w0


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_optimize_matches_oracle(smpl, omodel, gmodel, seed):
    api = _api()
    pm = synth.identity_part_map()
    fr = synth.make_frame(smpl, seed)
    p0, q0, w0 = _start_state(fr)
    opt = Options.demo()
    ref = omodel.optimize(pm, 24, fr["data"], fr["labels"], opt, p0, q0, w0, aggregate=1)
    ctx = api.Context(gmodel, 24, pm, 60000, 1)
    p, q, w, st = ctx.optimize_batch([fr["data"]], [fr["labels"]], opt, p0[None], q0[None], w0[None])
    corr = ctx.correspondences(0, len(fr["labels"]))
    assert np.array_equal(corr, ref["corr"]), "correspondence indices must be bit-exact"
    assert st[0].num_correspondences == ref["stats"].num_correspondences
    assert st[0].matched_model_points == ref["stats"].matched_model_points
    assert st[0].gn_iterations == ref["stats"].gn_iterations
    assert st[0].accepted_steps == ref["stats"].accepted_steps
    assert abs(st[0].final_cost - ref["stats"].final_cost) <= 1e-9 * abs(ref["stats"].final_cost)
    assert np.abs(p[0] - ref["p"]).max() < 1e-7
    assert np.abs(w[0] - ref["w"]).max() < 1e-6
    # joint angles: quaternion sign-insensitive comparison, well inside the 1e-4 bar
    dq = np.minimum(np.abs(q[0] - ref["q"]).max(1), np.abs(q[0] + ref["q"]).max(1))
    assert dq.max() < 1e-7
    assert np.abs(ctx.cloud(0) - ref["cloud"]).max() < 1e-7          # vertex positions (bar: 1e-4)


def test_optimize_two_icp_iterations(smpl, omodel, gmodel):
    api = _api()
    pm = synth.identity_part_map()
    fr = synth.make_frame(smpl, 3)
    p0, q0, w0 = _start_state(fr)
    opt = Options.counted(icp_iters=2, max_iters_per_icp=5)
    ref = omodel.optimize(pm, 24, fr["data"], fr["labels"], opt, p0, q0, w0, aggregate=1)
    ctx = api.Context(gmodel, 24, pm, 60000, 1)
    p, q, w, st = ctx.optimize_batch([fr["data"]], [fr["labels"]], opt, p0[None], q0[None], w0[None])
    assert np.array_equal(ctx.correspondences(0, len(fr["labels"])), ref["corr"])
    assert np.abs(ctx.cloud(0) - ref["cloud"]).max() < 1e-6
    assert st[0].gn_iterations == 10


def test_batch_equals_single(smpl, gmodel):
    api = _api()
    pm = synth.identity_part_map()
    frs = [synth.make_frame(smpl, s) for s in (4, 5, 6)]
    opt = Options.demo(max_iters_per_icp=4)
    st0 = [_start_state(fr) for fr in frs]
    ctxb = api.Context(gmodel, 24, pm, 60000, 4)
    pb, qb, wb, _ = ctxb.optimize_batch([fr["data"] for fr in frs], [fr["labels"] for fr in frs], opt,
                                        np.array([s[0] for s in st0]), np.array([s[1] for s in st0]),
                                        np.array([s[2] for s in st0]))
    ctx1 = api.Context(gmodel, 24, pm, 60000, 1)
    for f, fr in enumerate(frs):
        p, q, w, _ = ctx1.optimize_batch([fr["data"]], [fr["labels"]], opt, st0[f][0][None], st0[f][1][None], st0[f][2][None])
        # same kernels, different G (eval workgroups per frame) => different summation order only
        assert np.abs(p[0] - pb[f]).max() < 1e-9 and np.abs(q[0] - qb[f]).max() < 1e-9 and np.abs(w[0] - wb[f]).max() < 1e-8


def test_facade_protocol(smpl, omodel, gmodel):
    """The demo.cpp:252-268 call protocol through the mirrored classes."""
    api = _api()
    fr = synth.make_frame(smpl, 7)
    ava = api.Avatar(gmodel)
    w0, p0, R0 = fr["start"]
    ava.w, ava.p, ava.r = w0.copy(), p0.copy(), R0.copy()
    ava.update()
    optim = api.AvatarOptimizer(ava, None, (1280, 720), 24, synth.identity_part_map(), max_points=60000)
    optim.betaPose, optim.betaShape = 0.05, 0.12
    optim.optimize(fr["data"], fr["labels"], 1, 4)
    ref = omodel.optimize(synth.identity_part_map(), 24, fr["data"], fr["labels"], optim.options(), p0, api.rot_to_quat(R0), w0, aggregate=1)
    assert np.abs(ava.cloud - ref["cloud"]).max() < 1e-6
    assert optim.last_stats.final_cost < optim.last_stats.initial_cost


def test_edge_cases(smpl, omodel, gmodel):
    api = _api()
    pm = synth.identity_part_map()
    fr = synth.make_frame(smpl, 8)
    p0, q0, w0 = _start_state(fr)
    ctx = api.Context(gmodel, 24, pm, 60000, 1)
    opt = Options.demo(max_iters_per_icp=3)
    # (a) a part with data but no visible model point: every point labelled with one part, model far rotated
    lab = np.full(500, 10, np.int32)
    ref = omodel.optimize(pm, 24, fr["data"][:500], lab, opt, p0, q0, w0, aggregate=1)
    p, q, w, st = ctx.optimize_batch([fr["data"][:500]], [lab], opt, p0[None], q0[None], w0[None])
    assert np.array_equal(ctx.correspondences(0, 500), ref["corr"])
    assert np.abs(ctx.cloud(0) - ref["cloud"]).max() < 1e-6
    # (b) ragged tiny input: 3 points
    ref = omodel.optimize(pm, 24, fr["data"][:3], fr["labels"][:3], opt, p0, q0, w0, aggregate=1)
    p, q, w, st = ctx.optimize_batch([fr["data"][:3]], [fr["labels"][:3]], opt, p0[None], q0[None], w0[None])
    assert np.array_equal(ctx.correspondences(0, 3), ref["corr"])
    assert np.abs(ctx.cloud(0) - ref["cloud"]).max() < 1e-6
    # (c) occlusion off
    opt2 = Options.demo(max_iters_per_icp=3, enable_occlusion=0)
    ref = omodel.optimize(pm, 24, fr["data"], fr["labels"], opt2, p0, q0, w0, aggregate=1)
    p, q, w, st = ctx.optimize_batch([fr["data"]], [fr["labels"]], opt2, p0[None], q0[None], w0[None])
    assert np.array_equal(ctx.correspondences(0, len(fr["labels"])), ref["corr"])
    assert np.abs(ctx.cloud(0) - ref["cloud"]).max() < 1e-6


def test_nn_full_size_against_nanoflann_goldens(smpl, omodel, gmodel):
    """k_nn / k_compact at FULL size against the reference's own nanoflann output (38 k, 125 k dense, coarse 6-part map):
    index arrays committed under tests/golden/, inputs regenerated from seeds.  Bit-exact."""
    import os
    import sys
    from avatar_amd import api
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "golden"))
    import make_nn_golden_full as mk
    z = np.load(os.path.join(here, "golden", "nn_golden_full.npz"))
    for k, c in enumerate(mk.CASES):
        pm, npart, cloud, vis, data, labels = mk.case_inputs(smpl, omodel, c)
        ctx = api.Context(gmodel, npart, pm, len(labels), 1, device=0)
        got = ctx.nn(cloud, vis, data, labels)
        assert np.array_equal(got, z[f"idx_{k}"]), (k, int((got != z[f"idx_{k}"]).sum()))
