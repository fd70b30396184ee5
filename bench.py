#!/usr/bin/env python3
"""bench.py — Gauss-Newton iterations/sec of the MI355X-native AvatarOptimizer hot path.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 launched one rank per GPU by
torch.distributed.run.  A "step" is one optimize() (AvatarOptimizer.cpp:1246-1517: back-face visibility, per-part
NN, `maxItersPerICP`=10 GN/LM iterations, LBS update) over this rank's resident batch of synthetic frames.
Workload at N=1: BASELINE.json configs[1] — ONE ~30k-point synthetic smplsynth cloud, 10 GN iterations,
10 shape + 24-joint pose (P=85), knobs of demo.cpp:54-57.  `--frames F` runs F independent frames per GPU
(configs[2]: 64), `--dense` the 120k-point stress frame (configs[4]).  Frames are independent, so ranks shard
them with no data-path collective ("scaling": "weak"): per-GPU work is fixed as N grows.

Rank 0 prints ONE JSON line; `value` = GN iterations of all frames on all ranks / max-over-ranks wall time of the
timed region, inputs already resident in HBM.  `roofline` is for the kernel class with the largest share of device
time; `cpu_baseline` times the CPU oracle (a scalar fp64 restatement of the reference algorithm, not Ceres) on
this host's cores on the same frame.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP64_MFMA_PEAK_TFLOPS = 78.6   # public MI355X spec, fp64 matrix (not listed in the guide's MFMA table)


def algorithmic_bytes_per_gn_iter(N, V, K, P):
    """data xyz + corr idx + base&key clouds + 4 (w,idx) pairs + cloud out + H,g out."""
    return 24 * N + 4 * N + 24 * V * (K + 1) + 48 * V + 24 * V + 8 * P * (P + 1)


PMC_FILES = {1: "profiles/r01_pmc_single_frame.json", 64: "profiles/r01_pmc_64_frames.json"}
KERNEL_SYMBOL = {"eval": "k_evalILi24ELi10", "solve": "k_solve", "reduce": "k_reduce", "nn": "k_nnILi"}   # mangled-name fragments


def pmc_traffic(frames, kernel_class):
    """HBM bytes per launch of `kernel_class` from the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this
    same command (tools/pmc_summary.py applies the gfx950 corrections of MI355X_MICROARCH.md §HBM); None if not measured."""
    path = os.path.join(ROOT, PMC_FILES.get(frames, ""))
    try:
        d = json.load(open(path))
        hit = [v for k, v in d["kernels"].items() if KERNEL_SYMBOL[kernel_class] in k]
        return int(hit[0]["hbm_bytes"])
    except Exception:
        return None


def measure(api, synth, Options, torch, dist, smpl, gm, args, F, steps, warmup, rank, world, local_rank, dense):
    """Times `steps` optimize() calls over F resident frames on this rank; returns the per-config result dict."""
    V, J, K, P = gm.numPoints(), gm.numJoints(), gm.numShapeKeys(), gm.arrays.P
    pm = synth.identity_part_map()
    # F distinct synthetic frames (seed = global frame id), rendered on the GPU straight into the resident buffers
    # (avt_synth_render_frames: depth + part render of the ground-truth avatar, back-projection, y flip)
    gts = [synth.sample_ground_truth(smpl, rank * F + f) for f in range(F)]
    starts = [synth.perturb_start(*gts[f], rank * F + f) for f in range(F)]
    ctx = api.Context(gm, 24, pm, 200000 if dense else 65536, F, device=local_rank)
    opt = Options.demo(icp_iters=args.icp_iters)
    npts = ctx.render_frames(np.array([g[0] for g in gts]), np.array([g[1] for g in gts]), np.array([g[2] for g in gts]),
                             res_scale=2 if dense else 1)                      # inputs resident in HBM
    p0 = np.array([s[1] for s in starts])
    q0 = np.array([api.rot_to_quat(s[2]) for s in starts])
    w0 = np.array([s[0] for s in starts])
    d0, l0 = ctx.frame_download(0)
    frs = [{"data": d0, "labels": l0}]

    ctx.state_upload(p0, q0, w0)          # the tracking start states (109 doubles per frame): resident like the frames

    def step():
        ctx.state_reset()                 # device-side reinstall of the start state (asynchronous, no host transfer)
        ctx.optimize_resident(opt)        # asynchronous on the context's stream (one hipGraph replay)

    # which kernel class dominates this configuration (one instrumented, untimed step)
    for _ in range(max(1, warmup)):
        step()
    ctx.sync()
    ctx.profile_begin()
    step()
    ctx.sync()
    prof = ctx.profile_end()
    dominant = max(prof, key=lambda k: prof[k][0])
    for _ in range(warmup):
        step()
    ctx.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    # timed region; HIP events only around the dominant kernel class, on the stream it is launched on
    ctx.profile_begin(classes=[dominant])
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    ctx.sync()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    prof_timed = ctx.profile_end()
    elapsed = t1 - t0
    if world > 1:
        te = torch.tensor([elapsed], dtype=torch.float64, device=("cuda" if args.backend == "nccl" else "cpu"))
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
        dist.barrier()
    # the same K steps once more WITHOUT any event records: `value` must not carry instrumentation overhead
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    ctx.sync()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    elapsed_clean = t1 - t0
    if world > 1:
        te = torch.tensor([elapsed_clean], dtype=torch.float64, device=("cuda" if args.backend == "nccl" else "cpu"))
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed_clean = float(te.item())
        dist.barrier()
    p, q, w, st = ctx.state_download()
    gn_per_step = F * opt.icp_iters * opt.max_iters_per_icp
    res = {"value": world * gn_per_step * steps / elapsed_clean, "elapsed": elapsed_clean, "elapsed_with_events": elapsed,
           "steps": steps, "F": F}
    tot = sum(v[0] for v in prof.values())
    res["kernels"] = {k: {"ms": round(v[0], 5), "launches": v[1], "share": round(v[0] / tot, 4)} for k, v in prof.items() if v[1]}
    Nmean = float(np.mean(npts))
    M = float(np.mean([s.matched_model_points for s in st]))
    bytes_launch = F * algorithmic_bytes_per_gn_iter(Nmean, V, K, P)
    avg_ms = prof_timed[dominant][0] / max(1, prof_timed[dominant][1])      # live, over the (event-instrumented) timed region
    achieved = bytes_launch / (avg_ms * 1e-3) / 1e9
    res["roofline"] = {"kernel": "k_" + dominant, "bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": pmc_traffic(F, dominant),
                       "avg_launch_us": round(avg_ms * 1e3, 3), "launches_timed": prof_timed[dominant][1],
                       "algorithmic_bytes_per_launch": int(bytes_launch),
                       "note": "achieved = frames x SURVEY 8(d) bytes per GN iteration / mean launch time of the dominant kernel class; "
                               "traffic = HBM bytes per launch from committed rocprofv3 PMC passes (profiles/)"}
    ev = prof["eval"]
    ev_ms = ev[0] / max(1, ev[1])
    tfl = F * 3.0 * M * P * (P + 1) / (ev_ms * 1e-3) / 1e12 if ev_ms > 0 else 0.0
    res["eval_kernel"] = {"avg_launch_us_with_events": round(ev_ms * 1e3, 3), "jtj_tflops_f64": round(tfl, 4), "mfma_peak_tflops": FP64_MFMA_PEAK_TFLOPS,
                          "mfma_frac": round(tfl / FP64_MFMA_PEAK_TFLOPS, 6)}
    res["points_per_frame"] = int(Nmean)
    res["matched_model_points"] = int(M)
    res["final_cost_frame0"] = st[0].final_cost
    res["accepted_steps_frame0"] = st[0].accepted_steps
    res["frames0"] = frs[0]
    res["opt"] = opt
    res["start0"] = (p0[0], q0[0], w0[0])
    del ctx
    return res


def render_stage(api, synth, smpl, gm, with_cpu, local_rank):
    """the synthetic-frame generator (depth + part render of a posed avatar, back-projection, labels)
    on the GPU, 8 frames per call into the resident frame buffers, against the host generator the parity tests use."""
    import numpy as np
    F = 8
    pm = synth.identity_part_map()
    gts = [synth.sample_ground_truth(smpl, 500 + f) for f in range(F)]
    W, Pp, R = (np.array([g[i] for g in gts]) for i in range(3))
    ctx = api.Context(gm, 24, pm, 65536, F, device=local_rank)
    for _ in range(3):
        npts = ctx.render_frames(W, Pp, R)
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.render_frames(W, Pp, R)            # synchronous: returns the point count of every frame
    dt = time.perf_counter() - t0
    res = {"workload": f"{F} posed avatars at 1280x720 (K4A intrinsics) per call, {int(np.mean(npts))} foreground points per frame",
           "value": round(F * reps / dt, 1), "unit": "frames/s", "ms_per_call_of_8": round(dt / reps * 1e3, 3)}
    if with_cpu:
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 2.0:
            w, p, Rg = gts[n % F]
            synth.render_images(smpl, synth.pose_vertices(smpl, w, p, Rg), pm)
            n += 1
        res["cpu_baseline"] = {"value": round(n / (time.perf_counter() - t0), 1), "unit": "frames/s", "cores": 1, "kind": "port",
                               "sample": f"{n} frames by the host generator (avatar_amd/csrc/synth_render.cpp + numpy LBS), 1 thread; "
                                         "tests/test_gpu_render.py checks the GPU frames against it bit for bit"}
    return res


def tracker_stage(api, synth, smpl, gm):
    """the per-frame protocol of demo.cpp:215-290 through the host-buffer entry points (interval
    subsampling of the foreground bounding box, reinit on the first frame, frame-to-frame warm start, 3 ICP iterations
    per frame, avatar refreshed after every frame).  Wall clock per frame INCLUDING host work and PCIe transfers."""
    import numpy as np
    from avatar_amd.tracker import FrameTracker
    w, p, R = synth.sample_ground_truth(smpl, 21, use_gmm=False)
    w = 0.5 * w
    frames = []
    for k in range(6):
        Rk = R.copy()
        Rk[16] = R[16] @ synth.rodrigues([0.0, 0.0, 0.05 * k]); Rk[4] = R[4] @ synth.rodrigues([0.04 * k, 0.0, 0.0])
        xyz, mask, _ = synth.render_images(smpl, synth.pose_vertices(smpl, w, p + np.array([0.01 * k, 0.0, 0.0]), Rk), synth.identity_part_map())
        ys, xs = np.nonzero(mask != 255)
        frames.append((xyz, mask, (ys.min(), xs.min(), ys.max(), xs.max())))
    ava = api.Avatar(gm)
    opt = api.AvatarOptimizer(ava, None, (1280, 720), 24, synth.identity_part_map(), max_points=8192)
    opt.betaPose, opt.betaShape = 0.05, 0.12
    tr = FrameTracker(opt, interval=3, frame_icp_iters=3, reinit_icp_iters=6, reinit_cnz=1000)
    npts = len(tr.subsample(*frames[0])[1])
    for xyz, mask, bbox in frames:            # warm-up: graph capture for both ICP budgets
        tr.process(xyz, mask, bbox)
    reps, t_sub = 5, 0.0
    t0 = time.perf_counter()
    for _ in range(reps):
        for xyz, mask, bbox in frames[1:]:
            tr.process(xyz, mask, bbox)
    dt = (time.perf_counter() - t0) / (reps * (len(frames) - 1))
    t1 = time.perf_counter()
    for xyz, mask, bbox in frames[1:]:
        tr.subsample(xyz, mask, bbox)
    t_sub = (time.perf_counter() - t1) / (len(frames) - 1)
    return {"workload": f"demo.cpp frame loop on 1280x720 renders: interval 3 ({npts} points per frame), 3 ICP x 10 GN iterations per frame, warm start",
            "value": round(1.0 / dt, 1), "unit": "frames/s", "ms_per_frame": round(dt * 1e3, 3),
            "of_which_host_subsampling_ms": round(t_sub * 1e3, 3), "gn_iterations_per_s": round(30.0 / dt, 1)}


def label_stage(synth, smpl, with_cpu):
    """RTree::predictBest on 1280x720 depth renders exactly as the tracker calls it (interval 2,
    foreground bounding box, gaps filled; demo.cpp:196-199), 8 resident images per launch, plus the full-resolution walk."""
    import numpy as np
    from avatar_amd import rtree, synth_forest
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "forest_small.srtr")
    tree = rtree.RTree(path)
    depths, boxes = [], []
    for s in range(8):
        w, p, R = synth.sample_ground_truth(smpl, 300 + s)
        xyz, mask, _ = synth.render_images(smpl, synth.pose_vertices(smpl, w, p, R), synth.identity_part_map())
        depths.append(synth_forest.depth_of(xyz))
        rr, cc = np.nonzero(mask != 255)
        boxes.append(((int(cc.min()), int(rr.min())), (int(cc.max()), int(rr.max()))))
    D = np.stack(depths)
    tl = (min(b[0][0] for b in boxes), min(b[0][1] for b in boxes)); br = (max(b[1][0] for b in boxes), max(b[1][1] for b in boxes))
    tree.upload_images(D)
    res = {"workload": "8 synthetic 1280x720 depth renders (~30k foreground pixels each), toy tree tests/golden/forest_small.srtr "
                       f"({len(tree.links)} nodes)", "unit": "images/s"}
    for name, kw in (("tracker_call_interval2_bbox", dict(interval=2, top_left=tl, bot_right=br)), ("full_image_interval1", dict(interval=1))):
        for _ in range(3):
            tree.predict_resident(**kw)
        tree.sync()
        t0 = time.perf_counter()
        reps = 50
        for _ in range(reps):
            tree.predict_resident(**kw)
        tree.sync()
        dt = time.perf_counter() - t0
        walked = sum(int((d[kw.get("top_left", (0, 0))[1] + kw["interval"]::kw["interval"], ::kw["interval"]] != 0).sum()) for d in D) \
            if "top_left" not in kw else sum(int((d[tl[1] + 2:br[1] + 1:2, tl[0]:br[0] + 1:2] != 0).sum()) for d in D)
        res[name] = {"value": round(8 * reps / dt, 1), "ms_per_launch_of_8": round(dt / reps * 1e3, 4), "tree_walks_per_s": round(walked * reps / dt, 0)}
    if with_cpu:
        from oracle import rtree_oracle
        ot = rtree_oracle.OracleRTree.load(path)
        t0 = time.perf_counter()
        n = 0
        while time.perf_counter() - t0 < 2.0:
            ref = ot.predictBest(D[n % 8], interval=2, top_left=tl, bot_right=br)
            n += 1
        cpu = n / (time.perf_counter() - t0)
        res["cpu_baseline"] = {"value": round(cpu, 1), "unit": "images/s", "cores": 1, "kind": "port",
                               "sample": f"{n} x predictBest(interval 2, bbox) by the CPU restatement (oracle/rtree_oracle.cpp), 1 thread"}
        tree.predict_resident(interval=2, top_left=tl, bot_right=br)
        assert np.array_equal(tree.download_labels((n - 1) % 8), ref), "label stage: GPU labels differ from the oracle"
        res["labels_bit_exact_vs_oracle"] = True
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=1, help="independent frames per GPU (configs[2] uses 64)")
    ap.add_argument("--dense", action="store_true", help="120k-point stress frames (configs[4])")
    ap.add_argument("--icp-iters", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-throughput-config", action="store_true", help="skip the secondary 64-frames-per-GPU measurement")
    ap.add_argument("--no-label-stage", action="store_true", help="skip the body-part forest (RTree) stage measurement")
    ap.add_argument("--no-render-stage", action="store_true", help="skip the synthetic-frame generator (row f1) measurement")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for single-GPU dry runs)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    if os.environ.get("AVT_BENCH_SHARE_GPU0"):      # dry run of the N>1 code path on a single-GPU box
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", init_method="env://", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend, init_method="env://")

    from avatar_amd import api, synth
    from avatar_amd.capi import Options

    smpl = synth.load_model(0)
    gm = api.AvatarModel(smpl)
    P = gm.arrays.P
    F = args.frames
    r = measure(api, synth, Options, torch, dist, smpl, gm, args, F, args.steps, args.warmup, rank, world, local_rank, args.dense)
    r2 = None
    if F == 1 and not args.dense and not args.no_throughput_config:
        r2 = measure(api, synth, Options, torch, dist, smpl, gm, args, 64, max(10, args.steps // 2), 3, rank, world, local_rank, False)
    if rank == 0:
        opt = r["opt"]
        out = {
            "metric": "Gauss-Newton iterations/sec (30k-pt cloud, 10 shape + 24-joint pose)",
            "value": round(r["value"], 2), "unit": "GN iterations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(r["elapsed"] / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("dense 120k-pt stress frame" if args.dense else "1 synthetic smplsynth cloud (~30k pts)")
                       + f", {F} frame(s)/GPU, icp_iters={opt.icp_iters}, maxItersPerICP={opt.max_iters_per_icp}, P={P}",
                       "frames_per_gpu": F, "points_per_frame": r["points_per_frame"], "matched_model_points": r["matched_model_points"],
                       "parallelism": f"frames sharded over {world} GPU(s), no data-path collective"},
            "roofline": r["roofline"], "eval_kernel": r["eval_kernel"], "kernels": r["kernels"],
            "ms_per_step_with_event_records": round(r["elapsed_with_events"] / args.steps * 1e3, 4),
            "final_cost_frame0": r["final_cost_frame0"], "accepted_steps_frame0": r["accepted_steps_frame0"],
        }
        if r2 is not None:
            out["throughput_config"] = {
                "workload": "BASELINE configs[2]: 64 independent ~30k-pt frames per GPU, same optimize()", "value": round(r2["value"], 2),
                "unit": "GN iterations/s", "steps": r2["steps"], "ms_per_step": round(r2["elapsed"] / r2["steps"] * 1e3, 4),
                "roofline": r2["roofline"], "eval_kernel": r2["eval_kernel"], "kernels": r2["kernels"]}
        out["frames_per_s"] = round(F * world * args.steps / r["elapsed"], 2)
        out["icp_iterations_per_s"] = round(F * world * opt.icp_iters * args.steps / r["elapsed"], 2)
        if F == 1 and not args.dense and not args.no_render_stage:
            out["render_stage"] = render_stage(api, synth, smpl, gm, not args.no_cpu_baseline, local_rank)
        if F == 1 and not args.dense and not args.no_label_stage:
            out["label_stage"] = label_stage(synth, smpl, not args.no_cpu_baseline)
        if F == 1 and not args.dense and not args.no_render_stage:
            out["tracker_stage"] = tracker_stage(api, synth, smpl, gm)
        if not args.no_cpu_baseline:
            from oracle import oracle as orc
            om = orc.OracleModel(smpl)
            fr = r["frames0"]
            p0, q0, w0 = r["start0"]
            pm = synth.identity_part_map()
            ncpu = os.cpu_count() or 1

            def cpu_rate(aggregate, nthreads, budget):
                reps, tt = 0, 0.0
                while tt < budget:
                    a = time.perf_counter()
                    om.optimize(pm, 24, fr["data"], fr["labels"], opt, p0, q0, w0, aggregate=aggregate, nthreads=nthreads)
                    tt += time.perf_counter() - a
                    reps += 1
                return reps * opt.icp_iters * opt.max_iters_per_icp / tt, reps

            v1, reps = cpu_rate(1, 1, args.cpu_seconds * 0.6)
            vlit, _ = cpu_rate(0, 1, args.cpu_seconds * 0.2)
            vall, _ = cpu_rate(1, min(ncpu, 32), args.cpu_seconds * 0.2)
            out["cpu_baseline"] = {
                "value": round(v1, 2), "unit": "GN iterations/s", "cores": 1, "kind": "port",
                "sample": f"{reps} x optimize() of frame 0 ({len(fr['labels'])} pts, 10 GN iterations): CPU restatement of the "
                          f"sxyu/avatar algorithm (oracle/, not Ceres), same LM schedule, aggregated normal equations, 1 thread",
                "reference_structure_per_residual_block_1thread": round(vlit, 2),
                "aggregated_%d_threads" % min(ncpu, 32): round(vall, 2), "host_cores": ncpu,
            }
            out["speedup_vs_cpu_port"] = round(r["value"] / v1, 1)
            if r2 is not None:
                out["throughput_config"]["speedup_vs_cpu_port"] = round(r2["value"] / v1, 1)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
