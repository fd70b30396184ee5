"""Frame protocol of the reference's trackers (demo.cpp:215-290, live-demo.cpp:364-426) on top of AvatarOptimizer:
interval subsampling of the labelled depth image, the reinitialisation policy, per-frame ICP budgets and the temporal
warm start (the avatar state simply carries over between frames).  SURVEY.md §8 row f3.

Inputs per frame are what the reference's perception front-end produces (out of scope here): an XYZ map (H,W,3)
float32 in camera coordinates and a per-pixel body-part mask (H,W) uint8 with 255 = background, plus the foreground
bounding box (top, left, bottom, right), inclusive.
"""
from __future__ import annotations

import numpy as np

from . import api


class FrameTracker:
    def __init__(self, ava_opt: "api.AvatarOptimizer", interval=12, frame_icp_iters=3, reinit_icp_iters=6, reinit_cnz=1000,
                 num_threads=4, rtree=None, rtree_interval=2, dist_to_pre_weight=0.001, initial_per_part_cnz=0, initial_icp_iters=None):
        self.opt = ava_opt
        self.ava = ava_opt.ava
        self.interval = interval                      # demo.cpp:58  --data-interval
        self.frameICPIters = frame_icp_iters          # demo.cpp:63  --frame-icp-iters
        self.reinitICPIters = reinit_icp_iters        # demo.cpp:66  --reinit-icp-iters
        self.reinitCnz = reinit_cnz                   # demo.cpp:71  --min-points
        self.num_threads = num_threads
        self.initialPerPartCnz = initial_per_part_cnz # live-demo.cpp:89-90 --initial-per-part-thresh (80 there); 0 = demo.cpp: no per-part check
        self.initialICPIters = reinit_icp_iters if initial_icp_iters is None else initial_icp_iters   # live-demo.cpp:80
        self.firstTime = True                         # live-demo.cpp:256
        self.reinit = True                            # demo.cpp:151
        self.rtree = rtree                            # avatar_amd.rtree.RTree or None (labels supplied by the caller)
        self.rtreeInterval = rtree_interval           # demo.cpp:198 (predictBest / postProcess interval)
        self.distToPreWeight = dist_to_pre_weight     # live-demo.cpp:104-108
        self.comPre = None                            # demo.cpp:148: previous centres of mass for the post-processor

    def subsample(self, xyz, part_mask, bbox=None):
        """Every `interval`-th pixel of the bounding box that carries a body-part label (demo.cpp:216-250);
        y is negated (:245).  Returns (data_cloud (n,3) float64, labels (n,) int32)."""
        H, W = part_mask.shape
        top, left, bottom, right = bbox if bbox is not None else (0, 0, H - 1, W - 1)
        rows = np.arange(top, bottom + 1, self.interval)
        cols = np.arange(left, right + 1, self.interval)
        sub_mask = part_mask[np.ix_(rows, cols)]
        keep = sub_mask != 255
        if (sub_mask[keep] >= self.opt.numParts).any():
            raise ValueError("body part prediction out of range (demo.cpp:236-243)")
        pts = xyz[np.ix_(rows, cols)][keep].astype(np.float64)
        pts[:, 1] = -pts[:, 1]
        return pts, sub_mask[keep].astype(np.int32)

    def process(self, xyz, part_mask, bbox=None):
        """One tracked frame.  Returns True if the avatar was fitted, False if tracking was declared lost
        (too few body pixels: the next frame reinitialises, demo.cpp:225,283-285)."""
        data, labels = self.subsample(xyz, part_mask, bbox)
        part_missing = False                          # live-demo.cpp:376-380: the first fit wants every body part seen
        if self.firstTime and self.initialPerPartCnz > 0:
            part_cnz = np.bincount(labels, minlength=self.opt.numParts)
            part_missing = part_cnz.min() < max(1, self.initialPerPartCnz // (self.interval * self.interval))
        if len(labels) == 0 or part_missing or len(labels) < self.reinitCnz // (self.interval * self.interval):   # an empty frame is never fitted
            self.reinit = True
            return False
        icp_iters = self.frameICPIters
        ava = self.ava
        if self.reinit:                               # demo.cpp:252-265
            ava.p = data.mean(0)
            ava.w = np.zeros_like(ava.w)
            ava.r = np.tile(np.eye(3), (ava.model.numJoints(), 1, 1))
            ava.r[0] = np.array([[-1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, -1.0]])   # AngleAxis(pi, y)
            self.reinit = False
            ava.update()
            icp_iters = self.initialICPIters if self.firstTime else self.reinitICPIters     # live-demo.cpp:417-418
            self.firstTime = False
        self.opt.optimize(data, labels, icp_iters, self.num_threads)
        return True

    def label(self, xyz, bbox):
        """Per-pixel body parts of a foreground XYZ map with the forest (demo.cpp:196-204): predictBest on the GPU at
        `rtree_interval` inside the bounding box, then postProcess.  bbox = (top, left, bottom, right) inclusive."""
        if self.rtree is None:
            raise RuntimeError("FrameTracker.label: no RTree attached")
        top, left, bottom, right = bbox
        depth = np.ascontiguousarray(xyz[:, :, 2], np.float32)
        mask = self.rtree.predictBest(depth, 0, self.rtreeInterval, (left, top), (right, bottom))
        self.comPre = self.rtree.postProcess(mask, self.comPre, self.rtreeInterval, 1, (left, top), (right, bottom), self.distToPreWeight)
        return mask

    def process_depth(self, xyz, bbox):
        """One tracked frame from depth alone: label() then process()."""
        return self.process(xyz, self.label(xyz, bbox), bbox)
