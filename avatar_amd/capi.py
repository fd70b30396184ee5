"""ctypes view of include/avt.h: struct layouts, model-descriptor marshalling and the library loader.

The product path is libavatar_hip.so (hand-written HIP for gfx950 behind the C ABI).  There is NO CPU
fallback: if the shared library is missing or fails to load, `load_library()` raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AVT_LIB", os.path.join(_HERE, "csrc", "libavatar_hip.so"))   # AVT_LIB: instrumented builds (tools/)

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)
c_ubyte_p = C.POINTER(C.c_ubyte)

AVT_K_NAMES = ["lbs", "visibility", "bucket", "nn", "aggregate", "prepare", "eval", "reduce", "solve", "decide", "moments"]
AVT_K_COUNT = len(AVT_K_NAMES)


class ModelDesc(C.Structure):
    _fields_ = [
        ("num_points", C.c_int), ("num_joints", C.c_int), ("num_shape_keys", C.c_int), ("num_faces", C.c_int),
        ("base_cloud", c_double_p), ("key_clouds", c_double_p), ("parent", c_int_p), ("mesh", c_int_p),
        ("weights_colptr", c_int_p), ("weights_row", c_int_p), ("weights_val", c_double_p),
        ("jreg_colptr", c_int_p), ("jreg_row", c_int_p), ("jreg_val", c_double_p),
        ("prior_ncomps", C.c_int), ("prior_ndims", C.c_int),
        ("prior_weight", c_double_p), ("prior_mean", c_double_p), ("prior_cov", c_double_p),
        # the reference's legacy model format only (zero / NULL for model.npz): include/avt.h
        ("limit_one_joint_per_point", C.c_int), ("reserved1", C.c_int),
        ("joint_shape_reg_base", c_double_p), ("joint_shape_reg", c_double_p),
    ]


class Options(C.Structure):
    _fields_ = [
        ("beta_pose", C.c_double), ("beta_shape", C.c_double),
        ("nn_step", C.c_int), ("max_iters_per_icp", C.c_int), ("enable_occlusion", C.c_int),
        ("icp_iters", C.c_int), ("num_threads", C.c_int), ("lm_policy", C.c_int),
        ("lm_lambda0", C.c_double), ("lm_up", C.c_double), ("lm_down", C.c_double),
        ("lm_lambda_min", C.c_double), ("lm_lambda_max", C.c_double), ("function_tolerance", C.c_double),
    ]

    LM_FIXED_FACTORS, LM_GAIN_RATIO = 0, 1      # include/avt.h AVT_LM_*

    @classmethod
    def reference_defaults(cls):
        """AvatarOptimizer.h:28-39 member defaults, the reference's function_tolerance (AvatarOptimizer.cpp:1333) and the step rule's defaults
        (the gain-ratio schedule since round 6).  Mirrors avt_options_default (avt_model.cpp), the one place the defaults
        are set: tests/test_oracle_cpu.py::test_python_option_defaults_are_the_c_defaults compares the two field by field."""
        return cls(beta_pose=0.1, beta_shape=1.0, nn_step=20, max_iters_per_icp=10, enable_occlusion=1,
                   icp_iters=1, num_threads=4, lm_policy=cls.LM_GAIN_RATIO, lm_lambda0=1e-3, lm_up=cls.GAIN_LM_UP, lm_down=1.0 / 3.0,
                   lm_lambda_min=1e-12, lm_lambda_max=1e8, function_tolerance=1e-4)

    @classmethod
    def demo(cls, **kw):
        """The knobs the reference's trackers run with (demo.cpp:54-57,139-143).  `lm_policy=0` without an `lm_up` selects the fixed-factor
        schedule of rounds 1-5 with the constant it was tuned with (avt_options_fixed_factors)."""
        o = cls.reference_defaults()
        o.beta_pose, o.beta_shape = 0.05, 0.12
        for k, v in kw.items():
            setattr(o, k, v)
        if "lm_up" not in kw:
            o.lm_up = cls.GAIN_LM_UP if o.lm_policy == cls.LM_GAIN_RATIO else cls.FIXED_LM_UP
        return o

    @classmethod
    def counted(cls, **kw):
        """demo() with the stopping rule off: exactly max_iters_per_icp iterations per ICP iteration (what the benchmark counts and what tests
        that compare iteration by iteration want)."""
        kw.setdefault("function_tolerance", 0.0)
        return cls.demo(**kw)

    # gain-ratio schedule (lm_policy = 1): the multiplier of the first rejection after an accepted step.  Nielsen's 2 climbs too slowly for
    # these frames (three rejections in a row after the second step of most of them); over the 12 bench seeds (tools/damping_policy_compare.py)
    # 2 / 4 / 8 / 16 / 32 accept 0.76 / 0.80 / 0.83 / 0.87 / 0.86 of the iterations of one ICP iteration and end at a mean objective of
    # 28.39 / 28.10 / 27.91 / 27.98 / 27.87 (fixed factors: 0.57, 28.98)
    GAIN_LM_UP = 16.0
    FIXED_LM_UP = 4.0       # the fixed-factor schedule's rejection multiplier (rounds 1-5)


class Tuning(C.Structure):
    """include/avt.h avt_tuning: launch-shape and algorithm knobs of a context (defaults = the measured optima)."""
    _fields_ = [(n, C.c_int) for n in ("use_graph", "groups", "g", "gcap", "vis_frame_min", "ride", "ride_strips", "ride_sizing_groups", "nspec",
                                       "nn_force_part", "nn_slab", "mom_min_frames", "debug", "asm_parts")] + [("ride_timeout_us", C.c_longlong), ("lbs_frames", C.c_int), ("spec_cost", C.c_int), ("xcd_frames", C.c_int), ("literal_dims", C.c_int)]
    DEFAULTS = dict(use_graph=1, groups=0, g=0, gcap=128, vis_frame_min=32, ride=1, ride_strips=0, ride_sizing_groups=0, nspec=4, nn_force_part=0,
                    nn_slab=1, mom_min_frames=8, debug=0, asm_parts=1, ride_timeout_us=2000000, lbs_frames=0, spec_cost=1, xcd_frames=1, literal_dims=1)

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_ if n != "reserved"}

    def non_default(self):
        return {k: v for k, v in self.as_dict().items() if v != self.DEFAULTS[k]}


class Stats(C.Structure):
    _fields_ = [
        ("initial_cost", C.c_double), ("final_cost", C.c_double), ("lambda_", C.c_double),
        ("num_correspondences", C.c_int), ("matched_model_points", C.c_int),
        ("gn_iterations", C.c_int), ("accepted_steps", C.c_int),
    ]


class Profile(C.Structure):
    _fields_ = [("ms", C.c_double * AVT_K_COUNT), ("launches", C.c_int * AVT_K_COUNT)]


def dptr(a):
    return a.ctypes.data_as(c_double_p)


def iptr(a):
    return a.ctypes.data_as(c_int_p)


def bptr(a):
    return a.ctypes.data_as(c_ubyte_p)


class ModelArrays:
    """Host arrays in the layout avt_model_desc wants, built from SMPL-npz-style arrays
    (AvatarModel.cpp:26-104: v_template (V,3), f (F,3), kintree_table (2,J), J_regressor (J,V),
    weights (V,J), shapedirs (V,3,K)) plus the GMM (prior_weight, prior_mean, prior_cov)."""

    def __init__(self, smpl: dict, limit_one_joint_per_point=False):
        """Optional keys of the reference's legacy model format (AvatarModel.cpp:128-288): "joint_shape_reg_base" (3J) and
        "joint_shape_reg" (3J, K) - joint_shape_regressor.txt, taken as given instead of being derived from J_regressor."""
        self.limit_one_joint_per_point = bool(limit_one_joint_per_point)
        v = np.ascontiguousarray(smpl["v_template"], dtype=np.float64)
        self.V = V = v.shape[0]
        self.J = J = int(np.asarray(smpl["kintree_table"]).shape[1])
        sd = np.asarray(smpl["shapedirs"], dtype=np.float64)
        self.K = K = sd.shape[2]
        f = np.asarray(smpl["f"])
        self.F = f.shape[0]
        self.P = 3 + 3 * J + K
        self.base_cloud = v.reshape(-1).copy()                                    # x1 y1 z1 ... (AvatarModel.cpp:46-48)
        self.key_clouds = np.asfortranarray(sd.reshape(3 * V, K))                 # 3V x K col-major (:99-103)
        self.parent = np.asarray(smpl["kintree_table"])[0].astype(np.int32).copy()
        self.parent[0] = -1                                                       # SMPL stores 2^32-1 for the root
        self.mesh = np.ascontiguousarray(f.astype(np.int32))                      # (F,3) rows == 3xF col-major
        W = sp.csc_matrix(np.asarray(smpl["weights"], dtype=np.float64).T)        # J x V (sparseView, :67-71)
        W.sort_indices()
        self.w_colptr = W.indptr.astype(np.int32); self.w_row = W.indices.astype(np.int32)
        self.w_val = W.data.astype(np.float64)
        R = sp.csc_matrix(np.asarray(smpl["J_regressor"], dtype=np.float64).T)    # V x J (:58-63)
        R.sort_indices()
        self.r_colptr = R.indptr.astype(np.int32); self.r_row = R.indices.astype(np.int32)
        self.r_val = R.data.astype(np.float64)
        if "prior_weight" in smpl and smpl["prior_weight"] is not None:
            self.prior_weight = np.ascontiguousarray(smpl["prior_weight"], dtype=np.float64)
            self.prior_mean = np.ascontiguousarray(smpl["prior_mean"], dtype=np.float64)
            self.prior_cov = np.ascontiguousarray(smpl["prior_cov"], dtype=np.float64)
            self.ncomps, self.ndims = self.prior_mean.shape
        else:
            self.prior_weight = self.prior_mean = self.prior_cov = None
            self.ncomps, self.ndims = 0, 0
        if smpl.get("joint_shape_reg") is not None:
            self.jsr_base = np.ascontiguousarray(smpl["joint_shape_reg_base"], dtype=np.float64).reshape(3 * J)
            self.jsr = np.asfortranarray(np.asarray(smpl["joint_shape_reg"], dtype=np.float64).reshape(3 * J, K))   # column-major 3J x K
        else:
            self.jsr_base = self.jsr = None

    def desc(self) -> ModelDesc:
        d = ModelDesc()
        d.num_points, d.num_joints, d.num_shape_keys, d.num_faces = self.V, self.J, self.K, self.F
        d.base_cloud = dptr(self.base_cloud); d.key_clouds = dptr(self.key_clouds)
        d.parent = iptr(self.parent); d.mesh = iptr(self.mesh)
        d.weights_colptr = iptr(self.w_colptr); d.weights_row = iptr(self.w_row); d.weights_val = dptr(self.w_val)
        d.jreg_colptr = iptr(self.r_colptr); d.jreg_row = iptr(self.r_row); d.jreg_val = dptr(self.r_val)
        d.prior_ncomps, d.prior_ndims = self.ncomps, self.ndims
        if self.ncomps > 0:
            d.prior_weight = dptr(self.prior_weight); d.prior_mean = dptr(self.prior_mean)
            d.prior_cov = dptr(self.prior_cov)
        d.limit_one_joint_per_point = 1 if self.limit_one_joint_per_point else 0
        if self.jsr is not None:
            d.joint_shape_reg_base = dptr(self.jsr_base)
            d.joint_shape_reg = self.jsr.ctypes.data_as(c_double_p)
        return d


_lib = None


def load_library():
    """Load libavatar_hip.so (the HIP product path).  Raises if it is missing: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` (hipcc --offload-arch=gfx950). There is deliberately no CPU fallback.")
    try:  # torch bundles its own libamdhip64.so.7; load it first so both share ONE HIP runtime in-process
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    lib.avt_last_error.restype = C.c_char_p
    lib.avt_kernel_name.restype = C.c_char_p
    lib.avt_kernel_name.argtypes = [C.c_int]
    vp = C.c_void_p
    sigs = {
        "avt_options_default": [C.POINTER(Options)],
        "avt_model_create": [C.POINTER(ModelDesc), C.POINTER(vp)],
        "avt_model_destroy": [vp],
        "avt_model_dims": [vp, c_int_p, c_int_p, c_int_p, c_int_p, c_int_p],
        "avt_model_main_joint": [vp, c_int_p],
        "avt_model_joint_regression": [vp, c_double_p, c_double_p],
        "avt_model_tile_layout": [vp, c_int_p, c_int_p, C.POINTER(C.c_ushort), c_int_p],
        "avt_ctx_create": [C.c_int, vp, C.c_int, c_int_p, C.c_int, C.c_int, C.POINTER(vp)],
        "avt_ctx_destroy": [vp],
        "avt_sync": [vp],
        "avt_lbs_update": [vp, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p],
        "avt_visibility": [vp, c_double_p, C.c_int, c_ubyte_p],
        "avt_nn": [vp, c_double_p, c_ubyte_p, c_double_p, c_int_p, C.c_int, c_int_p],
        "avt_optimize": [vp, c_double_p, c_int_p, C.c_int, C.POINTER(Options), c_double_p, c_double_p, c_double_p,
                         C.POINTER(Stats)],
        "avt_optimize_batch": [vp, C.c_int, c_double_p, c_int_p, c_int_p, C.POINTER(Options), c_double_p, c_double_p,
                               c_double_p, C.POINTER(Stats)],
        "avt_frames_upload": [vp, C.c_int, c_double_p, c_int_p, c_int_p],
        "avt_synth_render_frames": [vp, C.c_int, c_double_p, c_double_p, c_double_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, c_int_p],
        "avt_synth_render_frames_mode": [vp, C.c_int, c_double_p, c_double_p, c_double_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int,
                                         C.c_int, c_int_p],
        "avt_synth_render_images": [vp, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_ubyte)],
        "avt_frames_download": [vp, C.c_int, c_double_p, c_int_p],
        "avt_state_upload": [vp, C.c_int, c_double_p, c_double_p, c_double_p],
        "avt_optimize_resident": [vp, C.POINTER(Options)],
        "avt_state_reset": [vp],
        "avt_state_download": [vp, c_double_p, c_double_p, c_double_p, C.POINTER(Stats)],
        "avt_get_correspondences": [vp, C.c_int, c_int_p],
        "avt_get_cloud": [vp, C.c_int, c_double_p],
        "avt_get_posed": [vp, C.c_int, c_double_p, c_double_p, c_double_p],
        "avt_get_normal_equations": [vp, C.c_int, c_double_p, c_double_p, c_double_p],
        "avt_ctx_get_tuning": [vp, C.POINTER(Tuning)],
        "avt_ctx_set_tuning": [vp, C.POINTER(Tuning)],
        "avt_set_data_term": [vp, C.c_int],
        "avt_get_data_term": [vp],
        "avt_debug_trace": [vp, C.c_int, c_double_p],
        "avt_debug_mfma_count": [vp, C.c_int, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)],
        "avt_launch_shape": [vp, c_int_p, c_int_p, c_int_p],
        "avt_profile_begin": [vp],
        "avt_profile_select": [vp, C.c_uint],
        "avt_profile_end": [vp, C.POINTER(Profile)],
        # include/avt_shard.h
        "avt_shard_owner": [C.c_int, C.c_int],
        "avt_shard_local_count": [C.c_int, C.c_int, C.c_int],
        "avt_shard_local_index": [C.c_int, C.c_int],
        "avt_shard_global_frame": [C.c_int, C.c_int, C.c_int],
        "avt_model_pack_size": [C.POINTER(ModelDesc), C.POINTER(C.c_size_t)],
        "avt_model_pack": [C.POINTER(ModelDesc), vp, C.c_size_t],
        "avt_model_unpack": [vp, C.c_size_t, C.POINTER(vp)],
        "avt_shard_unique_id": [C.c_char_p],
        "avt_shard_create": [C.c_int, C.c_int, C.c_int, C.c_char_p, C.POINTER(vp)],
        "avt_shard_create_loopback": [C.c_int, C.c_int, C.c_int, C.c_char_p, C.POINTER(vp)],
        "avt_shard_create_shm": [C.c_int, C.c_int, C.c_int, C.c_char_p, C.POINTER(vp)],
        "avt_shard_destroy": [vp],
        "avt_shard_rank": [vp],
        "avt_shard_world": [vp],
        "avt_shard_backend": [vp],
        "avt_shard_broadcast_model": [vp, C.c_int, C.POINTER(ModelDesc), C.POINTER(vp)],
        "avt_shard_scatter_frames": [vp, vp, C.c_int, C.c_int, c_double_p, c_int_p, c_int_p, c_double_p, c_double_p, c_double_p],
        "avt_shard_gather_enqueue": [vp, vp, C.c_int],
        "avt_shard_gather_wait": [vp],
        "avt_shard_gather_download": [vp, vp, C.c_int, c_double_p, c_double_p, c_double_p, C.POINTER(Stats)],
        "avt_shard_gather_results": [vp, vp, C.c_int, c_double_p, c_double_p, c_double_p, C.POINTER(Stats)],
        "avt_shard_barrier": [vp, vp],
    }
    for name, args in sigs.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.argtypes = args
        if name == "avt_shard_backend":
            fn.restype = C.c_char_p
        elif name not in ("avt_model_destroy", "avt_ctx_destroy", "avt_options_default", "avt_shard_destroy"):
            fn.restype = C.c_int
        else:
            fn.restype = None
    _lib = lib
    return lib


EXPORTED_SYMBOLS = [
    "avt_last_error", "avt_kernel_name", "avt_options_default", "avt_options_fixed_factors", "avt_model_create", "avt_model_destroy",
    "avt_model_dims", "avt_model_main_joint", "avt_model_joint_regression", "avt_model_tile_layout", "avt_ctx_create", "avt_ctx_destroy",
    "avt_sync", "avt_lbs_update", "avt_visibility", "avt_nn", "avt_optimize", "avt_optimize_posed", "avt_optimize_batch",
    "avt_frames_upload", "avt_synth_render_frames", "avt_synth_render_frames_mode", "avt_synth_render_images", "avt_frames_download", "avt_state_upload", "avt_optimize_resident", "avt_state_reset", "avt_state_download",
    "avt_get_correspondences", "avt_get_cloud", "avt_get_posed", "avt_get_normal_equations", "avt_ctx_get_tuning", "avt_ctx_set_tuning", "avt_set_data_term", "avt_get_data_term", "avt_debug_trace", "avt_debug_mfma_count", "avt_launch_shape", "avt_profile_begin", "avt_profile_select", "avt_profile_end",
    # include/avt_shard.h
    "avt_shard_owner", "avt_shard_local_count", "avt_shard_local_index", "avt_shard_global_frame", "avt_model_pack_size", "avt_model_pack",
    "avt_model_unpack", "avt_shard_unique_id", "avt_shard_create", "avt_shard_create_loopback", "avt_shard_create_shm", "avt_shard_destroy", "avt_shard_rank", "avt_shard_world", "avt_shard_backend",
    "avt_shard_broadcast_model", "avt_shard_scatter_frames", "avt_shard_gather_enqueue", "avt_shard_gather_wait", "avt_shard_gather_download",
    "avt_shard_gather_results", "avt_shard_barrier", "avt_shard_set_self_exchange",
]
