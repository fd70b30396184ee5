/* avt.h — C ABI of the MI355X-native SMPL-to-depth fitting engine (libavatar_hip.so).
 *
 * This is the drop-in boundary for the AvatarOptimizer hot path of sxyu/avatar.  The reference has no FFI:
 * the path sits behind two C++ classes (`ark::Avatar`, `ark::AvatarOptimizer`).  Each entry point below
 * names the reference interface it replaces (file:line relative to the reference tree); the C++ facade in
 * include/ark/ re-creates those classes on top of this ABI.
 *
 * Conventions (identical to the reference on the host side of the ABI):
 *   - everything is fp64; indices are int                                  (Avatar.h:19)
 *   - clouds are column-major 3xN, point i at ptr + 3*i                     (AvatarOptimizer.cpp:901)
 *   - rotations are column-major 3x3; quaternion coefficient order (x,y,z,w) (AvatarOptimizer.cpp:295-298)
 *   - labels are ints in [0, num_parts)                                     (demo.cpp:236-243)
 *   - caller owns every host buffer; the library owns all device memory; handles are opaque.
 *   - every function returns 0 on success, non-zero on error; avt_last_error() describes the last failure
 *     on the calling thread.  (The reference's `void` + std::exit(1) asserts live in the C++ facade.)
 *   - a context is NOT thread-safe: one context per host thread / HIP stream   (Avatar.h:191 lifetimes)
 *   - resident state: avt_frames_upload / avt_synth_render_frames make frames resident, avt_state_upload the start
 *     states of exactly those frames; avt_optimize_resident needs both.  Uploading the same NUMBER of frames again keeps
 *     the resident states (temporal warm start); a different number invalidates them.  The stand-alone entry points
 *     avt_lbs_update / avt_visibility / avt_nn use the frame slots as scratch and invalidate frames and states: upload
 *     again before the next avt_optimize_resident (it fails with a message otherwise).
 */
#ifndef AVT_H_
#define AVT_H_

#ifdef __cplusplus
extern "C" {
#endif

/* Status codes (every non-zero return is an error; these two are told apart because a caller may want to): */
#define AVT_STATUS_NO_DEVICE 2      /* avt_ctx_create: no usable HIP device (there is no CPU fallback) */
#define AVT_STATUS_DEVICE_FAULT 3   /* a kernel could not vouch for a frame's result - e.g. a solver of the few-frames launch shape gave up
                                     * waiting for the in-launch reduction (avt_lm.hip) - and said so in the frame's fault word: returned by
                                     * the calls that hand out results (avt_optimize, avt_optimize_batch, avt_state_download,
                                     * avt_shard_gather_download); the word is cleared when it is reported */

#define AVT_MAX_JOINTS 64   /* SMPL: 24, SMPL-H: 52; SMPL-X: 55; additionally 3 + 3J + K <= 179 (avt_model_create) */
#define AVT_MAX_SHAPE 16    /* SMPL: 10 */
#define AVT_MAX_ASSIGN 4    /* AvatarOptimizer.cpp:164 MAX_ASSIGN */
#define AVT_MAX_PARTS 64

typedef struct avt_model avt_model;
typedef struct avt_ctx avt_ctx;

/* Immutable model data, the fields of `struct AvatarModel` the path reads (Avatar.h:64-151, filled at
 * AvatarModel.cpp:35-127).  All pointers are borrowed for the duration of avt_model_create only. */
typedef struct avt_model_desc {
    int num_points;             /* V  = weights.cols()            (Avatar.h:82)  */
    int num_joints;             /* J  = parent.rows()             (Avatar.h:80)  */
    int num_shape_keys;         /* K  = keyClouds.cols()          (Avatar.h:84)  */
    int num_faces;              /* F  = mesh.cols()               (Avatar.h:86)  */
    const double* base_cloud;   /* 3V, x1 y1 z1 x2 ...            (Avatar.h:113) */
    const double* key_clouds;   /* 3V x K column-major            (Avatar.h:117) */
    const int* parent;          /* J, parent[0] = -1, topo-sorted (Avatar.h:97, AvatarModel.cpp:41) */
    const int* mesh;            /* 3 x F column-major             (Avatar.h:94)  */
    /* LBS weights, J x V sparse, compressed column (one column per vertex)      (Avatar.h:142) */
    const int* weights_colptr;  /* V+1 */
    const int* weights_row;     /* nnz, joint ids, ascending within a column */
    const double* weights_val;  /* nnz */
    /* joint regressor, V x J sparse, compressed column (one column per joint)   (Avatar.h:124) */
    const int* jreg_colptr;     /* J+1 */
    const int* jreg_row;        /* nnz, vertex ids, ascending within a column */
    const double* jreg_val;     /* nnz */
    /* Gaussian-mixture pose prior, pose_prior.txt contents (GaussianMixture.cpp:20-58); ncomps<=0: none */
    int prior_ncomps;
    int prior_ndims;            /* must be 3*(J-1) */
    const double* prior_weight; /* ncomps */
    const double* prior_mean;   /* ncomps x ndims row-major */
    const double* prior_cov;    /* ncomps x ndims x ndims, each matrix row-major as in the text file */
    /* ---- the reference's legacy ("ad-hoc") model format only (AvatarModel.cpp:128-288); zero / NULL for model.npz ---- */
    int limit_one_joint_per_point;      /* AvatarModel(dir, limit_one_joint_per_point = true) (Avatar.h:76-77, AvatarModel.cpp:190-196):
                                         * assignedJoints[v] keeps its largest weight only, set to 1 - what the OPTIMISER's forward model
                                         * and Jacobians use; `weights`, i.e. Avatar::update(), keeps every entry */
    int reserved1;
    const double* joint_shape_reg_base; /* 3J: jointShapeRegBase read from joint_shape_regressor.txt (AvatarModel.cpp:231-243) instead of */
    const double* joint_shape_reg;      /* 3J x K column-major: jointShapeReg   being derived from the joint regressor (:112-127); both or neither */
} avt_model_desc;

/* Knobs: the public data members of AvatarOptimizer (AvatarOptimizer.h:25-39) and the arguments of
 * optimize() (AvatarOptimizer.h:17-19), plus the Gauss-Newton/LM step rule that replaces the reference's
 * Ceres BFGS line search (AvatarOptimizer.cpp:1313-1341, :1486; see DESIGN.md "step rule"). */
typedef struct avt_options {
    double beta_pose;           /* betaPose, reference default 0.1; demos use 0.05 (demo.cpp:54-57) */
    double beta_shape;          /* betaShape, reference default 1.0; demos use 0.12 */
    int nn_step;                /* nnStep = 20; unused in the inverted NN mode the reference runs (:933,:1393) */
    int max_iters_per_icp;      /* maxItersPerICP = 10: GN iterations per ICP iteration */
    int enable_occlusion;       /* enableOcclusion = true: back-face visibility (AvatarOptimizer.cpp:1349-1367) */
    int icp_iters;              /* optimize(..., icp_iters = 1, ...) */
    int num_threads;            /* optimize(..., num_threads = 4): accepted, ignored on the GPU path */
    int lm_policy;              /* damping schedule: AVT_LM_GAIN_RATIO (1, the default since round 6: Nielsen's gain-ratio update, the better
                                   optimiser on every bench seed) or AVT_LM_FIXED_FACTORS (0: lm_up / lm_down as plain factors); DESIGN.md section 4 */
    double lm_lambda0;          /* initial damping (relative to diag H); default 1e-3 */
    double lm_up;               /* > 1.  gain ratio: the multiplier of the FIRST rejection after an accepted step (it doubles with every further
                                   one); default AVT_LM_UP_GAIN_RATIO = 16 (Nielsen's own constant is 2).  fixed factors: the damping multiplier
                                   on every rejected step; AVT_LM_UP_FIXED_FACTORS = 4 is the value that schedule was tuned with */
    double lm_down;             /* in (0, 1); default 1/3.  gain ratio: the floor of the multiplier max(lm_down, 1 - (2 rho - 1)^3) an accepted
                                   step applies (Nielsen's 1/3).  fixed factors: the damping multiplier on an accepted step */
    double lm_lambda_min;       /* default 1e-12 */
    double lm_lambda_max;       /* default 1e8 */
    double function_tolerance;  /* >= 0, < 1.  The reference's stopping rule, options.function_tolerance = 1e-4 (AvatarOptimizer.cpp:1333; Ceres'
                                   line-search minimiser stops when |cost change| <= function_tolerance x cost): an ACCEPTED step whose decrease is
                                   at most this fraction of the objective it started from ends the Gauss-Newton iterations of that ICP iteration
                                   for that frame - the remaining launches of the iteration are idle for it, avt_stats.gn_iterations counts what
                                   ran.  Default 1e-4 as in the reference; 0 = always max_iters_per_icp iterations (what bench.py's headline counts). */
} avt_options;
enum { AVT_LM_FIXED_FACTORS = 0, AVT_LM_GAIN_RATIO = 1 };
#define AVT_LM_UP_GAIN_RATIO 16.0
#define AVT_LM_UP_FIXED_FACTORS 4.0

typedef struct avt_stats {
    double initial_cost;        /* 0.5*sum r^2 at entry to the last ICP iteration (data + priors) */
    double final_cost;          /* same objective after the last accepted step */
    double lambda;              /* damping at exit */
    int num_correspondences;    /* ICP residual blocks in the last ICP iteration (totalResiduals, :1441-1451) */
    int matched_model_points;   /* model points with >= 1 correspondence (caches, :1419-1431) */
    int gn_iterations;          /* GN iterations executed over all ICP iterations (fewer than icp_iters x max_iters_per_icp when function_tolerance ended some early) */
    int accepted_steps;
} avt_stats;

/* Per-kernel-class device timings (ms, HIP events on the context's stream) accumulated between
 * avt_profile_begin / avt_profile_end.  Profiling inserts event records around every launch. */
enum { AVT_K_LBS = 0, AVT_K_VISIBILITY, AVT_K_BUCKET, AVT_K_NN, AVT_K_AGGREGATE, AVT_K_PREPARE,
       AVT_K_EVAL, AVT_K_REDUCE, AVT_K_SOLVE /* the full LM solves */, AVT_K_DECIDE /* the closing accept/reject */,
       AVT_K_MOMENTS /* moment form: k_moments, once per ICP iteration */, AVT_K_COUNT };
typedef struct avt_profile {
    double ms[AVT_K_COUNT];
    int launches[AVT_K_COUNT];
} avt_profile;

const char* avt_last_error(void);
const char* avt_kernel_name(int kernel_class);
void avt_options_default(avt_options* o);      /* reference defaults (AvatarOptimizer.h:28-39, function_tolerance :1333) + the step rule's: the ONE place
                                                * the defaults are set (gain-ratio schedule, lm_up 16, lm_down 1/3); avatar_amd/capi.py mirrors it and a test compares */
void avt_options_fixed_factors(avt_options* o); /* the same with the fixed-factor schedule of rounds 1-5 (lm_policy 0, lm_up 4) */

/* ---- model: replaces `AvatarModel::AvatarModel` data preparation (AvatarModel.cpp:74-127) and the
 * pose-independent parts of `AvatarEvaluationCommonData` (AvatarOptimizer.cpp:187-245) and
 * `GaussianMixture::load` factorisations (GaussianMixture.cpp:44-76). */
int avt_model_create(const avt_model_desc* desc, avt_model** out);
void avt_model_destroy(avt_model* m);
int avt_model_dims(const avt_model* m, int* V, int* J, int* K, int* F, int* P);
/* main (largest-weight) joint of every vertex: assignedJoints[v][0].second (Avatar.h:101) */
int avt_model_main_joint(const avt_model* m, int* main_joint_V);
/* initialJointPos (3xJ) and jointShapeReg (3J x K col-major) (AvatarModel.cpp:112-127) */
int avt_model_joint_regression(const avt_model* m, double* initial_joint_pos_3xJ, double* joint_shape_reg_3JxK);
/* Column layout of the evaluation kernel's [J | r] tile (no reference counterpart; the parameter blocks are those of
 * AvatarOptimizer.cpp:620-629): `tile_param[16*ntiles]` = parameter index of every tile column (P = residual, -1 = padding),
 * `vertex_tiles[V]` = bit mask of the 16-column tiles a vertex's residual rows touch, `vertex_order[V]` = the order in which
 * matched vertices are batched (by tile set, then id).  Any pointer may be NULL; *ntiles = ceil((P+1)/16). */
int avt_model_tile_layout(const avt_model* m, int* ntiles, int* tile_param, unsigned short* vertex_tiles, int* vertex_order);

/* ---- context: one HIP device + stream + persistent buffers.  `part_map` (>= J entries) and `num_parts`
 * are the AvatarOptimizer ctor arguments (AvatarOptimizer.h:14, AvatarOptimizer.cpp:1213-1244). */
int avt_ctx_create(int device, const avt_model* m, int num_parts, const int* part_map,
                   int max_points_per_frame, int max_frames, avt_ctx** out);
void avt_ctx_destroy(avt_ctx* c);
int avt_sync(avt_ctx* c);

/* ---- Avatar::update() (Avatar.cpp:22-75) for `nframes` independent avatars.
 * w: K x nframes, p: 3 x nframes, R: 9J x nframes (J column-major 3x3 blocks per frame).
 * Outputs (any may be NULL): cloud 3V x nframes, joint_pos 3J x nframes, joint_trans 12J x nframes. */
int avt_lbs_update(avt_ctx* c, int nframes, const double* w, const double* p, const double* R,
                   double* cloud, double* joint_pos, double* joint_trans);

/* ---- back-face visibility (AvatarOptimizer.cpp:1342-1367). visible: V bytes (0/1). */
int avt_visibility(avt_ctx* c, const double* cloud_3xV, int enable_occlusion, unsigned char* visible);

/* ---- findNN(..., invert=true) (AvatarOptimizer.cpp:841-907): for each data point the exact nearest
 * visible model point of its own part; -1 where the part has no visible model point (:899). */
int avt_nn(avt_ctx* c, const double* model_cloud_3xV, const unsigned char* visible,
           const double* data_3xN, const int* labels, int N, int* model_idx_out);

/* ---- AvatarOptimizer::optimize() (AvatarOptimizer.cpp:1246-1517), one frame.
 * In/out: p (3), q (4 x J, xyzw), w (K).  The caller converts ava.r <-> q (AvatarOptimizer.cpp:1250-1254,
 * :1494-1496; done by the C++ facade).  stats may be NULL. */
int avt_optimize(avt_ctx* c, const double* data_3xN, const int* labels, int N, const avt_options* opt,
                 double* p, double* q, double* w, avt_stats* stats);

/* ---- the same, returning what the update() that ends optimize() left (AvatarOptimizer.cpp:1494-1497: ava.cloud 3 x V, ava.jointPos 3 x J,
 * ava.jointTrans 12 x J; any of the three may be NULL) in the SAME synchronisation as the fit - what ark::AvatarOptimizer::optimize binds: one
 * call, one wait, instead of avt_optimize + avt_get_posed (three more copies to pageable memory and a second wait).  Needs icp_iters >= 1. */
int avt_optimize_posed(avt_ctx* c, const double* data_3xN, const int* labels, int N, const avt_options* opt,
                       double* p, double* q, double* w, avt_stats* stats, double* cloud_3xV, double* joint_pos_3xJ, double* joint_trans_12xJ);

/* ---- batch of independent frames (one optimize() each).  frame f owns points
 * [frame_offsets[f], frame_offsets[f+1]) of data/labels; p/q/w/stats are per-frame, frame-major. */
int avt_optimize_batch(avt_ctx* c, int nframes, const double* data, const int* labels,
                       const int* frame_offsets, const avt_options* opt,
                       double* p, double* q, double* w, avt_stats* stats);

/* ---- the same, split so that inputs can be resident in HBM before a timed region starts:
 * upload (H2D + nothing else), run (asynchronous on the context's stream), download (syncs). */
int avt_frames_upload(avt_ctx* c, int nframes, const double* data, const int* labels, const int* frame_offsets);
int avt_state_upload(avt_ctx* c, int nframes, const double* p, const double* q, const double* w);
int avt_optimize_resident(avt_ctx* c, const avt_options* opt);
/* Re-installs, on the device and asynchronously, the start state of the last avt_state_upload (a device-side copy is kept):
 * a caller that fits the same resident frames repeatedly from the same start (benchmarks, multi-hypothesis restarts)
 * enqueues avt_state_reset + avt_optimize_resident without any host-to-device transfer or host synchronisation. */
int avt_state_reset(avt_ctx* c);
int avt_state_download(avt_ctx* c, double* p, double* q, double* w, avt_stats* stats);

/* ---- synthetic-frame generator on the GPU: AvatarRenderer::renderDepth / renderPartMask
 * (AvatarRenderer.cpp:72-101, :174-202) of `nframes` posed avatars + back-projection (Calibration.cpp:68-74, y negated as
 * optim.cpp:116-119), written straight into the context's resident frame buffers (as if by avt_frames_upload).
 * w: K x nframes, p: 3 x nframes, R: 9J x nframes; points_per_frame (nframes ints, may be NULL) receives N of each frame.
 * avt_frames_download copies one resident frame back (data 3 x N doubles, labels N ints). */
int avt_synth_render_frames(avt_ctx* c, int nframes, const double* w, const double* p, const double* R, double fx, double fy,
                            double cx, double cy, int width, int height, int* points_per_frame);
/* The same with the visibility rule chosen by the caller:
 *   AVT_RENDER_ZBUFFER  nearest surface per pixel (the fast generator the benchmarks use; inside-triangle coverage);
 *   AVT_RENDER_PAINTER  the reference's renderer pixel for pixel: faces painted by decreasing mean depth, later faces
 *                       overwrite (AvatarRenderer.cpp:39-70), renderDepth's row scanline fill with the floored / ceiled end
 *                       vertices (AvatarHelpers.cpp:61-139), renderPartMask's column fill and nearest-vertex rule
 *                       (AvatarHelpers.cpp:153-245), edge-on faces painting 0 / 255 with the end-exclusive fill
 *                       (AvatarHelpers.cpp:247-303); every pixel with depth > 0 becomes a point (optim.cpp:104-120) labelled by
 *                       the part mask at that pixel (255 where the two fills disagree about coverage: such points carry a
 *                       label outside [0, num_parts) and optimize() drops them).  Intrinsics are rounded to float first
 *                       (Calibration.h:13).  Equal face sort keys are ordered by face id (std::sort leaves them unspecified).
 * After a PAINTER call whose frames fit one scratch chunk, avt_synth_render_images copies out what renderDepth (H x W float32,
 * 0 = background) and renderPartMask (H x W uint8, 255 = background) return for frame `frame`; either pointer may be NULL. */
enum { AVT_RENDER_ZBUFFER = 0, AVT_RENDER_PAINTER = 1 };
int avt_synth_render_frames_mode(avt_ctx* c, int nframes, const double* w, const double* p, const double* R, double fx, double fy,
                                 double cx, double cy, int width, int height, int mode, int* points_per_frame);
int avt_synth_render_images(avt_ctx* c, int frame, float* depth_HxW, unsigned char* part_mask_HxW);
int avt_frames_download(avt_ctx* c, int frame, double* data_3xN, int* labels);

/* ---- introspection of the last optimize call (tests / diagnostics) */
int avt_get_correspondences(avt_ctx* c, int frame, int* model_idx_out /* N of that frame */);
int avt_get_cloud(avt_ctx* c, int frame, double* cloud_3xV);       /* ava.cloud after the final update() */
/* ava.cloud, ava.jointPos (3 x J) and ava.jointTrans (12 x J) as the update() that ends optimize() left them
 * (AvatarOptimizer.cpp:1494-1497, Avatar.cpp:22-75): a caller refreshes its Avatar from these instead of running
 * update() a second time.  Any pointer may be NULL. */
int avt_get_posed(avt_ctx* c, int frame, double* cloud_3xV, double* joint_pos_3xJ, double* joint_trans_12xJ);
/* data-term Gauss-Newton normal equations J^T J (P x P) and J^T r (P) at the current point + its objective.  Evaluated by
 * this call (optimize() itself never builds the system of its last trial point) with the correspondences of the last ICP
 * iteration, for all resident frames; the fit, its cost and the LM state are not changed. */
int avt_get_normal_equations(avt_ctx* c, int frame, double* H /* P x P */, double* g /* P */, double* cost);

/* How the ICP data term of a Gauss-Newton iteration is evaluated (same objective, same normal equations up to rounding):
 *   AVT_DATA_TERM_ROWS     the residual / Jacobian rows of every matched model point are rebuilt and contracted on the matrix cores
 *                          every iteration (AvatarCostFunctorCache::updateData + the ICP cost functor, AvatarOptimizer.cpp:505-644);
 *   AVT_DATA_TERM_MOMENTS  the correspondences' sufficient statistics are accumulated once per ICP iteration and every iteration
 *                          contracts them with the state (avt_moments.hip);
 *   AVT_DATA_TERM_AUTO     (default) the moment form from 8 frames per launch on (avt_tuning.mom_min_frames), where it is the faster one; rows below.
 * Takes effect for the following calls; avt_get_normal_equations evaluates with the form selected here (AUTO: the form the last
 * optimize() ran), so tests can compare the two on the same correspondences. */
enum { AVT_DATA_TERM_ROWS = 0, AVT_DATA_TERM_MOMENTS = 1, AVT_DATA_TERM_AUTO = 2 };
int avt_set_data_term(avt_ctx* c, int form);
int avt_get_data_term(avt_ctx* c);

/* diagnostics: 64 doubles per frame (objective after every GN iteration; with -DAVT_TIMING builds also in-kernel
 * s_memtime probes, see tools/kernel_timing_probe.py) */
int avt_debug_trace(avt_ctx* c, int frame, double* out64);

/* diagnostics: fp64 matrix instructions (v_mfma_f64_16x16x4_f64, 2048 flop each) the kernels EXECUTE for frame `frame` with the
 * correspondences of the last optimize(), from the kernels' own trip counts (block-sparse: dead tile pairs are skipped):
 *   eval_rows   one full row-form evaluation (k_eval; -1 when the last optimize() ran the moment form or the skeleton is not the six-tile shape),
 *   moments     one k_moments pass (once per ICP iteration; -1 when the model has no moment form),
 *   solve       one LDL^T factorisation of k_solve.
 * Any pointer may be NULL.  bench.py divides these by the measured launch times for the matrix-pipe utilisation it reports. */
int avt_debug_mfma_count(avt_ctx* c, int frame, long long* eval_rows, long long* moments, long long* solve);

/* Launch-shape and algorithm knobs of a context.  The defaults are the measured optima (DESIGN.md sections 5 and 7); they exist for
 * experiments and for tests that must force a particular code path.  avt_ctx_create fills the structure from the defaults and then
 * ONCE from the environment (AVT_<FIELD NAME IN UPPER CASE>, e.g. AVT_NSPEC=0; unknown AVT_* names are reported on stderr): nothing in
 * the library reads the environment afterwards, and what a context runs with can be read back and recorded.  avt_ctx_set_tuning
 * validates, drops the cached launch graphs and takes effect for the following calls. */
typedef struct avt_tuning {
    int use_graph;           /* 1: optimize() replays one hipGraph per launch shape; 0: plain launches (AVT_NO_GRAPH=1 sets 0) */
    int groups;              /* frame groups (streams) of one optimize(); 0 = automatic (2 from 44 frames on, one frame per group for 2-3 frames) */
    int g;                   /* row form: evaluation workgroups per frame, 0 = automatic */
    int gcap;                /* ... and their cap in the few-frames shape (128) */
    int vis_frame_min;       /* frames per launch from which visibility runs as one workgroup per frame (32: 64 frames per GPU 1.013 -> 1.003 ms against 64, the round-4 value; 0 = never) */
    int ride;                /* 1: up to three frames the reduction rides in k_solve's launch */
    int ride_strips;         /* 0 = automatic, else 4 or 8 strips per tile pair */
    int ride_sizing_groups;  /* 1: size the riding shapes by the frame groups running side by side instead of by the one launch */
    int nspec;               /* speculative solver workgroups per frame beside the solver (0 .. 4) */
    int nn_force_part;       /* 1: the throughput shape of the nearest neighbour on small inputs too */
    int nn_slab;             /* 1: that shape walks y-sorted candidates outwards from the wave's slab of queries; 0: full scan */
    int mom_min_frames;      /* AVT_DATA_TERM_AUTO: frames per launch from which the moment form is used (8: the measured crossover of round 5; 32 in round 4) */
    int debug;               /* 1: occupancy report on stderr at context creation */
    int asm_parts;           /* moment form: 1 (default) the assembly of a frame runs as six independent 256-thread role workgroups, 0 as one 1024-thread workgroup */
    long long ride_timeout_us; /* how long a solver role waits for the riding reduction before it raises the frame's fault (2 000 000) */
    int lbs_frames;          /* frame batches: frames skinned per workgroup (k_lbs_multi); 0 = automatic (4 from 128 frames per launch on, 2 from 64), 1 = one (k_lbs), 2, 4 */
    int spec_cost;           /* riding shapes: 1 (default) the COST of the first queued speculative LM step is evaluated beside the trial point, so that the solve
                              * launch that rejects the trial point takes that step's accept test as well when it fails too - two rejections in one launch pair
                              * (DESIGN section 4); 0: one launch pair per rejection */
    int xcd_frames;          /* frame-batch kernels whose workgroups share per-frame data (k_moments, the nearest neighbour's throughput shape): 1 (default) the
                              * workgroups of a frame run on ONE of the eight XCDs (grid remap, xcd_frame_block), so that the frame's data is fetched into one
                              * L2 instead of eight; 0: grid order */
    int literal_dims;        /* 1 (default): a model with SMPL's dimensions (24 joints, 10 shape keys, a 69-dimensional pose prior) runs the copies of
                              * k_solve, k_pairpass and k_prior that are compiled for them (every count a literal: DESIGN section 5, "Round 6, second
                              * half"); 0: the run-time-dimension copies every other model runs - same results to rounding, ~25 % slower per solve */
} avt_tuning;
int avt_ctx_get_tuning(avt_ctx* c, avt_tuning* out);
int avt_ctx_set_tuning(avt_ctx* c, const avt_tuning* t);

/* How optimize() will launch over the resident frames: the batch runs as `groups` frame groups of `frames_per_group` frames
 * (each kernel of the sequence is launched once per group, on the group's own stream when replayed as a hipGraph), the
 * evaluation kernel with `eval_workgroups_per_frame` workgroups per frame.  Lets a profiler label launches by shape. */
int avt_launch_shape(avt_ctx* c, int* groups, int* frames_per_group, int* eval_workgroups_per_frame);

int avt_profile_begin(avt_ctx* c);
/* restrict event insertion to the kernel classes in `mask` (bit k = class k); default: all classes */
int avt_profile_select(avt_ctx* c, unsigned mask);
int avt_profile_end(avt_ctx* c, avt_profile* out);

#ifdef __cplusplus
}
#endif
#endif /* AVT_H_ */
