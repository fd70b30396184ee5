// avt_internal.h — private data layout of libavatar_hip.so (host + device), gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/avt.h"

#define AVT_ANC_MAX 16        // max deduplicated ancestors per skin point (SMPL needs <= 12)
#define AVT_TILE 16           // MFMA f64 16x16x4 tile edge
#define AVT_EVAL_PTS 16       // model points per eval batch (48 Jacobian rows)
#define AVT_EVAL_ROWS (3 * AVT_EVAL_PTS)
#define AVT_EVAL_RS 49        // LDS row stride (doubles) of the transposed Jacobian tile.  ODD on purpose: the compiler pairs the MFMA
                              // operand fetches of two k-steps into ds_read2_b64, which is banked modulo 32 in 16-lane groups - an even
                              // stride makes columns c and c+8 collide (2-way), an odd one is conflict-free (plain ds_read_b64 too)
#define AVT_EVAL_TILE(ncols) ((((ncols) * AVT_EVAL_RS) + 1) & ~1)   // doubles of a tile of ncols columns, kept even (16-byte alignment of what follows)
#define AVT_ERANGE 66         // G + 1 entries per frame for G < 64 (FrameBuffers::erange)
#define AVT_MAX_TILES 12      // ceil((P+1)/16) with P <= 179
#define AVT_MAX_P 179         // k_solve<1024, true>: the packed factor of the bordered system (45 x 46 / 2 blocks of 144 B) must fit the LDS
#define AVT_MAX_COMPS 16      // GMM components
#define AVT_G_MAX 128         // most evaluation workgroups a frame can have (k_reduce_strip, avt_decide.h)
#define AVT_MAX_GROUPS 4      // frame groups of one optimize() running on separate streams
#define AVT_PRIOR_STRIDE (2 + 3 * AVT_MAX_JOINTS)   // doubles per (frame, component) of prior scratch
#define AVT_FIX_SCALE 1099511627776.0  // 2^40 fixed-point scale of the centred correspondence sums

// ---- per-frame state block (doubles), double-buffered: slot 0/1 --------------------------------
// x = (p[3], q[4J], w[K])
struct AvtDims {
    int V, J, K, F, P;       // P = 3 + 3J + K
    int NT;                  // column tiles of the augmented matrix [J | r]: ceil((P+1)/16)
    int NPAIR;               // NT*(NT+1)/2 upper-triangular tile pairs
    int xsize;               // 3 + 4J + K
    int prep_size;           // doubles per prep block
    int num_parts;
    int anc_max;             // actual max #ancestors in this model
    int ncomps, ndims;       // GMM
    int nlevels;             // depth of the kinematic tree + 1
    int fk_reg;              // the skeleton pass's work items also exist as DeviceModel::fk_titems (<= AVT_PREP_LEVELS_REG levels of <= AVT_PREP_TITEM_THREADS items)
    int HS;                  // row stride of the dense normal-equation block: 4*ceil((P+1)/4)
    int rec_quad;            // doubles per 4-point matched-point record (avt_eval.hip): 12K + 84
    int nb_max;              // eval batches a frame can have: ceil(V/16)
    int col_tr, col_shape, col_res;   // storage columns of the evaluation tile: root translation, first shape key, residual (avt_model.cpp)
    int res_tile, res_pair;  // the column tile that holds the residual column and the index of its diagonal tile pair
    int res_elem;            // element of that pair's 16x16 tile (k_eval's partial-tile layout) that holds sum c|r|^2
    // six-tile evaluation (SMPL shape): which wave contracts which tile pair - pair_deal[wave] (five 5-bit pair indices each; 20 pairs), and the pair
    // whose 12 k-steps are split over the four waves; dealt on the host so that the waves' expected loads are even (avt_model.cpp)
    unsigned pair_deal[4];   // wave w: five 5-bit pair indices, pair i at bits 5i..5i+4; bit 25+i: pair i is diagonal (i = 5: the split pair)
    int pair_split;
    unsigned long long tile_zpass[AVT_MAX_TILES];   // per tile: the 5-column zeroing passes of build_rows that overlap its storage columns (bit = pass)
    // moment form of the data term (avt_moments.hip; build_moment_tables, avt_model.cpp)
    int mom_ok;              // the model can run it (K + 1 <= 16 lanes of a pair group, 256-thread solve)
    int mom_np;              // unordered co-assigned joint pairs (k <= k')
    int mom_npsi;            // 3 (K + 1) + 1: entries of psi_m = [base | keys | 1], index (K+1) i + s
    int mom_ntp;             // ceil(mom_npsi / 16): 16-row tiles of psi
    int mom_nm1;             // non-empty (k, j') entries of the rot-rot stage 1
    int mom_nb2, mom_nz2;    // (j <= j') rot-rot blocks with / without ordered pairs under them
    int mom_lmax;            // longest per-pair vertex list
    int mom_nopk, mom_nsub, mom_nm1l, mom_ns2l;   // lengths of the index lists mom_opk, mom_sub, mom_m1, mom_s2
    int mom_nseg;            // 16-entry segments of the rot-rot lists
    int mom_toff[8];         // word offsets of the lists inside DeviceModel::mom_tab16 (opk_start, opk, sub_start, sub, bseg, seg, jj, end)
    int mom_rr_doubles;      // ... and the LDS doubles the largest of them needs for its segment and block sums
    int mom_rsplit[5];       // k_assemble_parts: rot-rot role r takes the listed blocks [mom_rsplit[r], mom_rsplit[r + 1]) - equal numbers of segments (MOM_ASM_NR <= 4 roles)
};

// prep block layout (doubles), one per frame per slot: what an evaluation needs about the skeleton state
//   Rw[J][9] row-major world rotations R(-1,j)        (AvatarOptimizer.cpp:303-315)
//   o[J][3]  world joint origins t(-1,j)
//   Jh[J][3] jointPosInit, root-subtracted             (AvatarOptimizer.cpp:249-281)
//   G[J][3][K] = H[j] - Rw[j]*S[j]                     (AvatarOptimizer.cpp:318-324, :568-580)
//   q[J][4], w[K], off[3] (root joint offset), pad
__host__ __device__ inline int prep_off_Rw(const AvtDims& d) { return 0; }
__host__ __device__ inline int prep_off_o(const AvtDims& d) { return 9 * d.J; }
__host__ __device__ inline int prep_off_Jh(const AvtDims& d) { return 12 * d.J; }
__host__ __device__ inline int prep_off_G(const AvtDims& d) { return 15 * d.J; }
__host__ __device__ inline int prep_off_q(const AvtDims& d) { return 15 * d.J + 3 * d.J * d.K; }
__host__ __device__ inline int prep_off_w(const AvtDims& d) { return 19 * d.J + 3 * d.J * d.K; }
__host__ __device__ inline int prep_off_off(const AvtDims& d) { return 19 * d.J + 3 * d.J * d.K + d.K; }
__host__ __device__ inline int prep_total(const AvtDims& d) { return ((19 * d.J + 3 * d.J * d.K + d.K + 3) + 7) & ~7; }

// LDS scratch of k_solve's skeleton pass (avt_lm.hip): offsets in doubles; the host-built work items (DeviceModel::
// fk_items) address it with 14-bit offsets
struct PrepLayout {
    int rot, Rw, o, jp, dv, H, Sp, S, jsr, jsrb, ident, zero, w, x0;
    int ndoubles;   // even
    int nitems;     // J*(12+3K)
};
__host__ __device__ inline PrepLayout prep_layout(int J, int K, int xsize) {
    PrepLayout L;
    int o = 0;
    L.rot = o; o += 9 * J;
    L.Rw = o; o += 9 * J;
    L.o = o; o += 3 * J;
    L.jp = o; o += 3 * J;
    L.dv = o; o += 3 * J;
    L.H = o; o += 3 * J * K;
    L.Sp = o; o += 3 * J * K;
    L.S = o; o += 3 * J * K;
    L.jsr = o; o += 3 * J * K;
    L.jsrb = o; o += 3 * J;
    L.ident = o; o += 9;
    L.zero = o; o += 3;
    L.w = o; o += K;
    L.x0 = o;                 // (unused since round 2: both state slots are staged in their own LDS area)
    L.ndoubles = (o + 1) & ~1;
    L.nitems = J * (12 + 3 * K);
    return L;
}

#define AVT_PREP_LEVELS_REG 10      // tree levels whose work items a thread of the skeleton pass keeps in registers (SMPL: 9; deeper trees / wider levels: items from LDS, level by level)
#define AVT_PREP_TITEM_THREADS 256

// per-frame scalar control block
struct AvtFrameCtl {
    double lambda;
    double cost_cur;          // objective of the current state (data part uses centred form + cost_const)
    double cost_const;        // 0.5 * sum_i |d_i - dbar_m(i)|^2 for the current correspondences
    double cost_initial;
    double sbp, sbs;          // scaledBetaPose / scaledBetaShape (AvatarOptimizer.cpp:1457-1458)
    double centre[3];         // fixed-point centring offset of this frame's data
    int cur_slot;             // which state/H slot holds the current point
    int try_valid;            // 1: the trial point is a real LM step (the factorisation succeeded); 0: it is not (no accept test); AVT_TRY_DONE (2): the
                              // frame met the stopping rule (avt_options::function_tolerance) - no trial point, every further launch of this ICP iteration is idle for it
    int M;                    // matched model points
    int T;                    // total correspondences
    int gn_iterations;
    int accepted;
    int comp_cur;             // GMM component chosen at the current point
    int N;                    // data points of this frame
    // what the accept test of the LAST trial point of an ICP iteration reads: cur_slot, try_valid, cost_cur and lambda as the last
    // k_solve left them.  Every workgroup of the k_lbs launch that follows takes that decision for itself while one of them
    // rewrites the fields above, so the inputs live in fields nobody writes during that launch.
    int dec_cur_slot, dec_try_valid;
    double dec_cost_cur, dec_lambda;
    // gain-ratio damping schedule (avt_options::lm_policy = 1): the decrease the quadratic model predicts for the trial point,
    // 1/2 delta^T (lambda D delta - g), and the factor the next rejection multiplies lambda by (Nielsen: 2, 4, 8 .. in a run of
    // rejections); dec_*: the copies the accept test of the last trial point reads
    double pred, nu, dec_pred, dec_nu;
};

// the knobs of optimize() the kernels read (avt_options), kept in device memory so that they are not baked into the
// captured launch sequence: a tracker that varies betaPose / the LM scalars replays the same hipGraph
// Speculative LM steps (avt_lm.hip): a rejected trial point is followed by a solve of the SAME system with a larger lambda, so
// the solve launch factors that system for lambda, lambda up, lambda up^2 .. side by side (one workgroup each); after a rejection
// the next launch installs the step that is already there instead of factoring again.
#define AVT_MAX_SPEC 4
#define AVT_SPEC_FRAMES 4     // frames per context that can have their speculative steps' costs evaluated ahead (one-frame launch shapes: up to three frames)
struct AvtSpecCtl {
    int next, n;                   // next speculative step to use, how many the last full solve launch made
    int valid[AVT_MAX_SPEC];       // its factorisation succeeded
    double lambda[AVT_MAX_SPEC];   // the damping it was made with (= what the accept test would have set)
    double pred[AVT_MAX_SPEC];     // the decrease its quadratic model predicts (gain-ratio schedule)
    int ahead, pad;                // accept tests taken AHEAD of their launches in this ICP iteration (folded rejections, avt_lm.hip): the solve launch `seq` takes test number seq - 1 + ahead
};

// What the solver roles of a riding k_solve launch decide on (avt_lm.hip): the control block, the speculative-step queue and the
// shape coefficients of the trial point as the evaluation launch in front of it saw them.  The solver rewrites the live copies
// while the launch runs; a speculative workgroup that starts late (another stream's work on its CU) must still see the inputs
// the solver saw, so every role reads this snapshot, which nobody writes during the launch.
struct AvtSolveSnap {
    AvtFrameCtl ctl;
    AvtSpecCtl sp;
    double xw[AVT_MAX_SHAPE];
};

// FrameBuffers::fault bits
#define AVT_FAULT_NOT_RESIDENT 2u   // (batch split) the owning rank's context did not hold the frame when the results were gathered
#define AVT_FAULT_RIDE_TIMEOUT 1u   // a solver role of a riding k_solve launch gave up waiting for the reduction workgroups of its launch

#define AVT_TRY_DONE 2
struct AvtRunParams {
    double beta_pose, beta_shape, lambda0, lm_up, lm_down, lm_min, lm_max, lm_policy;      // lm_policy: avt_options::lm_policy as a double (0 / 1)
    double ftol, pad;                                                                       // avt_options::function_tolerance (0 = off)
};

struct DeviceModel {
    AvtDims d;
    // shape planes [(K+1)*3][V]: plane k*3+c = keyClouds component c of key k; planes K*3+c = baseCloud
    double* shape_planes;
    // LBS weights exactly as the sparse matrix (CSC order, all nnz, <=4): Avatar::update (Avatar.cpp:69)
    double* lbs_w;   // [4][V]
    int* lbs_j;      // [4][V]
    // assignedJoints (weight > 1e-12, sorted desc): the optimiser's forward model (AvatarOptimizer.cpp:508-514)
    double* asg_w;   // [4][V], 0-padded
    int* asg_j;      // [4][V], padded with joint 0
    // ancestors: anc_n[V]; anc[a][V] = jid | (mask<<8), mask bit t set if assigned joint t lies under jid
    unsigned char* anc_n;
    unsigned short* anc;  // [AVT_ANC_MAX][V]
    int* mesh;            // [3][F] SoA
    int* parent;          // [J]
    int* jlevel;          // [J] depth of the joint in the kinematic tree (root 0)
    // column layout of the evaluation tile (build_tile_layout, avt_model.cpp)
    int* tile_col;        // [16*NT] tile column -> storage column (P+1 = the all-zero column for padding)
    int* deal_col;        // [4 waves][6 pairs][2 operands][16] six-tile shape: storage columns of the A / B fragments of the wave's dealt pairs (5) and the split pair
    int* tile_param;      // [16*NT] tile column -> parameter index (P = residual), -1 = padding
    int* joint_col;       // [J] storage column of the joint's first rotation parameter
    int* vorder;          // [V] vertices ordered by the set of tiles their rows touch, then by id
    unsigned short* vmask; // [V] that set (bit = tile)
    // the state-independent part of a matched point's record, one contiguous block per vertex (what k_records gathers: 33 shape-plane
    // values, 4 weights, 4 joints and 16 ancestor words would otherwise be 57 scattered 8-byte reads per matched point)
    double* vrec;         // [V][rec_quad / 4] = record fields of the vertex, the mean-data-point and sqrt(count) fields zero
    int* fk_items;        // [J*(12+3K)][2] per-level work items of k_solve's skeleton pass (see avt_lm.hip), grouped by level
    int* fk_level_off;    // [nlevels+1] offsets into fk_items
    int* fk_titems;       // [AVT_PREP_LEVELS_REG][AVT_PREP_TITEM_THREADS][2] the same items by (level, thread of a 256-thread workgroup), -1 -1 = none: a thread requests its item of
                          // every level at kernel start, one independent load each (AvtDims::fk_reg)
    double* jsr_base;     // [3J] initialJointPos
    double* jsr;          // [3J][K] row-major jointShapeReg
    double* S;            // [J][3][K]
    double* Sp;           // [J][3][K]
    // GMM
    double* prior_mean;   // [C][n]
    double* prior_prec;   // [C][n][n] precision = L L^T
    double* prior_clog;   // [C]
    // part structure (context-level, depends on part_map)
    int* part_of_vertex;  // [V]
    int* part_start;      // [num_parts+1] into part_vertices
    int* part_vertices;   // [V] vertex ids grouped by part, ascending inside a part
    int* part_pos;        // [V] inverse of part_vertices
    // moment form (avt_moments.hip)
    int* mom_pair;        // [2 np] (k, k') of every unordered pair
    int* mom_lstart;      // [np + 1] per-pair vertex lists: the vertices both joints are assigned to, ascending id ...
    int* mom_lv;          // ... vertex ids
    double* mom_lw;       // ... [2] the two skinning weights (a_k, a_k')
    double* mom_psi;      // [V][16 ntp] psi_m, zero padded
    int* mom_opk_start;   // [J + 1] ordered pairs (op = 2 p: k -> k', 2 p + 1: k' -> k) whose lever joint is k ...
    int* mom_opk;         // ... op ids
    int* mom_sub_start;   // [J + 1] subtree of every joint (itself included) ...
    int* mom_sub;         // ... joint ids, ascending
    int* mom_m1_start;    // [nm1 + 1] stage 1 of the rot-rot block: entry (k, j') sums the ordered pairs (k, k'), k' under j' ...
    int* mom_m1;          // ... op ids
    int* mom_s2_start;    // [2 nb2 + 1] stage 2: block (j <= j') sums the stage-1 entries (k, j'), k under j (list 2b) and (k, j), k under j' (list 2b + 1) ...
    int* mom_s2;          // ... stage-1 entry ids
    int* mom_s2_jj;       // [nb2] j | j' << 8
    int* mom_z2_jj;       // [nz2] the blocks that are structural zeros
    unsigned short* mom_tab16;   // the 16-bit index lists of k_assemble as one block (AvtDims::mom_toff)
};

struct FrameBuffers {
    int max_frames, max_points;   // per frame
    int G;                        // eval blocks per frame
    int nspec;                    // speculative solver workgroups per frame in the current k_solve launch (riding shape; 0: none); k_eval: of the launch that follows
    int nspec_cost;               // k_solve: workgroups per frame that reduce the speculative steps' costs (= nspec when k_eval evaluated them, else 0)
    int seq;                      // which solve of the ICP iteration the current k_solve launch is (1 = FIRST): the riding reduction counts up to seq x its workgroups (k_eval: the solve that FOLLOWS it)
    int max_iters;                // GN iterations per ICP iteration of the call being enqueued (the accept tests a launch sequence may take ahead of its launches are bounded by it)
    int f0;                       // first frame of the frame group a launch covers (grid frame index is relative to it)
    int xcd_frames;               // avt_tuning::xcd_frames (xcd_frame_block, avt_device.h)
    // raw inputs
    double* data_raw;     // [max_frames*max_points][3]
    int* labels_raw;      // [max_frames*max_points]
    // part-sorted data (per frame segment at frame*max_points)
    double* dx; double* dy; double* dz;
    int* dorig;           // original index of sorted point
    int* part_off;        // [max_frames][num_parts+1] offsets (relative to frame segment)
    int* part_cnt;        // [max_frames][2][AVT_MAX_PARTS+1] label histogram | (unused since the scatter is stable)
    int* tile_hist;       // [max_frames][bucket_tiles][AVT_MAX_PARTS+1] label histogram of every 2048-point tile (stable scatter, avt_bucket.h)
    int bucket_tiles;     // ceil(max_points / 2048)
    int* corr;            // [max_frames*max_points] model idx per ORIGINAL data index (-1 none)
    int* corr_sorted;     // per sorted position
    // model-side per frame
    double* cloud;        // [max_frames][3V] xyz interleaved (ava.cloud)
    double* pcx; double* pcy; double* pcz;   // [max_frames][V] cloud in part-sorted order; invisible -> +inf
    unsigned char* visible;                  // [max_frames][V]
    double* vcx; double* vcy; double* vcz;   // [max_frames][V] visible model points, compacted per part segment
    int* vcid;                               // [max_frames][V] vertex id of each compacted candidate
    int* vcount;                             // [max_frames][num_parts] visible candidates per part
    unsigned* fault;                         // [max_frames] sticky error bits a kernel sets when it cannot vouch for its result (AVT_FAULT_*); the host reports and clears them
    long long ride_timeout;                  // how long a solver role waits for the riding reduction, in wall_clock64() ticks (100 MHz); AVT_RIDE_TIMEOUT_US
    unsigned* ride_ctr;                      // [max_frames] reduction workgroups that have delivered since k_finalize cleared it (k_solve launch seq waits for seq x its reduction workgroups)
    AvtSolveSnap* snap;                      // [max_frames] (above) written by k_eval, read by the solver roles of the k_solve launch behind it
    AvtSpecCtl* spec;                        // [max_frames] speculative steps of the current system (avt_lm.hip)
    double* x_spec;                          // [max_frames][AVT_MAX_SPEC][xsize] their trial states ...
    double* prep_spec;                       // [max_frames][AVT_MAX_SPEC][prep_size] ... and skeleton tables
    // the COST of every speculative step's trial point, evaluated beside the trial point (spec-cost workgroups of k_eval, riding shapes only):
    double* partial_spec;                    // [spec_frames][AVT_MAX_SPEC][AVT_G_MAX][256] the partial tile that holds sum c |r|^2 ...
    unsigned long long* wmask_spec;          // [spec_frames][AVT_MAX_SPEC][AVT_G_MAX] ... which workgroups wrote it
    double* prior_spec;                      // [spec_frames][AVT_MAX_SPEC][AVT_MAX_COMPS][AVT_PRIOR_STRIDE] GMM scores at those points
    int spec_frames;                         // frames these three hold (the riding shapes: <= AVT_SPEC_FRAMES); the verdict of the accept test taken on them travels in ride_ctr (avt_lm.hip)
    unsigned char* vis_sorted;               // [max_frames][V] the visibility flags in part-sorted order (inside optimize(): k_nn_vis)
    // correspondence aggregation
    int* cnt;             // [max_frames][V]
    long long* fsum;      // [max_frames][3][V] fixed-point centred sums
    int* matched;         // [max_frames][V] compacted matched vertex ids
    double* const_part;   // [max_frames][const_blocks]
    int const_blocks;
    int const_used;       // blocks written by the last k_cost_const launch
    // optimiser state
    double* x;            // [max_frames][2][xsize]
    double* x_start;      // copy of x as avt_state_upload installed it (avt_state_reset)
    AvtFrameCtl* ctl_start;
    double* prep;         // [max_frames][2][prep_size]
    double* rec;          // [max_frames][nb_max][4][rec_quad] matched-point records (k_records)
    int* bmask;           // [max_frames][nb_max] tiles touched by each batch of 16 matched points
    int* erange;          // [max_frames][AVT_ERANGE] frame batches (G < 64): evaluation workgroup g takes batches [erange[g], erange[g+1]) (k_solve INIT)
    double* partial;      // [max_frames][G][NPAIR][256]
    unsigned long long* wmask;   // [max_frames][G] tile pairs workgroup g of the frame wrote to `partial` (bit = pair); k_reduce skips the rest
    double* Hraw;         // [max_frames][2][HS*HS] reduced data-term [J|r]^T W [J|r] (full symmetric) per state slot
    double* prior;        // [max_frames][2][AVT_MAX_COMPS][AVT_PRIOR_STRIDE] GMM scores / Prec*(x-mu) per state slot
    AvtFrameCtl* ctl;     // [max_frames]
    double* jointpos;     // [max_frames][3J]
    double* jointtrans;   // [max_frames][12J]
    double* trace;        // [max_frames][64] cost trace (debug)
    double* results;      // [max_frames][xsize + 8] the result record of every frame - current state (p, q, w), seven statistics, the fault word (k_pack_results' layout) -
                          // written by workgroup 0 of the k_lbs launch that closes optimize(): what the batch split gathers and the host-pointer calls copy back
    const AvtRunParams* params;   // one block per context
    // moment form of the data term (avt_moments.hip): accumulated once per ICP iteration by k_moments
    double* mom_T;        // [max_frames][np][npsi (npsi + 1) / 2] T_kk' = sum_m c_m a_mk a_mk' psi_m psi_m^T, packed upper triangle
    double* mom_D;        // [max_frames][J][npsi][3] D_k = sum_m a_mk psi_m (sum_i (d_i - centre))^T
    double* mom_E;        // [max_frames][2] sum_m |fsum_m|^2 / c_m
    double* mom_rec;      // [max_frames][avt_moments_frame_scratch] what k_pairpass hands to k_assemble (avt_moments.hip)
    int use_moments;      // the GN iterations take their normal equations from the moments (k_assemble) instead of k_eval + k_reduce
};

struct avt_model {
    AvtDims d;
    // host copies (used by avt_ctx_create to build the device model and by accessors)
    std::vector<double> shape_planes, lbs_w, asg_w, jsr_base, jsr, S, Sp, vrec;
    std::vector<int> lbs_j, asg_j, mesh_soa, parent, main_joint, jlevel, fk_items, fk_level_off, fk_titems;
    std::vector<int> tile_col, tile_param, joint_col, vorder, deal_col;
    std::vector<unsigned char> anc_n;
    std::vector<unsigned short> anc, vmask;
    std::vector<double> prior_mean, prior_prec, prior_L, prior_clog;
    // moment form (build_moment_tables)
    std::vector<int> mom_pair, mom_lstart, mom_lv, mom_opk_start, mom_opk, mom_sub_start, mom_sub, mom_m1_start, mom_m1, mom_s2_start, mom_s2, mom_s2_jj, mom_z2_jj;
    std::vector<double> mom_lw, mom_psi;
    std::vector<unsigned short> mom_tab16;
};

struct avt_ctx {
    int device;
    hipStream_t stream;
    hipStream_t side[AVT_MAX_GROUPS - 1];   // branches of the frame-group pipeline (large batches)
    hipEvent_t ev_fork, ev_join[AVT_MAX_GROUPS - 1];
    hipStream_t cur_stream;          // stream the launch wrappers enqueue on
    const avt_model* model;
    DeviceModel dm;
    FrameBuffers fb;
    std::vector<int> part_map;
    int nframes;                     // frames currently resident
    std::vector<int> frame_N, frame_off;
    bool profiling;
    unsigned prof_mask;
    std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> prof_events;
    std::vector<hipEvent_t> event_pool;
    size_t event_pool_used;
    std::vector<void*> allocs;
    int ran_icp_iters, ran_max_iters;
    int launch_maxN;                 // max points per frame of the resident batch, rounded up to 2048 (grid sizing)
    int num_cus;                     // compute units of the device (hipDeviceProp_t::multiProcessorCount)
    bool lbs_cleared;                // the preceding k_lbs reset visibility / correspondence bookkeeping
    bool nn_from_cloud;              // inside optimize(), frame batches: k_compact gathers the candidates from the cloud (no pcx/pcy/pcz)
    bool scatter_in_compact;         // launch_visibility left the scatter pass of the bucketing to the k_compact launch that follows
    int vis_frame_min;               // frames per launch from which visibility runs as one workgroup per frame (0: never; tun.vis_frame_min where the LDS allows)
    struct GraphEntry { std::string key; hipGraphExec_t exec; unsigned long long last_used; };
    std::vector<GraphEntry> graphs;  // small LRU cache keyed on the launch SHAPE only (frames, groups, grids, iteration counts)
    unsigned long long graph_clock;
    AvtRunParams params_host;        // what fb.params currently holds
    bool params_valid;
    bool frames_valid, state_valid;  // resident frames / start state usable by avt_optimize_resident
    int data_term;                   // AVT_DATA_TERM_* policy (avt_set_data_term)
    avt_tuning tun;                  // launch-shape / algorithm knobs (include/avt.h): defaults, then the environment ONCE at creation, then avt_ctx_set_tuning
    bool last_run_moments;           // the form the last optimize() ran
    std::string mom_reason;          // why this context has no moment form (empty: it has one)
    bool have_moments, have_records; // what exists for the resident correspondences (avt_get_normal_equations makes the other on demand)
    int concurrent_groups;           // frame groups the current optimize() call runs side by side (sizes the riding launch shapes)
    // host-to-host calls (avt_optimize / avt_optimize_batch on host pointers): one pinned staging block for everything that crosses PCIe
    // in either direction and one device block for the packed results, grown on demand - the call then needs ONE host synchronisation
    char* host_pin; size_t host_pin_cap;
    bool results_fresh;              // fb.results describes the resident states (the last optimize() packed them and nothing changed them since)
    // persistent scratch of avt_synth_render_frames (z-buffer keys, labels, block counts), grown on demand
    unsigned long long* render_zkey; unsigned char* render_label; int* render_block; size_t render_cap_pix; size_t render_cap_blk;
    // painter's-order mode only: second key image, float depth image, per-face sort key / order position / edge-on flag
    unsigned long long* render_mkey; float* render_depth; float* render_fkey; int* render_frank; unsigned char* render_fedge;
    size_t render_cap_paint_pix; size_t render_cap_paint_face;
    int render_img_frames, render_img_w, render_img_h;   // what render_depth / render_label hold (avt_synth_render_images); 0 = nothing
};

void avt_set_error(const std::string& s);

// kernel launch wrappers (avt_kernels.hip / avt_nn.hip)
enum { SOLVE_INIT = 0, SOLVE_FIRST = 1, SOLVE_NORMAL = 2, SOLVE_DECIDE = 3 /* the accept test of the last trial point alone (moment form) */ };
void launch_lbs(avt_ctx* c, int nframes, const double* x_state_or_null, const double* w, const double* p, const double* R,
                int from_state, int vis_init /* -1: leave bookkeeping alone; 0/1: reset it, visibility flags to this value */,
                bool with_bucket_count = false, bool with_init = false, bool decide = false /* from_state 2: accept test of the last trial point first */,
                bool write_pc = true /* also the part-sorted copy of the cloud (pcx/pcy/pcz, vis_sorted) */, bool pack = false /* workgroup 0 of every frame also writes the frame's result record (fb.results) */);
int avt_lbs_set_attributes();
bool avt_lbs_can_init(const AvtDims& d);
size_t avt_visibility_frame_lds(const AvtDims& d);
bool avt_nn_few(const avt_ctx* c, int nframes);
void launch_visibility(avt_ctx* c, int nframes, int enable, bool with_bucket_scatter = false);
void launch_bucket(avt_ctx* c, int nframes, bool clear_after);
void launch_state_reset(avt_ctx* c, int nframes);
void launch_nn(avt_ctx* c, int nframes);
void launch_finalize(avt_ctx* c, int nframes);
void launch_eval(avt_ctx* c, int nframes, bool cost_only = false, int next_seq = 0 /* which solve of the ICP iteration follows (riding shapes) */);
int avt_solve_nspec(const avt_ctx* c, int nframes);       // speculative solver workgroups per frame of the k_solve launch that follows an evaluation (0: none)
void launch_records(avt_ctx* c, int nframes);
void launch_reduce(avt_ctx* c, int nframes);
bool avt_solve_rides(const avt_ctx* c, int nframes);      // the reduction rides in k_solve's launch: no launch_reduce in front of launch_solve
void launch_solve(avt_ctx* c, int nframes, int mode, int seq = 0 /* which solve of the ICP iteration (riding shape) */);
void launch_pack_results(avt_ctx* c, int nframes, double* out, int stride);
// avt_moments.hip
void launch_moments(avt_ctx* c, int nframes);             // once per ICP iteration, behind k_finalize (carries the cost-constant workgroups)
void launch_assemble(avt_ctx* c, int nframes);            // normal equations of the trial point from the moments (+ the pose-prior workgroups)
int avt_moments_set_attributes();
size_t avt_moments_frame_scratch(const AvtDims& d);   // doubles per frame of FrameBuffers::mom_rec
size_t avt_moments_T_doubles(const AvtDims& d);       // doubles per frame of FrameBuffers::mom_T
long long avt_moments_mfma_count(const avt_model* m, const int* cnt_V);   // matrix instructions of one k_moments pass over a frame with these counts
long long avt_solve_mfma_count(const AvtDims& d);     // ... of one LDL^T factorisation in k_solve
size_t avt_moments_lds_need(const AvtDims& d);        // dynamic LDS the moment form's GN-loop kernels ask for ...
size_t avt_moments_lds_cap();                         // ... and what a workgroup can have
