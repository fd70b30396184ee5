// avt_capi.cpp — the C ABI of include/avt.h: context management, host<->device staging and the launch
// sequence of AvatarOptimizer::optimize() (AvatarOptimizer.cpp:1246-1517).  Compiled with hipcc (host only).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <exception>
#include <mutex>

#include "avt_internal.h"

static std::mutex g_graph_mutex;      // every hipGraph call of the process (see run_optimize)

// No C++ exception may cross the C ABI: every entry point that allocates runs inside this guard.
#define AVT_API_GUARD_BEGIN try {
#define AVT_API_GUARD_END(fn)                                                                      \
    } catch (const std::exception& e) { avt_set_error(std::string(fn) + ": " + e.what()); return 1; } \
    catch (...) { avt_set_error(std::string(fn) + ": unknown exception"); return 1; }

int avt_solve_set_attributes();
int avt_eval_set_attributes();
int avt_render_enqueue(avt_ctx* c, int nframes, const int* d_vertex_part, unsigned long long* d_zkey, unsigned char* d_label, int* d_block,
                       double fx, double fy, double cx, double cy, int width, int height);
int avt_paint_enqueue(avt_ctx* c, int nframes, const int* d_vertex_part, unsigned long long* d_dkey, unsigned long long* d_mkey, float* d_fkey,
                      int* d_frank, unsigned char* d_fedge, float* d_depth, unsigned char* d_label, int* d_block, double fx, double fy, double cx,
                      double cy, int width, int height);
void avt_eval_report_occupancy(const AvtDims& d);

#define HIP_OK(expr)                                                                          \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            avt_set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                 \
            return 1;                                                                         \
        }                                                                                     \
    } while (0)

namespace {

template <class T>
int dev_alloc(avt_ctx* c, T** p, size_t n) {
    void* q = nullptr;
    HIP_OK(hipMalloc(&q, std::max<size_t>(n, 1) * sizeof(T)));
    c->allocs.push_back(q);
    *p = (T*)q;
    return 0;
}
template <class T>
int dev_upload(avt_ctx* c, T** p, const std::vector<T>& v) {
    if (dev_alloc(c, p, v.size())) return 1;
    // (never the legacy stream: another thread of the host may be capturing a graph, and the legacy stream synchronises with it)
    if (!v.empty()) {
        HIP_OK(hipMemcpyAsync(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, c->stream));
        HIP_OK(hipStreamSynchronize(c->stream));
    }
    return 0;
}

// frame groups of one optimize(): measured on MI355X, two groups pay off from ~32 frames, more never do (avt_tuning::groups overrides)
int choose_groups(int nframes, const avt_tuning& tun) {
    // two groups from 44 frames on - measured, round 5 (moment form from 8 frames per launch on), ms per step one / two groups: 16 frames 0.737 / 0.798, 24: 0.777 / 0.822,
    // 32: 0.810 / 0.838, 36: 0.865 / 0.878, 40: 0.880 / 0.892, 44: 0.938 / 0.913, 48: 0.952 / 0.928, 64: 1.077 / 1.011 (round 4, rows below 64 frames: the crossover was 32)
    int n = nframes >= 44 ? 2 : 1;
    if (tun.groups > 0) n = tun.groups;
    return std::max(1, std::min(std::min(n, AVT_MAX_GROUPS), nframes));
}

// Frame groups of one optimize() call.  Two or three frames: one frame per group when the one-frame launch shape has the speculative
// solver workgroups - in a shared launch a rejection saves its factorisation only if every frame of the launch
// rejects, on a stream of its own every frame keeps its own pace (2 frames 0.477 -> 0.457 ms, 3 frames 0.607 -> 0.526; four: 0.637 -> 0.709).
int choose_G(int nframes, int ngroups, const avt_tuning& tun);
int plan_groups(avt_ctx* c, int nf) {
    int ngroups = choose_groups(nf, c->tun);
    if ((nf == 2 || nf == 3) && c->tun.groups <= 0) {
        const int g_keep = c->fb.G;
        c->fb.G = choose_G(1, 1, c->tun);
        if (avt_solve_rides(c, 1)) ngroups = nf;
        c->fb.G = g_keep;
    }
    return ngroups;
}

int choose_G(int nframes, int ngroups, const avt_tuning& tun) {
    // k_eval workgroups per frame.  Up to 64 frames per launch: 768 workgroups = one resident round at 3 per CU (more rounds
    // cost more in partial tiles and prologues than they balance: 128 frames per group 645 k against 609 k GN it/s with twice as
    // many).  From 128 frames per launch on: 1536, two rounds, so that the hardware's dispatch evens out frames with different
    // numbers of matched points (512 frames: 747 k -> 774 k GN it/s; three rounds: 770 k).
    // Two frame groups side by side (from 32 frames on): the evaluation of one group shares the chip with the other group's
    // solves and reductions, and FEWER, longer-lived workgroups per frame win (round 3, tools/g_sweep.sh, ms per step at
    // 512 / 640 / 768 workgroups per launch: 32 frames 0.934 / 0.928 / 0.966; 48: 1.073 / 1.091 / 1.120; 64: 1.227 / 1.246 / 1.271;
    // 96: 1.625 / 1.623 / 1.622; 128: 2.080 / 1.954 / 1.982; 256 frames: 3.73 / 3.64 / 3.50, 1536: 3.50).
    const int target = nframes >= 128 ? 1536 : (ngroups >= 2 ? (nframes <= 40 ? 512 : 640) : 768);
    const int gcap = std::max(2, std::min(AVT_G_MAX, tun.gcap));
    int g = std::max(2, std::min(gcap, target / std::max(1, nframes)));
    // G >= 64 also selects the few-frames launch shapes (strided batches, k_reduce_strip, the trial point set up in k_lbs's
    // grid): they win up to 6 frames (8 frames: 0.762 ms against 0.712 with G = 63; 12: 0.822 / 0.770)
    if (nframes >= 7) g = std::min(g, 63);
    if (tun.g > 0) return std::max(1, std::min(g, tun.g));   // tuning knob, never above the allocation
    return g;
}

// the defaults of avt_tuning, then - ONCE, here - what the environment says (AVT_<FIELD>); unknown AVT_* names are reported
extern "C" char** environ;
avt_tuning tuning_from_environment() {
    avt_tuning t;
    std::memset(&t, 0, sizeof t);
    t.use_graph = 1; t.groups = 0; t.g = 0; t.gcap = 128; t.vis_frame_min = 32; t.ride = 1; t.ride_strips = 0; t.ride_sizing_groups = 0;
    t.asm_parts = 1; t.spec_cost = 1; t.xcd_frames = 1; t.literal_dims = 1; t.nspec = AVT_MAX_SPEC; t.nn_force_part = 0; t.nn_slab = 1; t.mom_min_frames = 8; t.debug = 0; t.ride_timeout_us = 2000000;
    struct Knob { const char* name; int* field; };
    const Knob knobs[] = {{"AVT_USE_GRAPH", &t.use_graph}, {"AVT_GROUPS", &t.groups}, {"AVT_G", &t.g}, {"AVT_GCAP", &t.gcap}, {"AVT_VIS_FRAME_MIN", &t.vis_frame_min},
                          {"AVT_RIDE", &t.ride}, {"AVT_RIDE_STRIPS", &t.ride_strips}, {"AVT_RIDE_SIZING_GROUPS", &t.ride_sizing_groups}, {"AVT_NSPEC", &t.nspec},
                          {"AVT_NN_FORCE_PART", &t.nn_force_part}, {"AVT_NN_SLAB", &t.nn_slab}, {"AVT_MOM_MIN_FRAMES", &t.mom_min_frames}, {"AVT_ASM_PARTS", &t.asm_parts}, {"AVT_LBS_FRAMES", &t.lbs_frames}, {"AVT_SPEC_COST", &t.spec_cost}, {"AVT_XCD_FRAMES", &t.xcd_frames}, {"AVT_LITERAL_DIMS", &t.literal_dims}, {"AVT_DEBUG", &t.debug}};
    // names other parts of the repository own (the batch split, the Python loader, bench.py, instrumented builds)
    const char* others[] = {"AVT_LIB", "AVT_RCCL_LIB", "AVT_SHARD_LOOPBACK_TIMEOUT_S", "AVT_BENCH_SHARE_GPU0", "AVT_TIMING"};
    for (char** e = environ; e && *e; ++e) {
        if (std::strncmp(*e, "AVT_", 4) != 0) continue;
        const char* eq = std::strchr(*e, '=');
        if (!eq) continue;
        const std::string name(*e, eq - *e);
        const char* val = eq + 1;
        bool known = false;
        for (const Knob& k : knobs) if (name == k.name) { *k.field = atoi(val); known = true; }
        if (name == "AVT_RIDE_TIMEOUT_US") { t.ride_timeout_us = std::max(0ll, atoll(val)); known = true; }
        // the spellings of earlier rounds
        if (name == "AVT_NO_GRAPH") { t.use_graph = 0; known = true; }
        if (name == "AVT_ONE_GROUP") { t.groups = 1; known = true; }
        if (name == "AVT_NO_RIDE") { t.ride = 0; known = true; }
        if (name == "AVT_NN_NO_SLAB") { t.nn_slab = 0; known = true; }
        if (name == "AVT_RIDE_SIZING") { t.ride_sizing_groups = std::strcmp(val, "groups") == 0; known = true; }
        for (const char* o : others) if (name == o) known = true;
        if (!known) fprintf(stderr, "libavatar_hip: environment variable %s is not a knob of this library (include/avt.h, avt_tuning) - ignored\n", name.c_str());
    }
    return t;
}

int validate_tuning(const avt_tuning& t) {
    if (t.groups < 0 || t.groups > AVT_MAX_GROUPS || t.g < 0 || t.gcap < 2 || t.vis_frame_min < 0 || t.nspec < 0 || t.nspec > AVT_MAX_SPEC ||
        (t.ride_strips != 0 && t.ride_strips != 4 && t.ride_strips != 8) || t.mom_min_frames < 1 || t.ride_timeout_us < 0 || (t.lbs_frames != 0 && t.lbs_frames != 1 && t.lbs_frames != 2 && t.lbs_frames != 4) || t.spec_cost < 0 || t.spec_cost > 1 || t.xcd_frames < 0 || t.xcd_frames > 1 || t.literal_dims < 0 || t.literal_dims > 1) {
        avt_set_error("avt_tuning: a field is out of range (include/avt.h)");
        return 1;
    }
    return 0;
}

hipEvent_t next_event(avt_ctx* c) {
    if (c->event_pool_used == c->event_pool.size()) {
        hipEvent_t e;
        (void)hipEventCreate(&e);
        c->event_pool.push_back(e);
    }
    return c->event_pool[c->event_pool_used++];
}

struct ProfScope {
    avt_ctx* c;
    int cls;
    hipEvent_t a, b;
    bool on;
    ProfScope(avt_ctx* c_, int cls_) : c(c_), cls(cls_) {
        on = c->profiling && ((c->prof_mask >> cls_) & 1u);
        if (on) { a = next_event(c); b = next_event(c); (void)hipEventRecord(a, c->stream); }
    }
    ~ProfScope() {
        if (on) { (void)hipEventRecord(b, c->stream); c->prof_events.push_back({cls, {a, b}}); }
    }
};

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { avt_set_error(std::string(what) + ": " + hipGetErrorString(e)); return 1; }
    return 0;
}

// the launch sequence of one optimize() over the resident frames
void enqueue_optimize(avt_ctx* c, const avt_options* o, int f0, int nf, hipStream_t stream) {
    c->fb.f0 = f0;
    c->fb.max_iters = o->max_iters_per_icp;
    c->cur_stream = stream;
    c->ran_icp_iters = 0;
    const int vis_init = o->enable_occlusion ? 0 : 1;
    c->lbs_cleared = true;
    // The two passes of the data bucketing ride in the trailing workgroups of the two launches they do not depend on (the label
    // histogram beside the skinning, the scatter beside the visibility pass of the first ICP iteration): two launches fewer on
    // the dependency chain; k_finalize restores the cursors.  (No ICP iteration: the stand-alone launches, cursors cleared.)
    const bool fuse_bucket = o->icp_iters > 0;
    // Few frames (the latency shape, G >= 64): the trial point of every ICP iteration (trial := current + its skeleton tables)
    // is set up by a workgroup in the grid of the k_lbs launch in front of it instead of by a k_solve INIT launch behind k_records.
    // (round 6: small frame batches on the moment form too - up to 16 frames per launch: 0.737 -> 0.726 ms; the launch then asks for the
    // workgroup's 90 KB of LDS for EVERY workgroup of the skinning, one per CU, and at 32 frames per launch that costs more than the INIT launch
    // saves: 0.980 -> 0.990 ms.  The row form's batches need k_solve INIT's deal of the evaluation ranges.)
    const bool fuse_init = (c->fb.G >= 64 || (c->fb.use_moments && nf <= 16)) && avt_lbs_can_init(c->dm.d);
    if (!fuse_bucket) { ProfScope ps(c, AVT_K_BUCKET); launch_bucket(c, nf, true); }
    // frame batches: k_compact gathers its candidates from the cloud, so k_lbs does not write the part-sorted copy of it
    const bool few = avt_nn_few(c, nf);
    c->nn_from_cloud = !few;
    { ProfScope ps(c, AVT_K_LBS); launch_lbs(c, nf, nullptr, nullptr, nullptr, nullptr, 1, vis_init, fuse_bucket, fuse_init && o->icp_iters > 0, false, few); }   // ava.update() precondition (:1356)
    for (int icp = 0; icp < o->icp_iters; ++icp) {
        { ProfScope ps(c, AVT_K_VISIBILITY); launch_visibility(c, nf, o->enable_occlusion, fuse_bucket && icp == 0); }
        { ProfScope ps(c, AVT_K_NN); launch_nn(c, nf); }
        if (c->fb.use_moments) {
            // Moment form (avt_moments.hip): the correspondences' sufficient statistics once per ICP iteration, then every GN iteration
            // assembles its normal equations from them - no Jacobian rows, no partial tiles, no reduction.
            { ProfScope ps(c, AVT_K_AGGREGATE); launch_finalize(c, nf); }
            { ProfScope ps(c, AVT_K_MOMENTS); c->fb.const_used = (c->launch_maxN + 2047) / 2048; launch_moments(c, nf); }
            if (!fuse_init) { ProfScope ps(c, AVT_K_PREPARE); launch_solve(c, nf, SOLVE_INIT); }
            { ProfScope ps(c, AVT_K_EVAL); launch_assemble(c, nf); }
            for (int it = 1; it <= std::max(1, o->max_iters_per_icp); ++it) {
                { ProfScope ps(c, AVT_K_SOLVE); launch_solve(c, nf, it == 1 ? SOLVE_FIRST : SOLVE_NORMAL, it); }
                if (o->max_iters_per_icp == 0) break;
                { ProfScope ps(c, AVT_K_EVAL); launch_assemble(c, nf); }
            }
            // the accept test of the last trial point: the decision part of the solve kernel alone (SOLVE_DECIDE: no factorisation, no step)
            if (o->max_iters_per_icp > 0) { ProfScope ps(c, AVT_K_DECIDE); launch_solve(c, nf, SOLVE_DECIDE, o->max_iters_per_icp + 1); }
            { ProfScope ps(c, AVT_K_LBS); const bool more = icp + 1 < o->icp_iters;
              launch_lbs(c, nf, nullptr, nullptr, nullptr, nullptr, 2, more ? vis_init : -1, false, fuse_init && more, false, few && more, !more); }      // (the last launch of the call also writes the result records)
            c->ran_icp_iters++;
            continue;
        }
        { ProfScope ps(c, AVT_K_AGGREGATE); launch_finalize(c, nf); launch_records(c, nf); }
        if (!fuse_init) { ProfScope ps(c, AVT_K_PREPARE); launch_solve(c, nf, SOLVE_INIT); }
        const bool rides = avt_solve_rides(c, nf);        // few frames: the reduction is part of the solve's launch
        { ProfScope ps(c, AVT_K_EVAL); launch_eval(c, nf, false, 1); }
        if (!rides) { ProfScope ps(c, AVT_K_REDUCE); launch_reduce(c, nf); }
        for (int it = 1; it <= std::max(1, o->max_iters_per_icp); ++it) {
            { ProfScope ps(c, AVT_K_SOLVE); launch_solve(c, nf, it == 1 ? SOLVE_FIRST : SOLVE_NORMAL, it); }
            if (o->max_iters_per_icp == 0) break;
            if (it < o->max_iters_per_icp) {
                { ProfScope ps(c, AVT_K_EVAL); launch_eval(c, nf, false, it + 1); }
                if (!rides) { ProfScope ps(c, AVT_K_REDUCE); launch_reduce(c, nf); }
            } else {   // no solve follows the last trial point: its cost alone; the accept test is taken by the k_lbs launch below
                ProfScope ps(c, AVT_K_DECIDE);
                launch_eval(c, nf, true);
            }
        }
        { ProfScope ps(c, AVT_K_LBS); const bool more = icp + 1 < o->icp_iters;      // the last launch of the call skins only: no bookkeeping reset, no part-sorted copy
          launch_lbs(c, nf, nullptr, nullptr, nullptr, nullptr, 2, more ? vis_init : -1, false, fuse_init && more, o->max_iters_per_icp > 0, few && more, !more); }   // :1494-1497 (2: from the skeleton tables of the current point)
        c->ran_icp_iters++;
    }
    c->lbs_cleared = false;
    c->nn_from_cloud = false;
    c->fb.f0 = 0;
    c->cur_stream = c->stream;
}

// the option scalars the kernels read live in device memory (AvtRunParams): uploaded only when they change
int sync_params(avt_ctx* c, const avt_options* o) {
    AvtRunParams pr;
    std::memset(&pr, 0, sizeof pr);
    pr.beta_pose = o->beta_pose; pr.beta_shape = o->beta_shape; pr.lambda0 = o->lm_lambda0;
    pr.lm_up = o->lm_up; pr.lm_down = o->lm_down; pr.lm_min = o->lm_lambda_min; pr.lm_max = o->lm_lambda_max; pr.lm_policy = o->lm_policy == 1 ? 1.0 : 0.0;
    pr.ftol = o->function_tolerance;
    if (c->params_valid && std::memcmp(&pr, &c->params_host, sizeof pr) == 0) return 0;
    c->params_host = pr;
    HIP_OK(hipMemcpyAsync((void*)c->fb.params, &c->params_host, sizeof pr, hipMemcpyHostToDevice, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));     // params_host may be rewritten by the next call
    c->params_valid = true;
    return 0;
}

constexpr size_t GRAPH_CACHE_ENTRIES = 8;

// which form of the data term an optimize() over nf resident frames runs (include/avt.h, avt_set_data_term): ONE rule for run_optimize,
// avt_launch_shape and avt_get_normal_equations.  AUTO: the moment form from tun.mom_min_frames frames per launch on.
bool choose_moments(const avt_ctx* c, int nf) {
    const int g0 = choose_groups(nf, c->tun), per_launch = (nf + g0 - 1) / g0;
    return c->dm.d.mom_ok && (c->data_term == AVT_DATA_TERM_MOMENTS || (c->data_term == AVT_DATA_TERM_AUTO && per_launch >= c->tun.mom_min_frames));
}

// what exists for the resident correspondences after an optimize() call has been enqueued - by launches, by a fresh capture or by the
// replay of a cached graph alike (set inside enqueue_optimize these flags described the last CAPTURED graph, not the last one run)
void note_products(avt_ctx* c, const avt_options* o) {
    if (o->icp_iters > 0) { c->have_moments = c->fb.use_moments != 0; c->have_records = !c->have_moments; }
    c->results_fresh = o->icp_iters > 0;      // the closing k_lbs launch of the call wrote fb.results (no ICP iteration: no closing launch)
}

int run_optimize(avt_ctx* c, const avt_options* o) {
    const int nf = c->nframes;
    if (nf <= 0 || !c->frames_valid) { avt_set_error("avt_optimize: no frames resident (upload frames first; the stand-alone entry points avt_nn / avt_visibility / avt_lbs_update invalidate them)"); return 1; }
    if (!c->state_valid) { avt_set_error("avt_optimize: no start state resident for these frames (avt_state_upload)"); return 1; }
    if (o->max_iters_per_icp < 0 || o->max_iters_per_icp > 62 || o->icp_iters < 0) { avt_set_error("avt_optimize: bad iteration counts"); return 1; }
    if (o->lm_policy != 0 && o->lm_policy != 1) { avt_set_error("avt_optimize: lm_policy must be 0 (fixed factors) or 1 (gain ratio)"); return 1; }
    if (!(o->lm_up > 1.0) || !(o->lm_down > 0.0 && o->lm_down < 1.0)) { avt_set_error("avt_optimize: lm_up must be > 1 and lm_down in (0, 1)"); return 1; }
    if (!(o->function_tolerance >= 0.0 && o->function_tolerance < 1.0)) { avt_set_error("avt_optimize: function_tolerance must be in [0, 1) (0 = no early exit)"); return 1; }
    if (sync_params(c, o)) return 1;
    c->ran_max_iters = o->max_iters_per_icp;
    // Large batches run as several frame groups: the latency-bound single-workgroup-per-frame kernels of one group
    // (k_solve, k_finalize, k_reduce) overlap the throughput kernels (k_eval, k_nn) of the others on separate streams.
    // The instrumented (profiling) path keeps the SAME groups and launch shapes and runs them one after the other on
    // the main stream, so that per-launch timings describe the launches the graph replays.
    c->fb.use_moments = choose_moments(c, nf);      // which form of the data term this call runs
    c->last_run_moments = c->fb.use_moments != 0;
    const int ngroups = plan_groups(c, nf);
    const int nfg = (nf + ngroups - 1) / ngroups;       // frames per group (the last group may be smaller)
    c->fb.G = choose_G(nfg, ngroups, c->tun);
    c->concurrent_groups = ngroups;
    if (!c->tun.use_graph || c->profiling) {
        for (int gi = 0; gi < ngroups; ++gi) {
            const int f0 = gi * nfg, n = std::min(nfg, nf - f0);
            if (n > 0) enqueue_optimize(c, o, f0, n, c->stream);
        }
        c->ran_icp_iters = o->icp_iters;
        note_products(c, o);
        return check_launch("optimize launch sequence");
    }
    // The launch sequence depends only on the launch shape: capture it once, replay it afterwards.
    // Every graph call of the process goes through one mutex: contexts are per thread, but HIP 7.0's hipGraphLaunch is not safe
    // against a hipGraphLaunch from another thread (found by the 8-thread loop-back tests: SIGSEGV in hip::Graph::UpdateStreams <-
    // hip::GraphExec::Run, which rearranges a per-device set of helper streams for graphs with parallel branches).  The calls
    // only enqueue, so the lock is held for microseconds.
    std::lock_guard<std::mutex> graph_lock(g_graph_mutex);
    char key[160];
    snprintf(key, sizeof key, "%d|%d|%d|%d|%d|%d|%d|%d", nf, ngroups, c->fb.G, c->launch_maxN, o->icp_iters, o->max_iters_per_icp, o->enable_occlusion, c->fb.use_moments);
    avt_ctx::GraphEntry* hit = nullptr;
    for (auto& e : c->graphs) if (e.key == key) { hit = &e; break; }
    if (!hit) {
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        HIP_OK(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
        bool bad = false;
        auto cap = [&](hipError_t e) { if (e != hipSuccess && !bad) { bad = true; avt_set_error(std::string("optimize graph capture: ") + hipGetErrorString(e)); } };
        if (ngroups > 1) {
            cap(hipEventRecord(c->ev_fork, c->stream));
            for (int gi = 1; gi < ngroups; ++gi) cap(hipStreamWaitEvent(c->side[gi - 1], c->ev_fork, 0));
            for (int gi = 0; gi < ngroups; ++gi) {
                const int f0 = gi * nfg, n = std::min(nfg, nf - f0);
                if (n > 0) enqueue_optimize(c, o, f0, n, gi == 0 ? c->stream : c->side[gi - 1]);
            }
            for (int gi = 1; gi < ngroups; ++gi) {
                cap(hipEventRecord(c->ev_join[gi - 1], c->side[gi - 1]));
                cap(hipStreamWaitEvent(c->stream, c->ev_join[gi - 1], 0));
            }
        } else {
            enqueue_optimize(c, o, 0, nf, c->stream);
        }
        cap(hipGetLastError());
        // the capture is ALWAYS ended, so that a failure does not leave the stream stuck in capture mode
        const hipError_t ee = hipStreamEndCapture(c->stream, &graph);
        if (bad || ee != hipSuccess) {
            if (!bad) avt_set_error(std::string("hipStreamEndCapture: ") + hipGetErrorString(ee));
            if (graph) (void)hipGraphDestroy(graph);
            return 1;
        }
        const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (ei != hipSuccess) { avt_set_error(std::string("hipGraphInstantiate: ") + hipGetErrorString(ei)); return 1; }
        if (c->graphs.size() >= GRAPH_CACHE_ENTRIES) {      // evict the least recently used launch shape
            size_t lru = 0;
            for (size_t i = 1; i < c->graphs.size(); ++i) if (c->graphs[i].last_used < c->graphs[lru].last_used) lru = i;
            HIP_OK(hipStreamSynchronize(c->stream));
            (void)hipGraphExecDestroy(c->graphs[lru].exec);
            c->graphs.erase(c->graphs.begin() + lru);
        }
        c->graphs.push_back({key, exec, 0});
        hit = &c->graphs.back();
    }
    hit->last_used = ++c->graph_clock;
    HIP_OK(hipGraphLaunch(hit->exec, c->stream));
    c->ran_icp_iters = o->icp_iters;
    note_products(c, o);
    return 0;
}

// Installs `nframes` frames with `counts[f]` points each; data / labels are packed back to back (frame f starts at the sum
// of the previous counts) in host memory or, with device_src, in device memory (the batch split's receive buffer).
// Everything is validated before any host or device state changes.  The per-frame point count is also written into the
// device control blocks (working copy and start copy), so that swapping frames under a resident state - the warm-start
// pattern avt_frames_upload + avt_optimize_resident - runs with the new counts.
int install_frames(avt_ctx* c, int nframes, const int* counts, const double* data, const int* labels, bool device_src) {
    if (nframes <= 0 || nframes > c->fb.max_frames) { avt_set_error("frames: nframes out of range for this context"); return 1; }
    int mx = 0;
    for (int f = 0; f < nframes; ++f) {
        if (counts[f] < 0) { avt_set_error("frames: negative point count (frame_offsets must be non-decreasing)"); return 1; }
        if (counts[f] > c->fb.max_points) { avt_set_error("frames: a frame has more points than max_points_per_frame"); return 1; }
        mx = std::max(mx, counts[f]);
    }
    const bool same_shape = c->frames_valid && nframes == c->nframes;
    c->frames_valid = false;
    c->have_moments = c->have_records = false;      // new frames: the moments / records of the old correspondences describe nothing resident
    c->results_fresh = false;
    c->nframes = nframes;
    c->frame_N.assign(counts, counts + nframes);
    c->frame_off.assign(nframes + 1, 0);
    for (int f = 0; f < nframes; ++f) c->frame_off[f + 1] = c->frame_off[f] + counts[f];
    c->launch_maxN = std::max(mx, std::min(c->fb.max_points, ((mx + 2047) / 2048) * 2048));
    const hipMemcpyKind kind = device_src ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    for (int f = 0; f < nframes; ++f) {
        const size_t N = (size_t)counts[f], o = (size_t)c->frame_off[f];
        if (N == 0) continue;
        HIP_OK(hipMemcpyAsync(c->fb.data_raw + (size_t)f * c->fb.max_points * 3, data + o * 3, N * 3 * sizeof(double), kind, c->stream));
        HIP_OK(hipMemcpyAsync(c->fb.labels_raw + (size_t)f * c->fb.max_points, labels + o, N * sizeof(int), kind, c->stream));
    }
    for (AvtFrameCtl* dst : {c->fb.ctl, c->fb.ctl_start})
        HIP_OK(hipMemcpy2DAsync(&dst->N, sizeof(AvtFrameCtl), c->frame_N.data(), sizeof(int), sizeof(int), (size_t)nframes, hipMemcpyHostToDevice, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    c->frames_valid = true;
    if (!same_shape) c->state_valid = false;     // a different number of frames needs a new start state
    return 0;
}

int upload_frames(avt_ctx* c, int nframes, const double* data, const int* labels, const int* offs) {
    if (nframes <= 0 || nframes > c->fb.max_frames) { avt_set_error("frames: nframes out of range for this context"); return 1; }
    std::vector<int> counts(nframes);
    for (int f = 0; f < nframes; ++f) counts[f] = offs[f + 1] - offs[f];
    return install_frames(c, nframes, counts.data(), data + (size_t)offs[0] * 3, labels + offs[0], false);
}

int upload_state(avt_ctx* c, int nframes, const double* p, const double* q, const double* w) {
    const AvtDims& d = c->dm.d;
    if (!c->frames_valid || nframes != c->nframes) { avt_set_error("state: nframes differs from the resident frames"); return 1; }
    std::vector<double> xs((size_t)nframes * 2 * d.xsize, 0.0);
    std::vector<AvtFrameCtl> ctl(nframes);
    for (int f = 0; f < nframes; ++f) {
        double* x = &xs[(size_t)f * 2 * d.xsize];
        std::copy(p + 3 * f, p + 3 * f + 3, x);
        std::copy(q + (size_t)4 * d.J * f, q + (size_t)4 * d.J * (f + 1), x + 3);
        std::copy(w + (size_t)d.K * f, w + (size_t)d.K * (f + 1), x + 3 + 4 * d.J);
        std::memset(&ctl[f], 0, sizeof(AvtFrameCtl));
        ctl[f].N = c->frame_N[f];
        ctl[f].comp_cur = -1;
    }
    HIP_OK(hipMemcpyAsync(c->fb.x, xs.data(), xs.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_OK(hipMemcpyAsync(c->fb.ctl, ctl.data(), ctl.size() * sizeof(AvtFrameCtl), hipMemcpyHostToDevice, c->stream));
    HIP_OK(hipMemcpyAsync(c->fb.x_start, c->fb.x, xs.size() * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
    HIP_OK(hipMemcpyAsync(c->fb.ctl_start, c->fb.ctl, ctl.size() * sizeof(AvtFrameCtl), hipMemcpyDeviceToDevice, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));  // host vectors go out of scope
    c->state_valid = true;
    c->results_fresh = false;
    return 0;
}

int download_state(avt_ctx* c, double* p, double* q, double* w, avt_stats* st) {
    const AvtDims& d = c->dm.d;
    const int nf = c->nframes;
    std::vector<double> xs((size_t)nf * 2 * d.xsize);
    std::vector<AvtFrameCtl> ctl(nf);
    HIP_OK(hipMemcpyAsync(xs.data(), c->fb.x, xs.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipMemcpyAsync(ctl.data(), c->fb.ctl, ctl.size() * sizeof(AvtFrameCtl), hipMemcpyDeviceToHost, c->stream));
    std::vector<unsigned> fault(nf, 0u);
    HIP_OK(hipMemcpyAsync(fault.data(), c->fb.fault, fault.size() * sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    int bad = -1, nbad = 0;
    for (int f = 0; f < nf; ++f) if (fault[f]) { if (bad < 0) bad = f; ++nbad; }
    if (bad >= 0) {     // reported once, then cleared: the next optimize() starts clean
        char msg[256];
        snprintf(msg, sizeof msg, "optimize: %d frame(s) carry a device fault (first: frame %d, bits 0x%x%s); their result is not valid", nbad, bad,
                 fault[bad], (fault[bad] & AVT_FAULT_RIDE_TIMEOUT) ? ": a solver gave up waiting for the in-launch reduction" : "");
        HIP_OK(hipMemsetAsync(c->fb.fault, 0, (size_t)c->fb.max_frames * sizeof(unsigned), c->stream));
        HIP_OK(hipStreamSynchronize(c->stream));
        c->results_fresh = false;      //  the result records carry the word that was just cleared: a later gather packs them again
        avt_set_error(msg);
        return AVT_STATUS_DEVICE_FAULT;
    }
    for (int f = 0; f < nf; ++f) {
        const double* x = &xs[((size_t)f * 2 + ctl[f].cur_slot) * d.xsize];
        if (p) std::copy(x, x + 3, p + 3 * f);
        if (q) std::copy(x + 3, x + 3 + 4 * d.J, q + (size_t)4 * d.J * f);
        if (w) std::copy(x + 3 + 4 * d.J, x + d.xsize, w + (size_t)d.K * f);
        if (st) {
            st[f].initial_cost = ctl[f].cost_initial;
            st[f].final_cost = ctl[f].cost_cur;
            st[f].lambda = ctl[f].lambda;
            st[f].num_correspondences = ctl[f].T;
            st[f].matched_model_points = ctl[f].M;
            st[f].gn_iterations = ctl[f].gn_iterations;
            st[f].accepted_steps = ctl[f].accepted;
        }
    }
    return 0;
}

// avt_optimize / avt_optimize_batch on HOST pointers - the reference's call shape, optimize(const CloudType&, const VectorXi&, ..)
// (AvatarOptimizer.h:17-19) - as one pass with ONE host synchronisation: clouds, labels, start states and control blocks are gathered in
// the context's pinned staging block and leave as asynchronous copies (no pageable-memory staging inside the runtime, no wait between
// them), the fit is enqueued behind them, k_pack_results puts (p, q, w), the statistics and the fault word of every frame into one block
// and ONE copy brings it back.  (The four-stage path - upload, upload, run, download, a synchronisation each - measured 81 + 26 + 58 us
// on top of the 0.41 ms fit of one 38 k-point frame, tools/host_call_breakdown.py.)  Leaves the context exactly as the staged calls do:
// frames and start state resident, avt_state_reset / avt_optimize_resident usable afterwards.
// (posed_*: optional - frame 0's ava.cloud / jointPos / jointTrans of the closing update(), brought back by the same synchronisation: avt_optimize_posed)
int optimize_host_to_host(avt_ctx* c, int nframes, const double* data, const int* labels, const int* offs, const avt_options* o,
                          double* p, double* q, double* w, avt_stats* st, double* posed_cloud = nullptr, double* posed_jpos = nullptr, double* posed_jtrans = nullptr) {
    const AvtDims& d = c->dm.d;
    if (nframes <= 0 || nframes > c->fb.max_frames) { avt_set_error("frames: nframes out of range for this context"); return 1; }
    long long total = 0;
    int mx = 0;
    for (int f = 0; f < nframes; ++f) {
        const int n = offs[f + 1] - offs[f];
        if (n < 0) { avt_set_error("frames: negative point count (frame_offsets must be non-decreasing)"); return 1; }
        if (n > c->fb.max_points) { avt_set_error("frames: a frame has more points than max_points_per_frame"); return 1; }
        mx = std::max(mx, n); total += n;
    }
    const int xs = d.xsize, stride = xs + 8;
    const size_t b_data = (size_t)total * 24, b_lab = ((size_t)total * 4 + 15) & ~(size_t)15, b_x = (size_t)nframes * 2 * xs * 8,
                 b_ctl = (size_t)nframes * sizeof(AvtFrameCtl), b_res = (size_t)nframes * stride * 8;
    const bool want_posed = posed_cloud || posed_jpos || posed_jtrans;
    const size_t b_posed = want_posed ? (size_t)(3 * d.V + 15 * d.J) * 8 : 0;
    const size_t need = b_data + b_lab + b_x + b_ctl + b_res + b_posed + 64;
    //  the pinned block grows with the largest call and shrinks again when a call needs less than a quarter of it: a tracker that once
    // fitted a dense 512-frame batch does not keep gigabytes pinned for its one-frame calls
    if (c->host_pin_cap < need || (c->host_pin_cap > (64u << 20) && need < c->host_pin_cap / 4)) {
        HIP_OK(hipStreamSynchronize(c->stream));
        if (c->host_pin) (void)hipHostFree(c->host_pin);
        c->host_pin = nullptr; c->host_pin_cap = 0;
        const size_t cap = need + need / 4;
        HIP_OK(hipHostMalloc((void**)&c->host_pin, cap, hipHostMallocDefault));
        c->host_pin_cap = cap;
    }
    char* pin = c->host_pin;
    double* h_data = (double*)pin; int* h_lab = (int*)(pin + b_data); double* h_x = (double*)(pin + b_data + b_lab);
    AvtFrameCtl* h_ctl = (AvtFrameCtl*)(pin + b_data + b_lab + b_x); double* h_res = (double*)(pin + b_data + b_lab + b_x + b_ctl);
    // the context's bookkeeping of the resident frames (install_frames)
    c->frames_valid = c->state_valid = false;
    c->have_moments = c->have_records = false;
    c->nframes = nframes;
    c->frame_N.resize(nframes); c->frame_off.assign(nframes + 1, 0);
    for (int f = 0; f < nframes; ++f) { c->frame_N[f] = offs[f + 1] - offs[f]; c->frame_off[f + 1] = c->frame_off[f] + c->frame_N[f]; }
    c->launch_maxN = std::max(mx, std::min(c->fb.max_points, ((mx + 2047) / 2048) * 2048));
    std::memcpy(h_data, data + (size_t)offs[0] * 3, b_data);
    std::memcpy(h_lab, labels + offs[0], (size_t)total * 4);
    std::memset(h_x, 0, b_x);
    std::memset((void*)h_ctl, 0, b_ctl);
    for (int f = 0; f < nframes; ++f) {
        double* x = h_x + (size_t)f * 2 * xs;
        std::copy(p + 3 * (size_t)f, p + 3 * (size_t)f + 3, x);
        std::copy(q + (size_t)4 * d.J * f, q + (size_t)4 * d.J * (f + 1), x + 3);
        std::copy(w + (size_t)d.K * f, w + (size_t)d.K * (f + 1), x + 3 + 4 * d.J);
        h_ctl[f].N = c->frame_N[f];
        h_ctl[f].comp_cur = -1;
    }
    for (int f = 0; f < nframes; ++f) {
        const size_t N = (size_t)c->frame_N[f], off = (size_t)c->frame_off[f];
        if (N == 0) continue;
        HIP_OK(hipMemcpyAsync(c->fb.data_raw + (size_t)f * c->fb.max_points * 3, h_data + off * 3, N * 24, hipMemcpyHostToDevice, c->stream));
        HIP_OK(hipMemcpyAsync(c->fb.labels_raw + (size_t)f * c->fb.max_points, h_lab + off, N * 4, hipMemcpyHostToDevice, c->stream));
    }
    if (nframes == c->fb.max_frames) {      // [x | ctl] and [x_start | ctl_start] are contiguous on the device as they are in the pinned block
        HIP_OK(hipMemcpyAsync(c->fb.x, h_x, b_x + b_ctl, hipMemcpyHostToDevice, c->stream));
        HIP_OK(hipMemcpyAsync(c->fb.x_start, h_x, b_x + b_ctl, hipMemcpyHostToDevice, c->stream));
    } else {
        HIP_OK(hipMemcpyAsync(c->fb.x, h_x, b_x, hipMemcpyHostToDevice, c->stream));
        HIP_OK(hipMemcpyAsync(c->fb.ctl, h_ctl, b_ctl, hipMemcpyHostToDevice, c->stream));
        HIP_OK(hipMemcpyAsync(c->fb.x_start, c->fb.x, b_x, hipMemcpyDeviceToDevice, c->stream));
        HIP_OK(hipMemcpyAsync(c->fb.ctl_start, c->fb.ctl, b_ctl, hipMemcpyDeviceToDevice, c->stream));
    }
    c->frames_valid = c->state_valid = true;
    c->results_fresh = false;
    if (run_optimize(c, o)) { c->results_fresh = false; return 1; }
    if (!c->results_fresh) { launch_pack_results(c, nframes, c->fb.results, stride); c->results_fresh = true; }      // (a call without ICP iterations has no closing k_lbs launch)
    HIP_OK(hipMemcpyAsync(h_res, c->fb.results, b_res, hipMemcpyDeviceToHost, c->stream));
    double* h_posed = (double*)(pin + b_data + b_lab + b_x + b_ctl + b_res);
    if (want_posed) {      // frame 0's posed outputs into the pinned block, behind the fit, in front of the one synchronisation
        HIP_OK(hipMemcpyAsync(h_posed, c->fb.cloud, (size_t)3 * d.V * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_OK(hipMemcpyAsync(h_posed + 3 * d.V, c->fb.jointpos, (size_t)3 * d.J * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_OK(hipMemcpyAsync(h_posed + 3 * d.V + 3 * d.J, c->fb.jointtrans, (size_t)12 * d.J * 8, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_OK(hipStreamSynchronize(c->stream));      // the one synchronisation of the call
    int bad = -1, nbad = 0;
    unsigned badbits = 0;
    for (int f = 0; f < nframes; ++f) {
        const double* x = h_res + (size_t)f * stride;
        if (x[xs + 7] != 0.0) { if (bad < 0) { bad = f; badbits = (unsigned)x[xs + 7]; } ++nbad; }
    }
    if (bad >= 0) {     // reported once, then cleared (download_state's rule)
        char msg[256];
        snprintf(msg, sizeof msg, "optimize: %d frame(s) carry a device fault (first: frame %d, bits 0x%x%s); their result is not valid", nbad, bad,
                 badbits, (badbits & AVT_FAULT_RIDE_TIMEOUT) ? ": a solver gave up waiting for the in-launch reduction" : "");
        HIP_OK(hipMemsetAsync(c->fb.fault, 0, (size_t)c->fb.max_frames * sizeof(unsigned), c->stream));
        HIP_OK(hipStreamSynchronize(c->stream));
        c->results_fresh = false;      //  the records carry the word that was just cleared: a later gather packs them again
        avt_set_error(msg);
        return AVT_STATUS_DEVICE_FAULT;
    }
    if (posed_cloud) std::memcpy(posed_cloud, h_posed, (size_t)3 * d.V * 8);
    if (posed_jpos) std::memcpy(posed_jpos, h_posed + 3 * d.V, (size_t)3 * d.J * 8);
    if (posed_jtrans) std::memcpy(posed_jtrans, h_posed + 3 * d.V + 3 * d.J, (size_t)12 * d.J * 8);
    for (int f = 0; f < nframes; ++f) {
        const double* x = h_res + (size_t)f * stride;
        std::copy(x, x + 3, p + 3 * (size_t)f);
        std::copy(x + 3, x + 3 + 4 * d.J, q + (size_t)4 * d.J * f);
        std::copy(x + 3 + 4 * d.J, x + xs, w + (size_t)d.K * f);
        if (st) {
            const double* t = x + xs;
            st[f].initial_cost = t[0]; st[f].final_cost = t[1]; st[f].lambda = t[2];
            st[f].num_correspondences = (int)t[3]; st[f].matched_model_points = (int)t[4];
            st[f].gn_iterations = (int)t[5]; st[f].accepted_steps = (int)t[6];
        }
    }
    return 0;
}

}  // namespace

// batch split (avt_shard.cpp): frames received into a device buffer become this context's resident frames
int avt_internal_install_frames(avt_ctx* c, int nframes, const int* counts, const double* data, const int* labels, int device_src) {
    return install_frames(c, nframes, counts, data, labels, device_src != 0);
}

extern "C" {

static int ctx_create_impl(int device, const avt_model* m, int num_parts, const int* part_map, int max_points, int max_frames, avt_ctx** out) {
    if (!m || !part_map || !out) { avt_set_error("avt_ctx_create: null argument"); return 1; }
    if (num_parts <= 0 || num_parts > AVT_MAX_PARTS) { avt_set_error("avt_ctx_create: num_parts out of range (1..64)"); return 1; }
    if (max_points <= 0 || max_frames <= 0) { avt_set_error("avt_ctx_create: max_points/max_frames must be positive"); return 1; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        avt_set_error("avt_ctx_create: no HIP device available (this library has no CPU fallback)");
        return 2;
    }
    if (device < 0 || device >= ndev) { avt_set_error("avt_ctx_create: device index out of range"); return 1; }
    HIP_OK(hipSetDevice(device));
    avt_ctx* c = new avt_ctx();     // value-initialised: every handle starts null, avt_ctx_destroy copes with a partial context
    *out = c;
    c->device = device;
    c->model = m;
    c->profiling = false;
    c->prof_mask = 0xffffffffu;
    c->event_pool_used = 0;
    c->nframes = 0;
    c->ran_icp_iters = 0;
    c->ran_max_iters = 0;
    c->launch_maxN = 0;
    c->tun = tuning_from_environment();
    if (validate_tuning(c->tun)) return 1;
    c->lbs_cleared = false;
    c->scatter_in_compact = false;
    c->nn_from_cloud = false;
    { hipDeviceProp_t pr; c->num_cus = hipGetDeviceProperties(&pr, device) == hipSuccess ? pr.multiProcessorCount : 0; }
    c->vis_frame_min = 0;            // set once the model dimensions are known (below)
    c->graph_clock = 0;
    c->params_valid = false;
    c->frames_valid = c->state_valid = false;
    c->have_moments = c->have_records = false;
    c->concurrent_groups = 1;
    c->host_pin = nullptr; c->host_pin_cap = 0; c->results_fresh = false;
    c->render_zkey = nullptr; c->render_label = nullptr; c->render_block = nullptr; c->render_cap_pix = c->render_cap_blk = 0;
    c->render_mkey = nullptr; c->render_depth = nullptr; c->render_fkey = nullptr; c->render_frank = nullptr; c->render_fedge = nullptr;
    c->render_cap_paint_pix = c->render_cap_paint_face = 0;
    c->render_img_frames = c->render_img_w = c->render_img_h = 0;
    HIP_OK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIP_OK(hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    for (int i = 0; i < AVT_MAX_GROUPS - 1; ++i) {
        HIP_OK(hipStreamCreateWithFlags(&c->side[i], hipStreamNonBlocking));
        HIP_OK(hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming));
    }
    c->cur_stream = c->stream;
    if (avt_solve_set_attributes() || avt_eval_set_attributes() || avt_lbs_set_attributes() || avt_moments_set_attributes()) { avt_set_error("avt_ctx_create: hipFuncSetAttribute failed"); return 1; }
    DeviceModel& dm = c->dm;
    dm.d = m->d;
    if (!dm.d.mom_ok) c->mom_reason = "K + 1 <= 16 and 3 + 3J + K <= 87 required";
    else if (avt_moments_lds_need(dm.d) > avt_moments_lds_cap()) {      // AUTO then keeps the row form, avt_set_data_term(MOMENTS) says why
        dm.d.mom_ok = 0;
        c->mom_reason = "its assembly needs " + std::to_string(avt_moments_lds_need(dm.d)) + " bytes of LDS per workgroup, " + std::to_string(avt_moments_lds_cap()) + " available";
    }
    // visibility as one workgroup per frame when the frame's x, y fit the LDS
    c->vis_frame_min = avt_visibility_frame_lds(m->d) <= 150 * 1024 ? c->tun.vis_frame_min : 0;
    dm.d.num_parts = num_parts;
    const int V = dm.d.V, J = dm.d.J;
    c->part_map.assign(part_map, part_map + J);   // >= J entries (AvatarOptimizer.cpp:1229)
    // model part buckets (AvatarOptimizer.cpp:1227-1243)
    std::vector<int> pov(V), pstart(num_parts + 1, 0), pverts(V), ppos(V);
    for (int v = 0; v < V; ++v) {
        const int q = part_map[m->main_joint[v]];
        if (q < 0 || q >= num_parts) { avt_set_error("avt_ctx_create: part_map entry out of [0,num_parts)"); return 1; }
        pov[v] = q;
        pstart[q + 1]++;
    }
    for (int q = 0; q < num_parts; ++q) pstart[q + 1] += pstart[q];
    {
        std::vector<int> fill(pstart.begin(), pstart.end() - 1);
        for (int v = 0; v < V; ++v) { const int pos = fill[pov[v]]++; pverts[pos] = v; ppos[v] = pos; }
    }
    if (dev_upload(c, &dm.vrec, m->vrec) || dev_upload(c, &dm.shape_planes, m->shape_planes) || dev_upload(c, &dm.lbs_w, m->lbs_w) || dev_upload(c, &dm.lbs_j, m->lbs_j) ||
        dev_upload(c, &dm.asg_w, m->asg_w) || dev_upload(c, &dm.asg_j, m->asg_j) || dev_upload(c, &dm.anc_n, m->anc_n) ||
        dev_upload(c, &dm.anc, m->anc) || dev_upload(c, &dm.mesh, m->mesh_soa) || dev_upload(c, &dm.parent, m->parent) || dev_upload(c, &dm.jlevel, m->jlevel) || dev_upload(c, &dm.fk_items, m->fk_items) || dev_upload(c, &dm.tile_col, m->tile_col) || dev_upload(c, &dm.deal_col, m->deal_col) || dev_upload(c, &dm.tile_param, m->tile_param) ||
        dev_upload(c, &dm.joint_col, m->joint_col) || dev_upload(c, &dm.vorder, m->vorder) || dev_upload(c, &dm.vmask, m->vmask) || dev_upload(c, &dm.fk_level_off, m->fk_level_off) || dev_upload(c, &dm.fk_titems, m->fk_titems) ||
        dev_upload(c, &dm.jsr_base, m->jsr_base) || dev_upload(c, &dm.jsr, m->jsr) || dev_upload(c, &dm.S, m->S) ||
        dev_upload(c, &dm.Sp, m->Sp) || dev_upload(c, &dm.prior_mean, m->prior_mean) || dev_upload(c, &dm.prior_prec, m->prior_prec) ||
        dev_upload(c, &dm.prior_clog, m->prior_clog) || dev_upload(c, &dm.part_of_vertex, pov) ||
        dev_upload(c, &dm.part_start, pstart) || dev_upload(c, &dm.part_vertices, pverts) || dev_upload(c, &dm.part_pos, ppos) ||
        dev_upload(c, &dm.mom_pair, m->mom_pair) || dev_upload(c, &dm.mom_lstart, m->mom_lstart) || dev_upload(c, &dm.mom_lv, m->mom_lv) || dev_upload(c, &dm.mom_lw, m->mom_lw) ||
        dev_upload(c, &dm.mom_psi, m->mom_psi) || dev_upload(c, &dm.mom_opk_start, m->mom_opk_start) || dev_upload(c, &dm.mom_opk, m->mom_opk) ||
        dev_upload(c, &dm.mom_sub_start, m->mom_sub_start) || dev_upload(c, &dm.mom_sub, m->mom_sub) || dev_upload(c, &dm.mom_m1_start, m->mom_m1_start) ||
        dev_upload(c, &dm.mom_m1, m->mom_m1) || dev_upload(c, &dm.mom_s2_start, m->mom_s2_start) || dev_upload(c, &dm.mom_s2, m->mom_s2) || dev_upload(c, &dm.mom_s2_jj, m->mom_s2_jj) || dev_upload(c, &dm.mom_z2_jj, m->mom_z2_jj) || dev_upload(c, &dm.mom_tab16, m->mom_tab16))
        return 1;
    FrameBuffers& fb = c->fb;
    std::memset(&fb, 0, sizeof(fb));
    fb.max_frames = max_frames;
    fb.max_points = max_points;
    fb.G = choose_G(max_frames, 1, c->tun);
    fb.const_blocks = (max_points + 255) / 256;
    fb.bucket_tiles = (max_points + 2047) / 2048;
    const size_t FN = (size_t)max_frames * max_points, FV = (size_t)max_frames * V;
    const AvtDims& d = dm.d;
    // eval workgroups over all frames, for every way run_optimize may split them into groups
    size_t part_cap = 0;
    avt_tuning widest = c->tun;      // (the largest cap a later avt_ctx_set_tuning may ask for)
    widest.gcap = AVT_G_MAX; widest.g = 0;
    for (int nf = 1; nf <= max_frames; ++nf)
        for (int k = 1; k <= std::min(AVT_MAX_GROUPS, nf); ++k) part_cap = std::max(part_cap, (size_t)nf * std::max(choose_G((nf + k - 1) / k, 1, widest), choose_G((nf + k - 1) / k, 2, widest)));
    char* cntsum = nullptr;
    char* state_block = nullptr;
    if (dev_alloc(c, &fb.data_raw, FN * 3) || dev_alloc(c, &fb.labels_raw, FN) || dev_alloc(c, &fb.dx, FN) || dev_alloc(c, &fb.dy, FN) ||
        dev_alloc(c, &fb.dz, FN) || dev_alloc(c, &fb.dorig, FN) || dev_alloc(c, &fb.part_off, (size_t)max_frames * (num_parts + 1)) || dev_alloc(c, &fb.part_cnt, (size_t)max_frames * 2 * (AVT_MAX_PARTS + 1)) || dev_alloc(c, &fb.tile_hist, (size_t)max_frames * ((max_points + 2047) / 2048) * (AVT_MAX_PARTS + 1)) ||
        dev_alloc(c, &fb.corr, FN) || dev_alloc(c, &fb.corr_sorted, FN) || dev_alloc(c, &fb.cloud, FV * 3) || dev_alloc(c, &fb.pcx, FV) ||
        dev_alloc(c, &fb.pcy, FV) || dev_alloc(c, &fb.pcz, FV) || dev_alloc(c, &fb.visible, FV) || dev_alloc(c, &fb.vcx, FV) || dev_alloc(c, &fb.vcy, FV) ||
        dev_alloc(c, &fb.vcz, FV) || dev_alloc(c, &fb.vcid, FV) || dev_alloc(c, &fb.vcount, (size_t)max_frames * num_parts) || dev_alloc(c, &fb.vis_sorted, FV) || dev_alloc(c, &fb.ride_ctr, (size_t)max_frames) || dev_alloc(c, &fb.fault, (size_t)max_frames) || dev_alloc(c, &fb.spec, (size_t)max_frames) || dev_alloc(c, &fb.snap, (size_t)max_frames) || dev_alloc(c, &fb.x_spec, (size_t)max_frames * AVT_MAX_SPEC * d.xsize) || dev_alloc(c, &fb.prep_spec, (size_t)max_frames * AVT_MAX_SPEC * d.prep_size) ||
        dev_alloc(c, &cntsum, FV * (sizeof(int) + 3 * sizeof(long long)) + 64) || dev_alloc(c, &fb.matched, FV) ||
        dev_alloc(c, &fb.const_part, (size_t)max_frames * fb.const_blocks) ||
        dev_alloc(c, &state_block, 2 * ((size_t)max_frames * 2 * d.xsize * sizeof(double) + (size_t)max_frames * sizeof(AvtFrameCtl))) ||
        dev_alloc(c, &fb.prep, (size_t)max_frames * 2 * d.prep_size) ||
        dev_alloc(c, &fb.rec, (size_t)max_frames * d.nb_max * 4 * d.rec_quad) || dev_alloc(c, &fb.partial, part_cap * d.NPAIR * 256) || dev_alloc(c, &fb.wmask, part_cap) || dev_alloc(c, &fb.bmask, (size_t)max_frames * d.nb_max) || dev_alloc(c, &fb.erange, (size_t)max_frames * AVT_ERANGE) || dev_alloc(c, &fb.Hraw, (size_t)max_frames * 2 * d.HS * d.HS) ||
        dev_alloc(c, &fb.prior, (size_t)max_frames * 2 * AVT_MAX_COMPS * AVT_PRIOR_STRIDE) ||
        dev_alloc(c, &fb.jointpos, (size_t)max_frames * 3 * J) || dev_alloc(c, &fb.jointtrans, (size_t)max_frames * 12 * J) ||
        dev_alloc(c, &fb.trace, (size_t)max_frames * 64) || dev_alloc(c, &fb.results, (size_t)max_frames * (d.xsize + 8)))
        return 1;
    {   // the states and control blocks and their start copies are ONE allocation, [x | ctl | x_start | ctl_start]: a context used at its full frame
        // count (the facade's: one frame) uploads state + control block with one copy and keeps the start copy with one more (round 6: they were four)
        const size_t bx = (size_t)max_frames * 2 * d.xsize * sizeof(double), bc = (size_t)max_frames * sizeof(AvtFrameCtl);
        fb.x = (double*)state_block; fb.ctl = (AvtFrameCtl*)(state_block + bx);
        fb.x_start = (double*)(state_block + bx + bc); fb.ctl_start = (AvtFrameCtl*)(state_block + 2 * bx + bc);
    }
    fb.use_moments = 0;
    if (d.mom_ok) {      // moment form of the data term (avt_moments.hip): T per (frame, joint pair), D per (frame, joint), scratch of the assembly
        if (dev_alloc(c, &fb.mom_T, (size_t)max_frames * avt_moments_T_doubles(d)) || dev_alloc(c, &fb.mom_D, (size_t)max_frames * J * d.mom_npsi * 3) ||
            dev_alloc(c, &fb.mom_E, (size_t)max_frames * 2) || dev_alloc(c, &fb.mom_rec, (size_t)max_frames * avt_moments_frame_scratch(d)))
            return 1;
        HIP_OK(hipMemsetAsync(fb.mom_D, 0, (size_t)max_frames * J * d.mom_npsi * 3 * sizeof(double), c->stream));      // joints no vertex is assigned to keep zeros
    }
    c->data_term = AVT_DATA_TERM_AUTO;
    // (tun.mom_min_frames = 8, round 5 - ms per step moments / rows, frames per launch [frames per GPU]: 4 [4] 0.701 / 0.648, 6 [6] 0.710 / 0.692, 8 [8] 0.709 / 0.724,
    //  12 [12] 0.747 / 0.787, 16 [16] 0.737 / 0.800, 12 [24] 0.777 / 0.877, 16 [32] 0.838 / 0.938, 24 [48] 0.934 / 1.075, 28 [56] 0.985 / 1.186; dense frames 8 [8] 0.757 / 0.777,
    //  16 [16] 0.817 / 0.884; tools/ab_env.sh AVT_MOM_MIN_FRAMES.  Round 4's crossover was 32 frames per launch: the assembly as role workgroups, the decide-only closing
    //  pass and the XCD frame mapping took 15 % off the moment form's GN iteration since)
    c->last_run_moments = false;
    {
        AvtRunParams* pr = nullptr;
        if (dev_alloc(c, &pr, 1)) return 1;
        fb.params = pr;
    }
    fb.spec_frames = std::min(max_frames, AVT_SPEC_FRAMES);
    if (dev_alloc(c, &fb.partial_spec, (size_t)fb.spec_frames * AVT_MAX_SPEC * AVT_G_MAX * 256) || dev_alloc(c, &fb.wmask_spec, (size_t)fb.spec_frames * AVT_MAX_SPEC * AVT_G_MAX) ||
        dev_alloc(c, &fb.prior_spec, (size_t)fb.spec_frames * AVT_MAX_SPEC * AVT_MAX_COMPS * AVT_PRIOR_STRIDE))
        return 1;
    fb.fsum = (long long*)cntsum;                                   // 8-byte aligned first
    fb.cnt = (int*)(cntsum + FV * 3 * sizeof(long long));
    HIP_OK(hipMemsetAsync(fb.trace, 0, (size_t)max_frames * 64 * sizeof(double), c->stream));
    HIP_OK(hipMemsetAsync(fb.ctl, 0, (size_t)max_frames * sizeof(AvtFrameCtl), c->stream));
    HIP_OK(hipMemsetAsync(fb.ride_ctr, 0, (size_t)max_frames * sizeof(unsigned), c->stream));
    HIP_OK(hipMemsetAsync(fb.fault, 0, (size_t)max_frames * sizeof(unsigned), c->stream));
    fb.ride_timeout = c->tun.ride_timeout_us * 100;      // wall_clock64() ticks at 100 MHz; 0 makes every wait that is not already satisfied fail (tests)
    fb.xcd_frames = c->tun.xcd_frames;
    HIP_OK(hipMemsetAsync(fb.spec, 0, (size_t)max_frames * sizeof(AvtSpecCtl), c->stream));
    fb.nspec = 0; fb.seq = 0;
    HIP_OK(hipMemsetAsync(fb.part_cnt, 0, (size_t)max_frames * 2 * (AVT_MAX_PARTS + 1) * sizeof(int), c->stream));   // invariant of launch_bucket
    HIP_OK(hipStreamSynchronize(c->stream));
    if (c->tun.debug) avt_eval_report_occupancy(dm.d);
    return 0;
}


int avt_ctx_create(int device, const avt_model* m, int num_parts, const int* part_map, int max_points, int max_frames, avt_ctx** out) {
    if (out) *out = nullptr;
    int rc = 1;
    try {
        rc = ctx_create_impl(device, m, num_parts, part_map, max_points, max_frames, out);
    } catch (const std::exception& e) { avt_set_error(std::string("avt_ctx_create: ") + e.what()); rc = 1; }
    if (rc != 0 && out && *out) {     // streams, events and every allocation made so far are released
        const std::string keep = avt_last_error();
        avt_ctx_destroy(*out);
        *out = nullptr;
        avt_set_error(keep);
    }
    return rc;
}

void avt_ctx_destroy(avt_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (void* p : c->allocs) (void)hipFree(p);
    if (c->host_pin) (void)hipHostFree(c->host_pin);
    if (c->render_zkey) (void)hipFree(c->render_zkey);
    if (c->render_label) (void)hipFree(c->render_label);
    if (c->render_block) (void)hipFree(c->render_block);
    for (void* p : {(void*)c->render_mkey, (void*)c->render_depth, (void*)c->render_fkey, (void*)c->render_frank, (void*)c->render_fedge})
        if (p) (void)hipFree(p);
    for (hipEvent_t e : c->event_pool) (void)hipEventDestroy(e);
    { std::lock_guard<std::mutex> graph_lock(g_graph_mutex); for (auto& g : c->graphs) (void)hipGraphExecDestroy(g.exec); }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    for (int i = 0; i < AVT_MAX_GROUPS - 1; ++i) {
        if (c->ev_join[i]) (void)hipEventDestroy(c->ev_join[i]);
        if (c->side[i]) (void)hipStreamDestroy(c->side[i]);
    }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int avt_sync(avt_ctx* c) {
    HIP_OK(hipStreamSynchronize(c->stream));
    return 0;
}

int avt_lbs_update(avt_ctx* c, int nframes, const double* w, const double* p, const double* R, double* cloud, double* joint_pos,
                   double* joint_trans) {
    AVT_API_GUARD_BEGIN
    if (!c || !w || !p || !R) { avt_set_error("avt_lbs_update: null argument"); return 1; }
    const AvtDims& d = c->dm.d;
    if (nframes <= 0 || nframes > c->fb.max_frames) { avt_set_error("avt_lbs_update: nframes out of range"); return 1; }
    HIP_OK(hipSetDevice(c->device));
    c->frames_valid = c->state_valid = false;    // frame slots' posed clouds / skeleton scratch are overwritten
    // stage the parameters in the (otherwise unused here) prep buffer: w | p | R
    double* dw = c->fb.prep;
    double* dp = dw + (size_t)nframes * d.K;
    double* dR = dp + (size_t)nframes * 3;
    if ((size_t)nframes * (d.K + 3 + 9 * d.J) > (size_t)c->fb.max_frames * 2 * d.prep_size) { avt_set_error("avt_lbs_update: staging overflow"); return 1; }
    HIP_OK(hipMemcpyAsync(dw, w, (size_t)nframes * d.K * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_OK(hipMemcpyAsync(dp, p, (size_t)nframes * 3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_OK(hipMemcpyAsync(dR, R, (size_t)nframes * 9 * d.J * sizeof(double), hipMemcpyHostToDevice, c->stream));
    { ProfScope ps(c, AVT_K_LBS); launch_lbs(c, nframes, nullptr, dw, dp, dR, 0, -1); }
    if (check_launch("k_lbs")) return 1;
    if (cloud) HIP_OK(hipMemcpyAsync(cloud, c->fb.cloud, (size_t)nframes * 3 * d.V * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (joint_pos) HIP_OK(hipMemcpyAsync(joint_pos, c->fb.jointpos, (size_t)nframes * 3 * d.J * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (joint_trans) HIP_OK(hipMemcpyAsync(joint_trans, c->fb.jointtrans, (size_t)nframes * 12 * d.J * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    return 0;
    AVT_API_GUARD_END("avt_lbs_update")
}

int avt_visibility(avt_ctx* c, const double* cloud, int enable, unsigned char* visible) {
    if (!c || !cloud || !visible) { avt_set_error("avt_visibility: null argument"); return 1; }
    const AvtDims& d = c->dm.d;
    HIP_OK(hipSetDevice(c->device));
    c->frames_valid = c->state_valid = false;
    HIP_OK(hipMemcpyAsync(c->fb.cloud, cloud, (size_t)3 * d.V * sizeof(double), hipMemcpyHostToDevice, c->stream));
    { ProfScope ps(c, AVT_K_VISIBILITY); launch_visibility(c, 1, enable); }
    if (check_launch("k_visibility")) return 1;
    HIP_OK(hipMemcpyAsync(visible, c->fb.visible, (size_t)d.V, hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    return 0;
}

int avt_nn(avt_ctx* c, const double* model_cloud, const unsigned char* visible, const double* data, const int* labels, int N, int* out) {
    AVT_API_GUARD_BEGIN
    if (!c || !model_cloud || !visible || !data || !labels || !out) { avt_set_error("avt_nn: null argument"); return 1; }
    const AvtDims& d = c->dm.d;
    const int V = d.V;
    HIP_OK(hipSetDevice(c->device));
    if (N == 0) return 0;
    const int offs[2] = {0, N};
    if (upload_frames(c, 1, data, labels, offs)) return 1;
    c->frames_valid = c->state_valid = false;    // frame slot 0 is scratch for this call
    // part-sorted SoA copy of the model cloud
    std::vector<int> ppos(V);
    HIP_OK(hipMemcpyAsync(ppos.data(), c->dm.part_pos, (size_t)V * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    std::vector<double> px(V), py(V), pz(V);
    for (int v = 0; v < V; ++v) { px[ppos[v]] = model_cloud[3 * v]; py[ppos[v]] = model_cloud[3 * v + 1]; pz[ppos[v]] = model_cloud[3 * v + 2]; }
    HIP_OK(hipMemcpyAsync(c->fb.pcx, px.data(), V * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_OK(hipMemcpyAsync(c->fb.pcy, py.data(), V * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_OK(hipMemcpyAsync(c->fb.pcz, pz.data(), V * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_OK(hipMemcpyAsync(c->fb.visible, visible, (size_t)V, hipMemcpyHostToDevice, c->stream));
    AvtFrameCtl ctl;
    std::memset(&ctl, 0, sizeof(ctl));
    ctl.N = N;
    HIP_OK(hipMemcpyAsync(c->fb.ctl, &ctl, sizeof(ctl), hipMemcpyHostToDevice, c->stream));
    { ProfScope ps(c, AVT_K_BUCKET); launch_bucket(c, 1, true); }
    { ProfScope ps(c, AVT_K_NN); launch_nn(c, 1); }
    if (check_launch("k_nn")) return 1;
    HIP_OK(hipMemcpyAsync(out, c->fb.corr, (size_t)N * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    return 0;
    AVT_API_GUARD_END("avt_nn")
}

int avt_synth_render_frames(avt_ctx* c, int nframes, const double* w, const double* p, const double* R, double fx, double fy, double cx,
                            double cy, int width, int height, int* points_per_frame) {
    return avt_synth_render_frames_mode(c, nframes, w, p, R, fx, fy, cx, cy, width, height, AVT_RENDER_ZBUFFER, points_per_frame);
}

int avt_synth_render_frames_mode(avt_ctx* c, int nframes, const double* w, const double* p, const double* R, double fx, double fy, double cx,
                                 double cy, int width, int height, int mode, int* points_per_frame) {
    AVT_API_GUARD_BEGIN
    if (!c || !w || !p || !R || width <= 0 || height <= 0) { avt_set_error("avt_synth_render_frames: bad argument"); return 1; }
    if (mode != AVT_RENDER_ZBUFFER && mode != AVT_RENDER_PAINTER) { avt_set_error("avt_synth_render_frames: unknown mode"); return 1; }
    const bool painter = mode == AVT_RENDER_PAINTER;
    c->render_img_frames = 0;
    const AvtDims& d = c->dm.d;
    if (nframes <= 0 || nframes > c->fb.max_frames) { avt_set_error("avt_synth_render_frames: nframes out of range"); return 1; }
    HIP_OK(hipSetDevice(c->device));
    // pose the ground-truth avatars (Avatar::update) for all frames
    double* dw = c->fb.prep;
    double* dp = dw + (size_t)nframes * d.K;
    double* dR = dp + (size_t)nframes * 3;
    HIP_OK(hipMemcpyAsync(dw, w, (size_t)nframes * d.K * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_OK(hipMemcpyAsync(dp, p, (size_t)nframes * 3 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_OK(hipMemcpyAsync(dR, R, (size_t)nframes * 9 * d.J * sizeof(double), hipMemcpyHostToDevice, c->stream));
    launch_lbs(c, nframes, nullptr, dw, dp, dR, 0, -1);
    // rasterise in chunks of frames (8 bytes of z-buffer key per pixel)
    const size_t npix = (size_t)width * height;
    const int chunk = std::max(1, std::min(nframes, (int)((256ull << 20) / (npix * (painter ? 21 : 9) + 64))));
    const int nb = (int)((npix + 255) / 256);
    // z-buffer keys, labels and block counts live in the context and only grow (no hipMalloc / hipFree per call)
    if (c->render_cap_pix < npix * chunk || c->render_cap_blk < (size_t)nb * chunk) {
        HIP_OK(hipStreamSynchronize(c->stream));
        if (c->render_zkey) (void)hipFree(c->render_zkey);
        if (c->render_label) (void)hipFree(c->render_label);
        if (c->render_block) (void)hipFree(c->render_block);
        c->render_zkey = nullptr; c->render_label = nullptr; c->render_block = nullptr; c->render_cap_pix = c->render_cap_blk = 0;
        HIP_OK(hipMalloc((void**)&c->render_zkey, npix * chunk * sizeof(unsigned long long)));
        HIP_OK(hipMalloc((void**)&c->render_label, npix * chunk));
        HIP_OK(hipMalloc((void**)&c->render_block, (size_t)nb * chunk * sizeof(int)));
        c->render_cap_pix = npix * chunk; c->render_cap_blk = (size_t)nb * chunk;
    }
    const size_t nface = (size_t)d.F * chunk;
    if (painter && (c->render_cap_paint_pix < npix * chunk || c->render_cap_paint_face < nface)) {
        HIP_OK(hipStreamSynchronize(c->stream));
        for (void* q : {(void*)c->render_mkey, (void*)c->render_depth, (void*)c->render_fkey, (void*)c->render_frank, (void*)c->render_fedge})
            if (q) (void)hipFree(q);
        c->render_mkey = nullptr; c->render_depth = nullptr; c->render_fkey = nullptr; c->render_frank = nullptr; c->render_fedge = nullptr;
        c->render_cap_paint_pix = c->render_cap_paint_face = 0;
        HIP_OK(hipMalloc((void**)&c->render_mkey, npix * chunk * sizeof(unsigned long long)));
        HIP_OK(hipMalloc((void**)&c->render_depth, npix * chunk * sizeof(float)));
        HIP_OK(hipMalloc((void**)&c->render_fkey, nface * sizeof(float)));
        HIP_OK(hipMalloc((void**)&c->render_frank, nface * sizeof(int)));
        HIP_OK(hipMalloc((void**)&c->render_fedge, nface));
        c->render_cap_paint_pix = npix * chunk; c->render_cap_paint_face = nface;
    }
    int rc = 0;
    for (int f0 = 0; f0 < nframes && !rc; f0 += chunk) {
        c->fb.f0 = f0;
        const int nf = std::min(chunk, nframes - f0);
        if (painter)
            rc = avt_paint_enqueue(c, nf, c->dm.part_of_vertex, c->render_zkey, c->render_mkey, c->render_fkey, c->render_frank, c->render_fedge,
                                   c->render_depth, c->render_label, c->render_block, fx, fy, cx, cy, width, height);
        else
            rc = avt_render_enqueue(c, nf, c->dm.part_of_vertex, c->render_zkey, c->render_label, c->render_block, fx, fy, cx, cy, width, height);
    }
    c->fb.f0 = 0;
    if (rc) { avt_set_error("avt_synth_render_frames: render launch failed"); return 1; }
    std::vector<AvtFrameCtl> ctl(nframes);
    HIP_OK(hipMemcpyAsync(ctl.data(), c->fb.ctl, ctl.size() * sizeof(AvtFrameCtl), hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    c->frames_valid = c->state_valid = false;
    c->have_moments = c->have_records = false;
    c->results_fresh = false;
    c->nframes = nframes;
    c->frame_N.assign(nframes, 0);
    c->frame_off.assign(nframes + 1, 0);
    int mx = 0;
    for (int f = 0; f < nframes; ++f) {
        if (ctl[f].T > c->fb.max_points) { avt_set_error("avt_synth_render_frames: a rendered frame has more points than max_points_per_frame"); return 1; }
        c->frame_N[f] = ctl[f].N;
        c->frame_off[f + 1] = c->frame_off[f] + ctl[f].N;
        mx = std::max(mx, ctl[f].N);
        if (points_per_frame) points_per_frame[f] = ctl[f].N;
    }
    c->launch_maxN = std::max(mx, std::min(c->fb.max_points, ((mx + 2047) / 2048) * 2048));
    // the render kernels wrote N into the working control blocks; the start copies follow (avt_state_reset)
    HIP_OK(hipMemcpy2DAsync(&c->fb.ctl_start->N, sizeof(AvtFrameCtl), c->frame_N.data(), sizeof(int), sizeof(int), (size_t)nframes, hipMemcpyHostToDevice, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    c->frames_valid = true;
    if (painter && nframes <= chunk) { c->render_img_frames = nframes; c->render_img_w = width; c->render_img_h = height; }
    return 0;
    AVT_API_GUARD_END("avt_synth_render_frames")
}

int avt_synth_render_images(avt_ctx* c, int frame, float* depth, unsigned char* part_mask) {
    AVT_API_GUARD_BEGIN
    if (!c) { avt_set_error("avt_synth_render_images: null context"); return 1; }
    if (c->render_img_frames == 0) {
        avt_set_error("avt_synth_render_images: no images retained (needs a preceding AVT_RENDER_PAINTER call whose frames fit one scratch chunk)");
        return 1;
    }
    if (frame < 0 || frame >= c->render_img_frames) { avt_set_error("avt_synth_render_images: frame out of range"); return 1; }
    HIP_OK(hipSetDevice(c->device));
    const size_t npix = (size_t)c->render_img_w * c->render_img_h;
    if (depth) HIP_OK(hipMemcpyAsync(depth, c->render_depth + (size_t)frame * npix, npix * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    if (part_mask) HIP_OK(hipMemcpyAsync(part_mask, c->render_label + (size_t)frame * npix, npix, hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    return 0;
    AVT_API_GUARD_END("avt_synth_render_images")
}

int avt_frames_download(avt_ctx* c, int frame, double* data_3xN, int* labels) {
    if (!c || !c->frames_valid || frame < 0 || frame >= c->nframes) { avt_set_error("avt_frames_download: bad argument"); return 1; }
    HIP_OK(hipSetDevice(c->device));
    const size_t N = c->frame_N[frame];
    if (data_3xN) HIP_OK(hipMemcpyAsync(data_3xN, c->fb.data_raw + (size_t)frame * c->fb.max_points * 3, N * 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (labels) HIP_OK(hipMemcpyAsync(labels, c->fb.labels_raw + (size_t)frame * c->fb.max_points, N * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    return 0;
}

int avt_frames_upload(avt_ctx* c, int nframes, const double* data, const int* labels, const int* frame_offsets) {
    AVT_API_GUARD_BEGIN
    if (!c || !data || !labels || !frame_offsets) { avt_set_error("avt_frames_upload: null argument"); return 1; }
    HIP_OK(hipSetDevice(c->device));
    if (upload_frames(c, nframes, data, labels, frame_offsets)) return 1;
    HIP_OK(hipStreamSynchronize(c->stream));
    return 0;
    AVT_API_GUARD_END("avt_frames_upload")
}

int avt_state_upload(avt_ctx* c, int nframes, const double* p, const double* q, const double* w) {
    AVT_API_GUARD_BEGIN
    if (!c || !p || !q || !w) { avt_set_error("avt_state_upload: null argument"); return 1; }
    HIP_OK(hipSetDevice(c->device));
    return upload_state(c, nframes, p, q, w);
    AVT_API_GUARD_END("avt_state_upload")
}

int avt_optimize_resident(avt_ctx* c, const avt_options* opt) {
    AVT_API_GUARD_BEGIN
    if (!c || !opt) { avt_set_error("avt_optimize_resident: null argument"); return 1; }
    HIP_OK(hipSetDevice(c->device));
    const int rc = run_optimize(c, opt);
    if (rc) c->results_fresh = false;      //  a call that failed left no result records
    return rc;
    AVT_API_GUARD_END("avt_optimize_resident")
}

int avt_state_reset(avt_ctx* c) {
    if (!c || c->nframes <= 0 || !c->state_valid) { avt_set_error("avt_state_reset: no state resident"); return 1; }
    HIP_OK(hipSetDevice(c->device));
    launch_state_reset(c, c->nframes);
    c->results_fresh = false;
    return check_launch("k_state_reset");
}

int avt_state_download(avt_ctx* c, double* p, double* q, double* w, avt_stats* stats) {
    AVT_API_GUARD_BEGIN
    if (!c) { avt_set_error("avt_state_download: null context"); return 1; }
    HIP_OK(hipSetDevice(c->device));
    return download_state(c, p, q, w, stats);
    AVT_API_GUARD_END("avt_state_download")
}

int avt_optimize_batch(avt_ctx* c, int nframes, const double* data, const int* labels, const int* frame_offsets, const avt_options* opt,
                       double* p, double* q, double* w, avt_stats* stats) {
    AVT_API_GUARD_BEGIN
    if (!c || !data || !labels || !frame_offsets || !opt || !p || !q || !w) { avt_set_error("avt_optimize_batch: null argument"); return 1; }
    HIP_OK(hipSetDevice(c->device));
    return optimize_host_to_host(c, nframes, data, labels, frame_offsets, opt, p, q, w, stats);
    AVT_API_GUARD_END("avt_optimize_batch")
}

int avt_optimize(avt_ctx* c, const double* data, const int* labels, int N, const avt_options* opt, double* p, double* q, double* w,
                 avt_stats* stats) {
    const int offs[2] = {0, N};
    return avt_optimize_batch(c, 1, data, labels, offs, opt, p, q, w, stats);
}

int avt_optimize_posed(avt_ctx* c, const double* data, const int* labels, int N, const avt_options* opt, double* p, double* q, double* w,
                       avt_stats* stats, double* cloud, double* joint_pos, double* joint_trans) {
    AVT_API_GUARD_BEGIN
    if (!c || !data || !labels || !opt || !p || !q || !w) { avt_set_error("avt_optimize_posed: null argument"); return 1; }
    if (opt->icp_iters <= 0 && (cloud || joint_pos || joint_trans)) { avt_set_error("avt_optimize_posed: no ICP iteration, no closing update(): nothing posed to return"); return 1; }
    HIP_OK(hipSetDevice(c->device));
    const int offs[2] = {0, N};
    return optimize_host_to_host(c, 1, data, labels, offs, opt, p, q, w, stats, cloud, joint_pos, joint_trans);
    AVT_API_GUARD_END("avt_optimize_posed")
}

int avt_get_correspondences(avt_ctx* c, int frame, int* out) {
    if (!c || !out || frame < 0 || frame >= c->nframes) { avt_set_error("avt_get_correspondences: bad argument"); return 1; }
    HIP_OK(hipSetDevice(c->device));
    HIP_OK(hipMemcpyAsync(out, c->fb.corr + (size_t)frame * c->fb.max_points, (size_t)c->frame_N[frame] * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    return 0;
}

int avt_get_cloud(avt_ctx* c, int frame, double* cloud) {
    if (!c || !cloud || frame < 0 || frame >= c->fb.max_frames) { avt_set_error("avt_get_cloud: bad argument"); return 1; }
    HIP_OK(hipSetDevice(c->device));
    HIP_OK(hipMemcpyAsync(cloud, c->fb.cloud + (size_t)frame * 3 * c->dm.d.V, (size_t)3 * c->dm.d.V * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    return 0;
}

int avt_get_posed(avt_ctx* c, int frame, double* cloud, double* joint_pos, double* joint_trans) {
    if (!c || frame < 0 || frame >= c->fb.max_frames) { avt_set_error("avt_get_posed: bad argument"); return 1; }
    const AvtDims& d = c->dm.d;
    HIP_OK(hipSetDevice(c->device));
    if (cloud) HIP_OK(hipMemcpyAsync(cloud, c->fb.cloud + (size_t)frame * 3 * d.V, (size_t)3 * d.V * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (joint_pos) HIP_OK(hipMemcpyAsync(joint_pos, c->fb.jointpos + (size_t)frame * 3 * d.J, (size_t)3 * d.J * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (joint_trans) HIP_OK(hipMemcpyAsync(joint_trans, c->fb.jointtrans + (size_t)frame * 12 * d.J, (size_t)12 * d.J * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    return 0;
}

int avt_get_normal_equations(avt_ctx* c, int frame, double* H, double* g, double* cost) {
    AVT_API_GUARD_BEGIN
    if (!c || frame < 0 || frame >= c->nframes) { avt_set_error("avt_get_normal_equations: bad argument"); return 1; }
    if (!c->frames_valid || !c->state_valid || c->ran_icp_iters <= 0) { avt_set_error("avt_get_normal_equations: no optimize call has run on the resident frames"); return 1; }
    const AvtDims& d = c->dm.d;
    HIP_OK(hipSetDevice(c->device));
    // optimize() does not build the system of its last trial point (only that point's cost is needed): evaluate the data term
    // at the CURRENT point here - trial point := current point (SOLVE_INIT), one full evaluation, one reduction - with the
    // correspondences of the last ICP iteration.  The current point, its cost and the LM state are left untouched.
    const int G_keep = c->fb.G;
    c->fb.G = choose_G(c->nframes, 1, c->tun);
    c->fb.f0 = 0;
    c->cur_stream = c->stream;
    // the form of the data term: the one selected (AUTO: the one the last optimize() ran); what that form needs of the resident
    // correspondences - moments or matched-point records - is made here if the last optimize() ran the other form
    const bool want_mom = c->dm.d.mom_ok && (c->data_term == AVT_DATA_TERM_MOMENTS || (c->data_term == AVT_DATA_TERM_AUTO && c->last_run_moments));
    const int use_moments_keep = c->fb.use_moments;      // (restored below: plan_groups / avt_launch_shape read it)
    c->fb.use_moments = want_mom;
    launch_solve(c, c->nframes, SOLVE_INIT);
    if (want_mom) {
        if (!c->have_moments) { c->fb.const_used = (c->launch_maxN + 2047) / 2048; launch_moments(c, c->nframes); c->have_moments = true; }
        launch_assemble(c, c->nframes);
    } else {
        if (!c->have_records) { c->fb.const_used = (c->launch_maxN + 255) / 256; launch_records(c, c->nframes); c->have_records = true; }
        launch_eval(c, c->nframes, false);
        launch_reduce(c, c->nframes);
    }
    c->fb.G = G_keep;
    c->fb.use_moments = use_moments_keep;
    if (check_launch("avt_get_normal_equations")) return 1;
    HIP_OK(hipStreamSynchronize(c->stream));
    AvtFrameCtl ctl;
    HIP_OK(hipMemcpyAsync(&ctl, c->fb.ctl + frame, sizeof(ctl), hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    std::vector<double> buf((size_t)d.HS * d.HS);
    HIP_OK(hipMemcpyAsync(buf.data(), c->fb.Hraw + ((size_t)frame * 2 + (1 - ctl.cur_slot)) * buf.size(), buf.size() * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    for (int r = 0; r < d.P; ++r) {
        if (H) for (int q = 0; q < d.P; ++q) H[(size_t)r * d.P + q] = buf[(size_t)r * d.HS + q];
        if (g) g[r] = buf[(size_t)d.P * d.HS + r];
    }
    if (cost) *cost = ctl.cost_cur;
    return 0;
    AVT_API_GUARD_END("avt_get_normal_equations")
}

int avt_debug_mfma_count(avt_ctx* c, int frame, long long* eval_rows, long long* moments, long long* solve) {
    AVT_API_GUARD_BEGIN
    if (!c || frame < 0 || frame >= c->nframes) { avt_set_error("avt_debug_mfma_count: bad argument"); return 1; }
    if (!c->frames_valid || c->ran_icp_iters <= 0) { avt_set_error("avt_debug_mfma_count: no optimize call has run on the resident frames"); return 1; }
    const AvtDims& d = c->dm.d;
    HIP_OK(hipSetDevice(c->device));
    HIP_OK(hipStreamSynchronize(c->stream));
    if (solve) *solve = avt_solve_mfma_count(d);
    if (moments) {
        *moments = -1;
        if (d.mom_ok) {
            std::vector<int> cnt(d.V);
            HIP_OK(hipMemcpyAsync(cnt.data(), c->fb.cnt + (size_t)frame * d.V, cnt.size() * sizeof(int), hipMemcpyDeviceToHost, c->stream));
            HIP_OK(hipStreamSynchronize(c->stream));
            *moments = avt_moments_mfma_count(c->model, cnt.data());
        }
    }
    if (eval_rows) {
        *eval_rows = -1;
        if (c->have_records && d.NT <= 6) {
            // k_records leaves one word per batch of 16 matched points: bits 0..23 = the live tile pairs (avt_eval.hip); k_eval runs the
            // 12 k-steps (48 rows / 4) of every live pair - the split pair's are dealt over the four waves, 12 in total all the same
            AvtFrameCtl ctl;
            HIP_OK(hipMemcpyAsync(&ctl, c->fb.ctl + frame, sizeof(ctl), hipMemcpyDeviceToHost, c->stream));
            HIP_OK(hipStreamSynchronize(c->stream));
            const int nb = (ctl.M + AVT_EVAL_PTS - 1) / AVT_EVAL_PTS;
            std::vector<int> bm(std::max(nb, 1));
            HIP_OK(hipMemcpyAsync(bm.data(), c->fb.bmask + (size_t)frame * d.nb_max, (size_t)nb * sizeof(int), hipMemcpyDeviceToHost, c->stream));
            HIP_OK(hipStreamSynchronize(c->stream));
            long long n = 0;
            for (int b = 0; b < nb; ++b) n += 12ll * __builtin_popcount((unsigned)bm[b] & 0xffffffu);
            *eval_rows = n;
        }
    }
    return 0;
    AVT_API_GUARD_END("avt_debug_mfma_count")
}

int avt_ctx_get_tuning(avt_ctx* c, avt_tuning* out) {
    if (!c || !out) { avt_set_error("avt_ctx_get_tuning: null argument"); return 1; }
    *out = c->tun;
    return 0;
}

int avt_ctx_set_tuning(avt_ctx* c, const avt_tuning* t) {
    AVT_API_GUARD_BEGIN
    if (!c || !t) { avt_set_error("avt_ctx_set_tuning: null argument"); return 1; }
    if (validate_tuning(*t)) return 1;
    HIP_OK(hipSetDevice(c->device));
    HIP_OK(hipStreamSynchronize(c->stream));
    {   // the launch sequences captured so far were made with the old knobs
        std::lock_guard<std::mutex> graph_lock(g_graph_mutex);
        for (auto& e : c->graphs) (void)hipGraphExecDestroy(e.exec);
        c->graphs.clear();
    }
    c->tun = *t;
    c->vis_frame_min = avt_visibility_frame_lds(c->dm.d) <= 150 * 1024 ? c->tun.vis_frame_min : 0;
    c->fb.ride_timeout = c->tun.ride_timeout_us * 100;
    c->fb.xcd_frames = c->tun.xcd_frames;
    return 0;
    AVT_API_GUARD_END("avt_ctx_set_tuning")
}

int avt_set_data_term(avt_ctx* c, int form) {
    if (!c || (form != AVT_DATA_TERM_ROWS && form != AVT_DATA_TERM_MOMENTS && form != AVT_DATA_TERM_AUTO)) { avt_set_error("avt_set_data_term: bad argument"); return 1; }
    if (form == AVT_DATA_TERM_MOMENTS && !c->dm.d.mom_ok) { avt_set_error("avt_set_data_term: this model has no moment form (" + c->mom_reason + ")"); return 1; }
    c->data_term = form;
    return 0;
}

int avt_get_data_term(avt_ctx* c) { return c ? c->data_term : -1; }

int avt_debug_trace(avt_ctx* c, int frame, double* out64) {
    if (!c || !out64 || frame < 0 || frame >= c->fb.max_frames) { avt_set_error("avt_debug_trace: bad argument"); return 1; }
    HIP_OK(hipSetDevice(c->device));
    HIP_OK(hipStreamSynchronize(c->stream));
    HIP_OK(hipMemcpyAsync(out64, c->fb.trace + (size_t)frame * 64, 64 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_OK(hipStreamSynchronize(c->stream));
    return 0;
}

int avt_launch_shape(avt_ctx* c, int* groups, int* frames_per_group, int* eval_workgroups_per_frame) {
    if (!c || c->nframes <= 0) { avt_set_error("avt_launch_shape: no frames resident"); return 1; }
    const int use_moments_keep = c->fb.use_moments;
    c->fb.use_moments = choose_moments(c, c->nframes);      // the shape of the form the next optimize() will choose, whatever ran last
    const int ng = plan_groups(c, c->nframes), nfg = (c->nframes + ng - 1) / ng;
    c->fb.use_moments = use_moments_keep;
    if (groups) *groups = ng;
    if (frames_per_group) *frames_per_group = nfg;
    if (eval_workgroups_per_frame) *eval_workgroups_per_frame = choose_G(nfg, ng, c->tun);
    return 0;
}

int avt_profile_begin(avt_ctx* c) {
    if (!c) { avt_set_error("avt_profile_begin: null context"); return 1; }
    c->profiling = true;
    c->prof_events.clear();
    c->event_pool_used = 0;
    return 0;
}

int avt_profile_select(avt_ctx* c, unsigned mask) {
    if (!c) { avt_set_error("avt_profile_select: null context"); return 1; }
    c->prof_mask = mask;
    return 0;
}

int avt_profile_end(avt_ctx* c, avt_profile* out) {
    if (!c || !out) { avt_set_error("avt_profile_end: null argument"); return 1; }
    HIP_OK(hipSetDevice(c->device));
    HIP_OK(hipStreamSynchronize(c->stream));
    std::memset(out, 0, sizeof(*out));
    for (auto& rec : c->prof_events) {
        float ms = 0.f;
        HIP_OK(hipEventElapsedTime(&ms, rec.second.first, rec.second.second));
        out->ms[rec.first] += ms;
        out->launches[rec.first] += 1;
    }
    c->profiling = false;
    c->prof_events.clear();
    c->event_pool_used = 0;
    return 0;
}

}  // extern "C"
