// avt_shard.cpp — batch split of independent frames over the GPUs of one node behind the C ABI of include/avt_shard.h
// (SURVEY.md §8e): frame f -> rank f mod W, model constants replicated, RCCL over xGMI for exactly three exchanges
// (model broadcast, cloud scatter, result all-gather), all on device buffers.  The reference has no counterpart: it is
// a single process whose parallelism are per-call std::thread pools (AvatarOptimizer.cpp:337-343, :883-889).
// Compiled with hipcc (host only).  librccl is opened with dlopen at the first use, see rccl_api().
#include <dlfcn.h>
#include <link.h>
#include <rccl/rccl.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/avt_shard.h"
#include "avt_internal.h"

int avt_internal_install_frames(avt_ctx* c, int nframes, const int* counts, const double* data, const int* labels, int device_src);
void launch_pack_results(avt_ctx* c, int nframes, double* out, int stride);

#define HIP_OK(expr)                                                                          \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            avt_set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                 \
            return 1;                                                                         \
        }                                                                                     \
    } while (0)

// ---------------------------------------------------------------------------------------------------------------------
// partition
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int avt_shard_owner(int frame, int world) { return world > 0 ? frame % world : 0; }
extern "C" int avt_shard_local_count(int num_frames, int rank, int world) {
    if (world <= 0 || rank < 0 || rank >= world || num_frames <= rank) return 0;
    return (num_frames - rank + world - 1) / world;
}
extern "C" int avt_shard_local_index(int frame, int world) { return world > 0 ? frame / world : frame; }
extern "C" int avt_shard_global_frame(int local_index, int rank, int world) { return rank + world * local_index; }

// ---------------------------------------------------------------------------------------------------------------------
// packed model: header + the arrays of avt_model_desc, each 8-byte aligned, in declaration order
// ---------------------------------------------------------------------------------------------------------------------
namespace {

struct PackHeader {
    char magic[8];                 // "AVTMODEL"
    std::int32_t version, V, J, K, F, nnz_w, nnz_r, ncomps, ndims, pad;
    std::uint64_t bytes;           // of the whole block
};

struct PackLayout {
    size_t off[15], len[15];       // byte offset / byte length of every array (13, 14: the legacy format's joint shape regressor)
    size_t total;
};
enum { PACK_LIMIT_ONE_JOINT = 1, PACK_HAS_JSR = 2 };      // PackHeader::pad carries these flags (0 in blocks packed before they existed)

size_t align8(size_t x) { return (x + 7) & ~(size_t)7; }

bool pack_layout(const PackHeader& h, PackLayout& L) {
    if (h.V <= 0 || h.J <= 0 || h.K < 0 || h.F < 0 || h.nnz_w < 0 || h.nnz_r < 0 || h.ncomps < 0 || h.ndims < 0) return false;
    if (h.J > AVT_MAX_JOINTS || h.K > AVT_MAX_SHAPE || h.V > (1 << 24) || h.F > (1 << 25) || h.ncomps > 4096 || h.ndims > 4096) return false;
    const size_t V = h.V, J = h.J, K = h.K, F = h.F, C = h.ncomps, n = h.ndims;
    const size_t lens[13] = {3 * V * 8, 3 * V * K * 8, J * 4, 3 * F * 4, (V + 1) * 4, (size_t)h.nnz_w * 4, (size_t)h.nnz_w * 8,
                             (J + 1) * 4, (size_t)h.nnz_r * 4, (size_t)h.nnz_r * 8, C * 8, C * n * 8, C * n * n * 8};
    size_t o = align8(sizeof(PackHeader));
    for (int i = 0; i < 13; ++i) { L.off[i] = o; L.len[i] = lens[i]; o = align8(o + lens[i]); }
    const bool jsr = (h.pad & PACK_HAS_JSR) != 0;
    const size_t extra[2] = {jsr ? 3 * J * 8 : 0, jsr ? 3 * J * K * 8 : 0};
    for (int i = 0; i < 2; ++i) { L.off[13 + i] = o; L.len[13 + i] = extra[i]; o = align8(o + extra[i]); }
    L.total = o;
    return (h.pad & ~(PACK_LIMIT_ONE_JOINT | PACK_HAS_JSR)) == 0;
}

bool header_of(const avt_model_desc* d, PackHeader& h) {
    if (!d || d->num_points <= 0 || d->num_joints <= 0 || !d->weights_colptr || !d->jreg_colptr) return false;
    std::memset(&h, 0, sizeof h);
    std::memcpy(h.magic, "AVTMODEL", 8);
    h.version = 1;
    h.V = d->num_points; h.J = d->num_joints; h.K = d->num_shape_keys; h.F = d->num_faces;
    h.nnz_w = d->weights_colptr[d->num_points];
    h.nnz_r = d->jreg_colptr[d->num_joints];
    h.ncomps = d->prior_ncomps > 0 ? d->prior_ncomps : 0;
    h.ndims = h.ncomps ? d->prior_ndims : 0;
    h.pad = (d->limit_one_joint_per_point ? PACK_LIMIT_ONE_JOINT : 0) | ((d->joint_shape_reg_base && d->joint_shape_reg) ? PACK_HAS_JSR : 0);
    return true;
}

}  // namespace

extern "C" int avt_model_pack_size(const avt_model_desc* desc, size_t* bytes) {
    PackHeader h; PackLayout L;
    if (!bytes || !header_of(desc, h) || !pack_layout(h, L)) { avt_set_error("avt_model_pack_size: bad model description"); return 1; }
    *bytes = L.total;
    return 0;
}

extern "C" int avt_model_pack(const avt_model_desc* d, void* buf, size_t bytes) {
    PackHeader h; PackLayout L;
    if (!buf || !header_of(d, h) || !pack_layout(h, L)) { avt_set_error("avt_model_pack: bad model description"); return 1; }
    if (bytes < L.total) { avt_set_error("avt_model_pack: buffer too small"); return 1; }
    h.bytes = L.total;
    char* b = (char*)buf;
    std::memset(b, 0, L.total);
    std::memcpy(b, &h, sizeof h);
    const void* src[15] = {d->base_cloud, d->key_clouds, d->parent, d->mesh, d->weights_colptr, d->weights_row, d->weights_val,
                           d->jreg_colptr, d->jreg_row, d->jreg_val, d->prior_weight, d->prior_mean, d->prior_cov,
                           d->joint_shape_reg_base, d->joint_shape_reg};
    for (int i = 0; i < 15; ++i) {
        if (L.len[i] == 0) continue;
        if (!src[i]) { avt_set_error("avt_model_pack: a required array of the model description is NULL"); return 1; }
        std::memcpy(b + L.off[i], src[i], L.len[i]);
    }
    return 0;
}

extern "C" int avt_model_unpack(const void* buf, size_t bytes, avt_model** out) {
    if (!buf || !out || bytes < sizeof(PackHeader)) { avt_set_error("avt_model_unpack: block too small"); return 1; }
    PackHeader h;
    std::memcpy(&h, buf, sizeof h);
    PackLayout L;
    if (std::memcmp(h.magic, "AVTMODEL", 8) != 0 || h.version != 1 || !pack_layout(h, L) || h.bytes != L.total || bytes < L.total) {
        avt_set_error("avt_model_unpack: not a packed model (magic / version / size mismatch)");
        return 1;
    }
    const char* b = (const char*)buf;
    avt_model_desc d;
    std::memset(&d, 0, sizeof d);
    d.num_points = h.V; d.num_joints = h.J; d.num_shape_keys = h.K; d.num_faces = h.F;
    d.base_cloud = (const double*)(b + L.off[0]); d.key_clouds = (const double*)(b + L.off[1]);
    d.parent = (const int*)(b + L.off[2]); d.mesh = (const int*)(b + L.off[3]);
    d.weights_colptr = (const int*)(b + L.off[4]); d.weights_row = (const int*)(b + L.off[5]); d.weights_val = (const double*)(b + L.off[6]);
    d.jreg_colptr = (const int*)(b + L.off[7]); d.jreg_row = (const int*)(b + L.off[8]); d.jreg_val = (const double*)(b + L.off[9]);
    d.prior_ncomps = h.ncomps; d.prior_ndims = h.ndims;
    if (h.ncomps) { d.prior_weight = (const double*)(b + L.off[10]); d.prior_mean = (const double*)(b + L.off[11]); d.prior_cov = (const double*)(b + L.off[12]); }
    d.limit_one_joint_per_point = (h.pad & PACK_LIMIT_ONE_JOINT) ? 1 : 0;
    if (h.pad & PACK_HAS_JSR) { d.joint_shape_reg_base = (const double*)(b + L.off[13]); d.joint_shape_reg = (const double*)(b + L.off[14]); }
    // the column pointers index the arrays that follow them: check before avt_model_create walks them
    if (d.weights_colptr[0] != 0 || d.weights_colptr[h.V] != h.nnz_w || d.jreg_colptr[0] != 0 || d.jreg_colptr[h.J] != h.nnz_r) {
        avt_set_error("avt_model_unpack: sparse column pointers do not match the block");
        return 1;
    }
    for (int v = 0; v < h.V; ++v)
        if (d.weights_colptr[v + 1] < d.weights_colptr[v]) { avt_set_error("avt_model_unpack: weight column pointers not monotone"); return 1; }
    for (int j = 0; j < h.J; ++j) {
        if (d.jreg_colptr[j + 1] < d.jreg_colptr[j]) { avt_set_error("avt_model_unpack: regressor column pointers not monotone"); return 1; }
        for (int e = d.jreg_colptr[j]; e < d.jreg_colptr[j + 1]; ++e)
            if (d.jreg_row[e] < 0 || d.jreg_row[e] >= h.V) { avt_set_error("avt_model_unpack: regressor row out of range"); return 1; }
    }
    return avt_model_create(&d, out);
}

// ---------------------------------------------------------------------------------------------------------------------
// RCCL, opened at run time
// ---------------------------------------------------------------------------------------------------------------------
namespace {

struct RcclApi {
    void* handle = nullptr;
    std::string path, error;
    int version = 0;
    ncclResult_t (*GetVersion)(int*) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
};

int find_loaded_rccl(struct dl_phdr_info* info, size_t, void* data) {
    if (info->dlpi_name && std::strstr(info->dlpi_name, "librccl")) { *(std::string*)data = info->dlpi_name; return 1; }
    return 0;
}

RcclApi* rccl_api() {
    static RcclApi api;
    static bool tried = false;
    if (tried) {
        if (!api.handle) avt_set_error("avt_shard: cannot open librccl (" + api.error + "); set AVT_RCCL_LIB");
        return api.handle ? &api : nullptr;
    }
    tried = true;
    // one RCCL per process: if the host program (PyTorch) already mapped a librccl, use that very copy
    std::string loaded;
    dl_iterate_phdr(find_loaded_rccl, &loaded);
    std::vector<std::string> cand;
    if (const char* e = getenv("AVT_RCCL_LIB")) cand.push_back(e);
    if (!loaded.empty()) cand.push_back(loaded);
    cand.push_back("librccl.so.1"); cand.push_back("librccl.so"); cand.push_back("/opt/rocm/lib/librccl.so.1");
    for (const std::string& p : cand) {
        api.handle = dlopen(p.c_str(), RTLD_NOW | RTLD_GLOBAL);
        if (api.handle) { api.path = p; break; }
        const char* e = dlerror();              // dlerror() clears the message: read it once
        api.error = p + ": " + (e ? e : "dlopen failed");
    }
    if (!api.handle) { avt_set_error("avt_shard: cannot open librccl (" + api.error + "); set AVT_RCCL_LIB"); return nullptr; }
    bool ok = true;
    auto sym = [&](const char* n) { void* s = dlsym(api.handle, n); if (!s) { ok = false; api.error = std::string("missing symbol ") + n; } return s; };
    api.GetVersion = (decltype(api.GetVersion))sym("ncclGetVersion");
    api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
    api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    api.Broadcast = (decltype(api.Broadcast))sym("ncclBroadcast");
    api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
    api.Send = (decltype(api.Send))sym("ncclSend");
    api.Recv = (decltype(api.Recv))sym("ncclRecv");
    api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
    if (!ok) { avt_set_error("avt_shard: " + api.path + ": " + api.error); dlclose(api.handle); api.handle = nullptr; return nullptr; }
    Dl_info di;
    if (dladdr((void*)api.GetUniqueId, &di) && di.dli_fname) api.path = di.dli_fname;
    api.GetVersion(&api.version);
    return &api;
}

// ---------------------------------------------------------------------------------------------------------------------
// Loop-back transport: the same five collectives between the THREADS of one process (one avt_shard per thread, any mix of
// devices), as device-to-device copies behind host-side rendezvous.  It exists for two reasons: a single-process host that
// drives several GPUs from threads needs no RCCL, and every line of the multi-peer exchange code below can run with W > 1
// ranks on a one-GPU box (tests/test_gpu_shard.py) - RCCL refuses two ranks on one GPU.  Semantics are stricter than RCCL's
// (every call completes before it returns); a rank that never arrives makes its peers fail after a time-out instead of
// hanging (ncclSystemError), and the group stays broken.
// ---------------------------------------------------------------------------------------------------------------------
struct LoopSend { int dst; const void* ptr; size_t bytes; };
struct LoopMail { const void* ptr; size_t bytes; int state; };   // 0 posted, 1 taken by the receiver, 2 copied
struct LoopGroup {
    std::string name;
    int world = 0, joined = 0, refs = 0;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    unsigned long long generation = 0;
    bool broken = false;
    double timeout_s = 20.0;
    std::vector<const void*> slot;                 // one published pointer per rank (broadcast root / all-gather blocks)
    std::vector<std::deque<LoopMail>> mail;        // [source * world + destination] posted sends, oldest first
    std::vector<int> outstanding;                  // per source rank: posted sends not yet copied out by their receivers
    std::vector<char> present;                     // per rank: a member with this rank exists
};
struct LoopComm { LoopGroup* g; int rank; };
struct LoopPending { int depth = 0; std::vector<LoopSend> sends; struct R { int src; void* ptr; size_t bytes; }; std::vector<R> recvs; ncclComm_t comm = nullptr; hipStream_t stream = nullptr; };
thread_local LoopPending loop_pending;

std::mutex loop_registry_mutex;
std::map<std::string, LoopGroup*> loop_registry;

size_t nccl_type_bytes(ncclDataType_t t) {
    switch (t) {
        case ncclChar: case ncclUint8: return 1;
        case ncclInt32: case ncclUint32: case ncclFloat: return 4;
        case ncclInt64: case ncclUint64: case ncclDouble: return 8;
        default: return 0;
    }
}

// all ranks of the group meet here; false = somebody did not come (the group is broken from then on)
bool loop_barrier(LoopGroup* g) {
    std::unique_lock<std::mutex> lk(g->m);
    if (g->broken) return false;
    const unsigned long long gen = g->generation;
    if (++g->arrived == g->world) { g->arrived = 0; ++g->generation; g->cv.notify_all(); return true; }
    const bool ok = g->cv.wait_for(lk, std::chrono::duration<double>(g->timeout_s), [&] { return g->generation != gen || g->broken; });
    if (!ok || g->broken) { g->broken = true; g->cv.notify_all(); return false; }
    return true;
}

ncclResult_t loop_copy(void* dst, const void* src, size_t bytes, hipStream_t st) {
    if (bytes && dst != src && hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, st) != hipSuccess) return ncclUnhandledCudaError;
    return hipStreamSynchronize(st) == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;
}

ncclResult_t loop_GetVersion(int* v) { *v = 0; return ncclSuccess; }
const char* loop_GetErrorString(ncclResult_t r) {
    return r == ncclSuccess ? "no error" : r == ncclSystemError ? "loop-back transport: a rank did not arrive (time-out); the group is broken" : "loop-back transport: device copy failed";
}
ncclResult_t loop_CommDestroy(ncclComm_t c) {
    LoopComm* lc = (LoopComm*)c;
    std::lock_guard<std::mutex> lk(loop_registry_mutex);
    LoopGroup* g = lc->g;
    --g->joined;
    if (lc->rank >= 0 && lc->rank < (int)g->present.size()) g->present[lc->rank] = 0;
    if (--g->refs == 0) {
        auto it = loop_registry.find(g->name);
        if (it != loop_registry.end() && it->second == g) loop_registry.erase(it);      // (a broken group was already taken out of the registry)
        delete g;
    }
    delete lc;
    return ncclSuccess;
}
ncclResult_t loop_Broadcast(const void* send, void* recv, size_t count, ncclDataType_t t, int root, ncclComm_t c, hipStream_t st) {
    LoopComm* lc = (LoopComm*)c; LoopGroup* g = lc->g;
    if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;       // my buffer is ready
    if (lc->rank == root) g->slot[root] = send;
    if (!loop_barrier(g)) return ncclSystemError;
    const ncclResult_t r = loop_copy(recv, g->slot[root], count * nccl_type_bytes(t), st);
    if (!loop_barrier(g)) return ncclSystemError;                                    // the root may reuse its buffer
    return r;
}
ncclResult_t loop_AllGather(const void* send, void* recv, size_t count, ncclDataType_t t, ncclComm_t c, hipStream_t st) {
    LoopComm* lc = (LoopComm*)c; LoopGroup* g = lc->g;
    const size_t bytes = count * nccl_type_bytes(t);
    if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
    g->slot[lc->rank] = send;
    if (!loop_barrier(g)) return ncclSystemError;
    ncclResult_t r = ncclSuccess;
    for (int k = 0; k < g->world && r == ncclSuccess; ++k) r = loop_copy((char*)recv + (size_t)k * bytes, g->slot[k], bytes, st);
    if (!loop_barrier(g)) return ncclSystemError;
    return r;
}
// The end of a group (or a lone send / receive).  Point-to-point like RCCL's: only the two ends of a transfer meet - a rank with
// an empty group meets nobody - through a mailbox per (source, destination) pair: the sender posts (pointer, size), the receiver
// takes the oldest posted entry of its pair, copies device-to-device on its own stream and marks it done; the sender returns
// when all its entries are done (its buffers are free again).  All posts happen before any wait, so groups cannot deadlock.
ncclResult_t loop_flush() {
    LoopPending& p = loop_pending;
    if (!p.comm) return ncclSuccess;
    LoopComm* lc = (LoopComm*)p.comm; LoopGroup* g = lc->g;
    const int me = lc->rank, W = g->world;
    ncclResult_t r = hipStreamSynchronize(p.stream) == hipSuccess ? ncclSuccess : ncclUnhandledCudaError;   // what I send is ready
    const auto limit = std::chrono::duration<double>(g->timeout_s);
    {
        std::lock_guard<std::mutex> lk(g->m);
        for (const LoopSend& sd : p.sends) { g->mail[(size_t)me * W + sd.dst].push_back({sd.ptr, sd.bytes, 0}); ++g->outstanding[me]; }
        g->cv.notify_all();
    }
    for (const auto& rc : p.recvs) {
        if (r != ncclSuccess) break;
        std::deque<LoopMail>& box = g->mail[(size_t)rc.src * W + me];
        LoopMail* m = nullptr;
        {
            std::unique_lock<std::mutex> lk(g->m);
            const bool ok = g->cv.wait_for(lk, limit, [&] {
                if (g->broken) return true;
                for (LoopMail& e : box) if (e.state == 0) { m = &e; return true; }
                return false;
            });
            if (!ok || g->broken || !m) { g->broken = true; g->cv.notify_all(); r = ncclSystemError; break; }
            if (m->bytes != rc.bytes) { g->broken = true; g->cv.notify_all(); r = ncclInvalidUsage; break; }
            m->state = 1;                                   // taken (deque references stay valid: only the front is ever popped)
        }
        const ncclResult_t cr = loop_copy(rc.ptr, m->ptr, rc.bytes, p.stream);
        {
            std::lock_guard<std::mutex> lk(g->m);
            m->state = 2;
            while (!box.empty() && box.front().state == 2) box.pop_front();
            --g->outstanding[rc.src];
            g->cv.notify_all();
        }
        if (cr != ncclSuccess) r = cr;
    }
    {
        std::unique_lock<std::mutex> lk(g->m);
        const bool ok = g->cv.wait_for(lk, limit, [&] { return g->broken || g->outstanding[me] == 0; });
        if ((!ok || g->broken) && r == ncclSuccess) { g->broken = true; g->cv.notify_all(); r = ncclSystemError; }
    }
    p.sends.clear(); p.recvs.clear(); p.comm = nullptr; p.stream = nullptr;
    return r;
}
ncclResult_t loop_GroupStart() { ++loop_pending.depth; return ncclSuccess; }
ncclResult_t loop_GroupEnd() { return --loop_pending.depth == 0 ? loop_flush() : ncclSuccess; }
ncclResult_t loop_Send(const void* ptr, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st) {
    loop_pending.comm = c; loop_pending.stream = st;
    loop_pending.sends.push_back({peer, ptr, count * nccl_type_bytes(t)});
    return loop_pending.depth == 0 ? loop_flush() : ncclSuccess;
}
ncclResult_t loop_Recv(void* ptr, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st) {
    loop_pending.comm = c; loop_pending.stream = st;
    loop_pending.recvs.push_back({peer, ptr, count * nccl_type_bytes(t)});
    return loop_pending.depth == 0 ? loop_flush() : ncclSuccess;
}

RcclApi* loop_api() {
    static RcclApi api = [] {
        RcclApi a;
        a.path = "in-process loop-back";
        a.GetVersion = loop_GetVersion; a.CommDestroy = loop_CommDestroy; a.GetErrorString = loop_GetErrorString;
        a.Broadcast = loop_Broadcast; a.AllGather = loop_AllGather; a.Send = loop_Send; a.Recv = loop_Recv;
        a.GroupStart = loop_GroupStart; a.GroupEnd = loop_GroupEnd;
        return a;
    }();
    return &api;
}

// ---------------------------------------------------------------------------------------------------------------------
// Shared-memory transport: the same collectives between PROCESSES of one node, staged through a POSIX shared-memory
// segment (device -> segment -> device).  RCCL refuses two ranks on one GPU ("Duplicate GPU detected"), so on a one-GPU box
// the launch path a real node takes - bench.py --gpus N -> torch.distributed.run -> one process per rank -> avt_shard ->
// scatter -> optimize -> all-gather - could never run with N > 1; with this transport it does (tests/test_gpu_bench_ranks.py,
// AVT_BENCH_SHARE_GPU0=1), and a node without RCCL still has a batch split.  It is NOT the fast path: every byte crosses
// the host twice.
// Layout: a header, then one mailbox per ordered pair (source, destination): two sequence counters and SHM_CHUNK bytes.
// A message travels as chunks (a chunk never spans messages); chunk i of a box is written when chunk i - 1 has been taken
// (seq_empty == i) and published by seq_full = i + 1.  Everything is point-to-point; broadcast and all-gather are made of it.
// One progress loop serves all pending sends and receives of a call, so grouped exchanges cannot deadlock; a peer that
// never arrives makes the call fail after AVT_SHARD_LOOPBACK_TIMEOUT_S (default 60 s here) and marks the segment broken.
// ---------------------------------------------------------------------------------------------------------------------
constexpr size_t SHM_CHUNK = 1u << 20;
constexpr unsigned SHM_MAGIC = 0x41565453u;      // "AVTS"
struct ShmHeader {
    std::atomic<unsigned> magic;
    std::atomic<int> attached, broken;
    int world;
    unsigned long long chunk;
    char pad[40];
};
struct ShmBox {
    std::atomic<unsigned long long> seq_full, seq_empty;      // chunks published by the source / taken by the destination
    unsigned long long bytes;                                  // payload of the chunk in flight
    char pad[40];
};
static_assert(sizeof(ShmHeader) == 64 && sizeof(ShmBox) == 64, "shared-memory transport: 64-byte records");
struct ShmComm {
    void* base = nullptr; size_t map_bytes = 0;
    int rank = 0, world = 1;
    double timeout_s = 60.0;
    std::vector<unsigned long long> sent, taken;              // per peer: chunks I published to it / took from it
    ShmHeader* hdr() const { return (ShmHeader*)base; }
    ShmBox* box(int src, int dst) const { return (ShmBox*)((char*)base + sizeof(ShmHeader)) + (size_t)src * world + dst; }
    char* payload(int src, int dst) const { return (char*)base + sizeof(ShmHeader) + sizeof(ShmBox) * (size_t)world * world + ((size_t)src * world + dst) * SHM_CHUNK; }
};
size_t shm_bytes(int world) { return sizeof(ShmHeader) + (sizeof(ShmBox) + SHM_CHUNK) * (size_t)world * world; }
struct ShmXfer { int peer; char* ptr; size_t bytes, done; };
struct ShmPending { int depth = 0; std::vector<ShmXfer> sends, recvs; ncclComm_t comm = nullptr; hipStream_t stream = nullptr; };
thread_local ShmPending shm_pending;

// all of `sends` and `recvs` (device pointers), FIFO per peer; returns when everything I send has been copied into the
// segment and everything I receive has arrived in my buffers
ncclResult_t shm_progress(ShmComm* sc, std::vector<ShmXfer>& sends, std::vector<ShmXfer>& recvs, hipStream_t st) {
    if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;      // what I send is ready
    if (sc->hdr()->broken.load(std::memory_order_acquire)) return ncclSystemError;
    auto t0 = std::chrono::steady_clock::now();
    size_t left = 0;
    for (auto& x : sends) left += x.bytes == 0 ? 0 : 1;
    for (auto& x : recvs) left += x.bytes == 0 ? 0 : 1;
    std::vector<char> busy_send(sc->world), busy_recv(sc->world);
    while (left) {
        bool moved = false;
        std::fill(busy_send.begin(), busy_send.end(), 0); std::fill(busy_recv.begin(), busy_recv.end(), 0);
        for (auto& x : sends) {
            if (x.done == x.bytes) continue;
            if (busy_send[x.peer]) continue;                     // an earlier message to this peer goes first
            busy_send[x.peer] = 1;
            ShmBox* b = sc->box(sc->rank, x.peer);
            if (b->seq_empty.load(std::memory_order_acquire) != sc->sent[x.peer]) continue;      // the previous chunk is still in the box
            const size_t n = std::min(SHM_CHUNK, x.bytes - x.done);
            if (hipMemcpy(sc->payload(sc->rank, x.peer), x.ptr + x.done, n, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
            b->bytes = n;
            b->seq_full.store(++sc->sent[x.peer], std::memory_order_release);
            x.done += n; moved = true;
            if (x.done == x.bytes) --left;
        }
        for (auto& x : recvs) {
            if (x.done == x.bytes) continue;
            if (busy_recv[x.peer]) continue;
            busy_recv[x.peer] = 1;
            ShmBox* b = sc->box(x.peer, sc->rank);
            if (b->seq_full.load(std::memory_order_acquire) != sc->taken[x.peer] + 1) continue;   // nothing new from this peer
            const size_t n = (size_t)b->bytes;
            if (n != std::min(SHM_CHUNK, x.bytes - x.done)) { sc->hdr()->broken.store(1, std::memory_order_release); return ncclInvalidUsage; }   // the two ends disagree about a message size
            if (hipMemcpy(x.ptr + x.done, sc->payload(x.peer, sc->rank), n, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
            b->seq_empty.store(++sc->taken[x.peer], std::memory_order_release);
            x.done += n; moved = true;
            if (x.done == x.bytes) --left;
        }
        if (!left) break;
        if (moved) t0 = std::chrono::steady_clock::now();      //  the time-out bounds INACTIVITY, not the length of a long exchange
        else {
            if (sc->hdr()->broken.load(std::memory_order_acquire)) return ncclSystemError;
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > sc->timeout_s) {
                sc->hdr()->broken.store(1, std::memory_order_release);
                return ncclSystemError;
            }
            std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
    }
    return ncclSuccess;
}
ncclResult_t shm_GetVersion(int* v) { *v = 0; return ncclSuccess; }
const char* shm_GetErrorString(ncclResult_t r) {
    return r == ncclSuccess ? "no error" : r == ncclSystemError ? "shared-memory transport: a rank did not arrive (time-out); the segment is broken"
         : r == ncclInvalidUsage ? "shared-memory transport: the two ends of a transfer disagree about its size" : "shared-memory transport: device copy failed";
}
ncclResult_t shm_CommDestroy(ncclComm_t c) {
    ShmComm* sc = (ShmComm*)c;
    if (sc->base) munmap(sc->base, sc->map_bytes);
    delete sc;
    return ncclSuccess;
}
ncclResult_t shm_Broadcast(const void* send, void* recv, size_t count, ncclDataType_t t, int root, ncclComm_t c, hipStream_t st) {
    ShmComm* sc = (ShmComm*)c;
    const size_t bytes = count * nccl_type_bytes(t);
    std::vector<ShmXfer> sends, recvs;
    if (sc->rank == root) { for (int k = 0; k < sc->world; ++k) if (k != root) sends.push_back({k, (char*)send, bytes, 0}); }
    else recvs.push_back({root, (char*)recv, bytes, 0});
    const ncclResult_t r = shm_progress(sc, sends, recvs, st);
    if (r == ncclSuccess && sc->rank == root && recv != send && bytes && hipMemcpy(recv, send, bytes, hipMemcpyDeviceToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return r;
}
ncclResult_t shm_AllGather(const void* send, void* recv, size_t count, ncclDataType_t t, ncclComm_t c, hipStream_t st) {
    ShmComm* sc = (ShmComm*)c;
    const size_t bytes = count * nccl_type_bytes(t);
    std::vector<ShmXfer> sends, recvs;
    for (int k = 0; k < sc->world; ++k) if (k != sc->rank) { sends.push_back({k, (char*)send, bytes, 0}); recvs.push_back({k, (char*)recv + (size_t)k * bytes, bytes, 0}); }
    const ncclResult_t r = shm_progress(sc, sends, recvs, st);
    char* mine = (char*)recv + (size_t)sc->rank * bytes;
    if (r == ncclSuccess && mine != (const char*)send && bytes && hipMemcpy(mine, send, bytes, hipMemcpyDeviceToDevice) != hipSuccess) return ncclUnhandledCudaError;
    return r;
}
ncclResult_t shm_flush() {
    ShmPending& p = shm_pending;
    if (!p.comm) return ncclSuccess;
    const ncclResult_t r = shm_progress((ShmComm*)p.comm, p.sends, p.recvs, p.stream);
    p.sends.clear(); p.recvs.clear(); p.comm = nullptr; p.stream = nullptr;
    return r;
}
ncclResult_t shm_GroupStart() { ++shm_pending.depth; return ncclSuccess; }
ncclResult_t shm_GroupEnd() { return --shm_pending.depth == 0 ? shm_flush() : ncclSuccess; }
ncclResult_t shm_Send(const void* ptr, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st) {
    shm_pending.comm = c; shm_pending.stream = st;
    shm_pending.sends.push_back({peer, (char*)ptr, count * nccl_type_bytes(t), 0});
    return shm_pending.depth == 0 ? shm_flush() : ncclSuccess;
}
ncclResult_t shm_Recv(void* ptr, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st) {
    shm_pending.comm = c; shm_pending.stream = st;
    shm_pending.recvs.push_back({peer, (char*)ptr, count * nccl_type_bytes(t), 0});
    return shm_pending.depth == 0 ? shm_flush() : ncclSuccess;
}
RcclApi* shm_api() {
    static RcclApi api = [] {
        RcclApi a;
        a.path = "shared-memory segment";
        a.GetVersion = shm_GetVersion; a.CommDestroy = shm_CommDestroy; a.GetErrorString = shm_GetErrorString;
        a.Broadcast = shm_Broadcast; a.AllGather = shm_AllGather; a.Send = shm_Send; a.Recv = shm_Recv;
        a.GroupStart = shm_GroupStart; a.GroupEnd = shm_GroupEnd;
        return a;
    }();
    return &api;
}

}  // namespace

struct avt_shard {
    RcclApi* api = nullptr;
    ncclComm_t comm = nullptr;
    int device = 0, rank = 0, world = 1;
    hipStream_t stream = nullptr;          // exchanges that happen before a context exists (model broadcast)
    std::string backend;
    // result gather: per-rank send block and the gathered block, device
    double* d_send = nullptr; double* d_recv = nullptr; size_t gather_cap = 0;   // doubles per rank block
    const double* gathered = nullptr;      // where the last gather left every rank's block (d_recv; one rank: the context's own result records)
    void* d_stage = nullptr; size_t stage_cap = 0;                              // scatter / broadcast staging, bytes
    void* d_stage2 = nullptr; size_t stage2_cap = 0;
    hipStream_t gather_stream = nullptr;   // the stream the last all-gather was enqueued on
    bool self_exchange = false;            // avt_shard_set_self_exchange: a rank's own blocks go through the transport as well (dry runs)
};

#define NCCL_OK(s, expr)                                                                                       \
    do {                                                                                                       \
        ncclResult_t _r = (expr);                                                                              \
        if (_r != ncclSuccess) {                                                                               \
            avt_set_error(std::string(#expr) + ": " + ((s)->api->GetErrorString ? (s)->api->GetErrorString(_r) : "rccl error")); \
            return 1;                                                                                          \
        }                                                                                                      \
    } while (0)

namespace {

int grow(void** p, size_t* cap, size_t bytes) {
    if (*cap >= bytes && *p) return 0;
    if (*p) HIP_OK(hipFree(*p));
    *p = nullptr; *cap = 0;
    HIP_OK(hipMalloc(p, std::max<size_t>(bytes, 256)));
    *cap = std::max<size_t>(bytes, 256);
    return 0;
}

int gather_stride(const avt_ctx* c) { return c->dm.d.xsize + AVT_SHARD_STAT_DOUBLES; }

}  // namespace

extern "C" int avt_shard_unique_id(char id[AVT_SHARD_ID_BYTES]) {
    RcclApi* a = rccl_api();
    if (!a) return 1;
    if (!id) { avt_set_error("avt_shard_unique_id: null argument"); return 1; }
    static_assert(AVT_SHARD_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");
    ncclUniqueId u;
    const ncclResult_t r = a->GetUniqueId(&u);
    if (r != ncclSuccess) { avt_set_error(std::string("ncclGetUniqueId: ") + a->GetErrorString(r)); return 1; }
    std::memcpy(id, u.internal, AVT_SHARD_ID_BYTES);
    return 0;
}

extern "C" int avt_shard_create(int device, int rank, int world, const char id[AVT_SHARD_ID_BYTES], avt_shard** out) {
    if (!id || !out || world <= 0 || rank < 0 || rank >= world) { avt_set_error("avt_shard_create: bad argument"); return 1; }
    RcclApi* a = rccl_api();
    if (!a) return 1;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { avt_set_error("avt_shard_create: device index out of range"); return 2; }
    HIP_OK(hipSetDevice(device));
    avt_shard* s = new avt_shard();
    s->api = a; s->device = device; s->rank = rank; s->world = world;
    ncclUniqueId u;
    std::memcpy(u.internal, id, AVT_SHARD_ID_BYTES);
    const ncclResult_t r = a->CommInitRank(&s->comm, world, u, rank);
    if (r != ncclSuccess) {
        avt_set_error(std::string("ncclCommInitRank: ") + a->GetErrorString(r));
        delete s;
        return 1;
    }
    if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) {
        avt_set_error("avt_shard_create: hipStreamCreate failed");
        a->CommDestroy(s->comm);
        delete s;
        return 1;
    }
    char buf[512];
    snprintf(buf, sizeof buf, "rccl %d.%d.%d, %s", a->version / 10000, (a->version / 100) % 100, a->version % 100, a->path.c_str());
    s->backend = buf;
    *out = s;
    return 0;
}

extern "C" int avt_shard_create_loopback(int device, int rank, int world, const char* group, avt_shard** out) {
    if (!group || !out || world <= 0 || rank < 0 || rank >= world) { avt_set_error("avt_shard_create_loopback: bad argument"); return 1; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { avt_set_error("avt_shard_create_loopback: device index out of range"); return 2; }
    HIP_OK(hipSetDevice(device));
    LoopGroup* g = nullptr;
    {
        std::lock_guard<std::mutex> lk(loop_registry_mutex);
        auto it = loop_registry.find(group);
        if (it != loop_registry.end() && it->second->broken) {      // a group that timed out stays broken for its members; a fresh create under the same name starts clean
            loop_registry.erase(it);
            it = loop_registry.end();
        }
        if (it == loop_registry.end()) {
            g = new LoopGroup();
            g->name = group; g->world = world; g->present.assign(world, 0); g->slot.assign(world, nullptr); g->mail.resize((size_t)world * world); g->outstanding.assign(world, 0);
            if (const char* e = getenv("AVT_SHARD_LOOPBACK_TIMEOUT_S")) g->timeout_s = std::max(0.1, atof(e));
            loop_registry[group] = g;
        } else {
            g = it->second;
            if (g->world != world || g->joined >= world) { avt_set_error("avt_shard_create_loopback: group exists with another world size, or is full"); return 1; }
            if (g->present[rank]) { avt_set_error("avt_shard_create_loopback: this rank is already a member of the group"); return 1; }
        }
        ++g->joined; ++g->refs; g->present[rank] = 1;
    }
    avt_shard* s = new avt_shard();
    s->api = loop_api(); s->device = device; s->rank = rank; s->world = world;
    s->comm = (ncclComm_t) new LoopComm{g, rank};
    if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) {
        avt_set_error("avt_shard_create_loopback: hipStreamCreate failed");
        s->api->CommDestroy(s->comm);
        delete s;
        return 1;
    }
    s->backend = std::string("loop-back (threads of one process), group ") + group;
    *out = s;
    return 0;
}

extern "C" int avt_shard_create_shm(int device, int rank, int world, const char id[AVT_SHARD_ID_BYTES], avt_shard** out) {
    if (!id || !out || world <= 0 || world > 64 || rank < 0 || rank >= world) { avt_set_error("avt_shard_create_shm: bad argument"); return 1; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { avt_set_error("avt_shard_create_shm: device index out of range"); return 2; }
    HIP_OK(hipSetDevice(device));
    unsigned long long h = 1469598103934665603ull;      // FNV-1a of the rendezvous bytes names the segment
    for (int i = 0; i < AVT_SHARD_ID_BYTES; ++i) { h ^= (unsigned char)id[i]; h *= 1099511628211ull; }
    char name[64];
    snprintf(name, sizeof name, "/avt_shard_%016llx", h);
    double timeout_s = 60.0;
    if (const char* e = getenv("AVT_SHARD_LOOPBACK_TIMEOUT_S")) timeout_s = std::max(0.1, atof(e));
    const size_t bytes = shm_bytes(world);
    const auto t0 = std::chrono::steady_clock::now();
    auto late = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s; };
    int fd = -1;
    if (rank == 0) {
        fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) {
            avt_set_error(std::string("avt_shard_create_shm: cannot create the segment ") + name + ": " + strerror(errno));
            if (fd >= 0) { close(fd); shm_unlink(name); }
            return 1;
        }
    } else {
        struct stat sb;
        while (true) {      // rank 0 creates and sizes it
            fd = shm_open(name, O_RDWR, 0600);
            if (fd >= 0 && fstat(fd, &sb) == 0 && (size_t)sb.st_size >= bytes) break;
            if (fd >= 0) { close(fd); fd = -1; }
            if (late()) { avt_set_error(std::string("avt_shard_create_shm: rank 0 never created the segment ") + name); return 1; }
            std::this_thread::sleep_for(std::chrono::milliseconds(2));
        }
    }
    void* base = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (base == MAP_FAILED) { avt_set_error(std::string("avt_shard_create_shm: mmap: ") + strerror(errno)); if (rank == 0) shm_unlink(name); return 1; }
    ShmComm* sc = new ShmComm();
    sc->base = base; sc->map_bytes = bytes; sc->rank = rank; sc->world = world; sc->timeout_s = timeout_s;
    sc->sent.assign(world, 0); sc->taken.assign(world, 0);
    ShmHeader* hd = sc->hdr();
    if (rank == 0) {      // (a fresh segment is zero-filled: counters start at 0)
        hd->world = world; hd->chunk = SHM_CHUNK;
        hd->magic.store(SHM_MAGIC, std::memory_order_release);
    } else {
        while (hd->magic.load(std::memory_order_acquire) != SHM_MAGIC) {
            if (late()) { avt_set_error("avt_shard_create_shm: the segment was never initialised"); shm_CommDestroy((ncclComm_t)sc); return 1; }
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
        if (hd->world != world || hd->chunk != SHM_CHUNK) { avt_set_error("avt_shard_create_shm: the segment belongs to a group of another shape"); shm_CommDestroy((ncclComm_t)sc); return 1; }
    }
    hd->attached.fetch_add(1, std::memory_order_acq_rel);
    while (hd->attached.load(std::memory_order_acquire) < world) {      // creation is collective, like ncclCommInitRank
        if (late()) {
            avt_set_error("avt_shard_create_shm: not every rank attached in time");
            if (rank == 0) shm_unlink(name);
            shm_CommDestroy((ncclComm_t)sc);
            return 1;
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    if (rank == 0) shm_unlink(name);      // everybody holds a mapping: the name can go, the memory lives until the last unmap
    avt_shard* s = new avt_shard();
    s->api = shm_api(); s->device = device; s->rank = rank; s->world = world;
    s->comm = (ncclComm_t)sc;
    if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) {
        avt_set_error("avt_shard_create_shm: hipStreamCreate failed");
        s->api->CommDestroy(s->comm);
        delete s;
        return 1;
    }
    s->backend = "shared memory (processes of one node, staged through the host)";
    *out = s;
    return 0;
}

extern "C" void avt_shard_destroy(avt_shard* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    if (s->comm) s->api->CommDestroy(s->comm);
    for (void* p : {(void*)s->d_send, (void*)s->d_recv, s->d_stage, s->d_stage2})
        if (p) (void)hipFree(p);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

extern "C" int avt_shard_rank(const avt_shard* s) { return s ? s->rank : -1; }
extern "C" int avt_shard_world(const avt_shard* s) { return s ? s->world : 0; }
extern "C" const char* avt_shard_backend(const avt_shard* s) { return s ? s->backend.c_str() : ""; }

extern "C" int avt_shard_broadcast_model(avt_shard* s, int root, const avt_model_desc* desc, avt_model** out) try {
    if (!s || !out || root < 0 || root >= s->world) { avt_set_error("avt_shard_broadcast_model: bad argument"); return 1; }
    HIP_OK(hipSetDevice(s->device));
    // 1. size of the packed block (8 bytes), 2. the block itself; both as device-buffer broadcasts
    unsigned long long nbytes = 0;
    std::vector<char> host;
    if (s->rank == root) {
        size_t n = 0;
        if (avt_model_pack_size(desc, &n)) return 1;
        host.resize(n);
        if (avt_model_pack(desc, host.data(), n)) return 1;
        nbytes = n;
    }
    if (grow(&s->d_stage2, &s->stage2_cap, 64)) return 1;
    HIP_OK(hipMemcpyAsync(s->d_stage2, &nbytes, 8, hipMemcpyHostToDevice, s->stream));
    NCCL_OK(s, s->api->Broadcast(s->d_stage2, s->d_stage2, 8, ncclChar, root, s->comm, s->stream));
    HIP_OK(hipMemcpyAsync(&nbytes, s->d_stage2, 8, hipMemcpyDeviceToHost, s->stream));
    HIP_OK(hipStreamSynchronize(s->stream));
    if (nbytes < sizeof(PackHeader) || nbytes > (1ull << 32)) { avt_set_error("avt_shard_broadcast_model: implausible model size received"); return 1; }
    if (grow(&s->d_stage, &s->stage_cap, nbytes)) return 1;
    if (s->rank == root) HIP_OK(hipMemcpyAsync(s->d_stage, host.data(), nbytes, hipMemcpyHostToDevice, s->stream));
    NCCL_OK(s, s->api->Broadcast(s->d_stage, s->d_stage, nbytes, ncclChar, root, s->comm, s->stream));
    host.resize(nbytes);
    HIP_OK(hipMemcpyAsync(host.data(), s->d_stage, nbytes, hipMemcpyDeviceToHost, s->stream));
    HIP_OK(hipStreamSynchronize(s->stream));
    return avt_model_unpack(host.data(), host.size(), out);
} catch (const std::exception& e) { avt_set_error(std::string("avt_shard_broadcast_model: ") + e.what()); return 1; }

extern "C" int avt_shard_scatter_frames(avt_shard* s, avt_ctx* c, int root, int B, const double* data, const int* labels,
                                        const int* offs, const double* p, const double* q, const double* w) try {
    if (!s || !c || root < 0 || root >= s->world || B <= 0) { avt_set_error("avt_shard_scatter_frames: bad argument"); return 1; }
    const bool is_root = s->rank == root;
    const AvtDims& d = c->dm.d;
    const int W = s->world, xs = d.xsize, J = d.J, K = d.K;
    const int nloc = avt_shard_local_count(B, s->rank, W);
    HIP_OK(hipSetDevice(s->device));
    hipStream_t st = c->stream;
    // A rank that leaves before a collective its peers enter blocks them for ever.  Everything that can be wrong on ONE rank
    // only - the root's arguments, a context too small for this rank's share - is therefore carried INTO the exchange and all
    // ranks leave together: the root marks a bad batch in the table it broadcasts, and every rank contributes an ok / not-ok
    // word to a one-double all-gather before any cloud moves.
    std::string my_error;
    // ---- 1. frame table and start states: one small broadcast.  block = [B point counts as doubles | B x xsize states]
    const size_t tab_n = (size_t)B * (1 + xs);
    std::vector<double> tab(tab_n, 0.0);
    if (is_root) {
        if (!data || !labels || !offs || !p || !q || !w) my_error = "avt_shard_scatter_frames: root needs all host arrays";
        for (int f = 0; f < B && my_error.empty(); ++f) {
            const int n = offs[f + 1] - offs[f];
            if (n < 0) { my_error = "avt_shard_scatter_frames: frame_offsets not monotone"; break; }
            tab[f] = (double)n;
            double* x = &tab[(size_t)B + (size_t)f * xs];
            std::copy(p + 3 * (size_t)f, p + 3 * (size_t)f + 3, x);
            std::copy(q + (size_t)4 * J * f, q + (size_t)4 * J * (f + 1), x + 3);
            std::copy(w + (size_t)K * f, w + (size_t)K * (f + 1), x + 3 + 4 * J);
        }
        if (!my_error.empty()) tab[0] = -1.0;          // a point count cannot be negative: "the root's batch is unusable"
    }
    if (grow(&s->d_stage2, &s->stage2_cap, std::max<size_t>(tab_n, (size_t)W + 1) * 8)) return 1;   // (allocation failure: nothing a peer could wait for has begun)
    if (is_root) HIP_OK(hipMemcpyAsync(s->d_stage2, tab.data(), tab_n * 8, hipMemcpyHostToDevice, st));
    NCCL_OK(s, s->api->Broadcast(s->d_stage2, s->d_stage2, tab_n, ncclDouble, root, s->comm, st));
    HIP_OK(hipMemcpyAsync(tab.data(), s->d_stage2, tab_n * 8, hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    std::vector<int> cnt(B, 0);
    std::vector<long long> rank_pts(W, 0);
    const bool root_bad = tab[0] < 0.0;
    if (root_bad && my_error.empty()) my_error = "avt_shard_scatter_frames: the root rank rejected the batch";
    if (my_error.empty() && nloc > c->fb.max_frames) my_error = "avt_shard_scatter_frames: this rank's share exceeds the context's max_frames";
    for (int f = 0; f < B && !root_bad; ++f) {
        cnt[f] = (int)tab[f];
        // every rank checks ITS frames against ITS context (contexts may differ in size)
        if (f % W == s->rank && my_error.empty() && (cnt[f] < 0 || cnt[f] > c->fb.max_points))
            my_error = "avt_shard_scatter_frames: a frame has more points than max_points_per_frame";
        rank_pts[f % W] += std::max(cnt[f], 0);
    }
    {   // agreement: one double per rank (0 = fine), everybody sees everybody's
        std::vector<double> flags((size_t)W + 1, 0.0);
        flags[W] = my_error.empty() ? 0.0 : 1.0;
        double* dflag = (double*)s->d_stage2;
        HIP_OK(hipMemcpyAsync(dflag + W, &flags[W], 8, hipMemcpyHostToDevice, st));
        NCCL_OK(s, s->api->AllGather(dflag + W, dflag, 1, ncclDouble, s->comm, st));
        HIP_OK(hipMemcpyAsync(flags.data(), dflag, (size_t)W * 8, hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        int bad_rank = -1;
        for (int r = 0; r < W; ++r) if (flags[r] != 0.0 && bad_rank < 0) bad_rank = r;
        if (bad_rank >= 0) {
            c->nframes = 0; c->frames_valid = c->state_valid = false;
            avt_set_error(!my_error.empty() ? my_error : "avt_shard_scatter_frames: rank " + std::to_string(bad_rank) + " rejected the batch; nothing was exchanged");
            return 1;
        }
    }
    // ---- 2. clouds: the root permutes the batch so that every rank's frames are contiguous (rank-major, then local
    // order), uploads it once and sends each rank its block: [3 n doubles | n ints] -> two sends per peer, one group
    std::vector<long long> rank_off(W + 1, 0);
    for (int r = 0; r < W; ++r) rank_off[r + 1] = rank_off[r] + rank_pts[r];
    const long long total = rank_off[W], mine = rank_pts[s->rank];
    double* d_data = nullptr; int* d_lab = nullptr;
    if (is_root) {
        std::vector<double> hd((size_t)total * 3);
        std::vector<int> hl((size_t)total);
        std::vector<long long> cur(rank_off.begin(), rank_off.end() - 1);
        for (int f = 0; f < B; ++f) {
            const int r = f % W, n = cnt[f];
            std::memcpy(&hd[(size_t)cur[r] * 3], data + (size_t)offs[f] * 3, (size_t)n * 24);
            std::memcpy(&hl[(size_t)cur[r]], labels + offs[f], (size_t)n * 4);
            cur[r] += n;
        }
        if (grow(&s->d_stage, &s->stage_cap, (size_t)total * 28 + 64)) return 1;
        d_data = (double*)s->d_stage; d_lab = (int*)((char*)s->d_stage + (size_t)total * 24);
        if (total) {
            HIP_OK(hipMemcpyAsync(d_data, hd.data(), (size_t)total * 24, hipMemcpyHostToDevice, st));
            HIP_OK(hipMemcpyAsync(d_lab, hl.data(), (size_t)total * 4, hipMemcpyHostToDevice, st));
        }
        HIP_OK(hipStreamSynchronize(st));   // the host vectors go out of scope below
    } else {
        if (grow(&s->d_stage, &s->stage_cap, (size_t)mine * 28 + 64)) return 1;
        d_data = (double*)s->d_stage; d_lab = (int*)((char*)s->d_stage + (size_t)mine * 24);
    }
    const bool self_loop = s->self_exchange;   // dry runs (avt_shard_set_self_exchange): push the root's own block through RCCL too
    double* my_data = d_data; int* my_lab = d_lab;
    if (is_root) { my_data = d_data + rank_off[s->rank] * 3; my_lab = d_lab + rank_off[s->rank]; }
    double* loop_data = nullptr; int* loop_lab = nullptr;
    if (is_root && self_loop && mine) {
        if (grow(&s->d_stage2, &s->stage2_cap, std::max<size_t>((size_t)mine * 28 + 64, tab_n * 8))) return 1;
        loop_data = (double*)s->d_stage2; loop_lab = (int*)((char*)s->d_stage2 + (size_t)mine * 24);
    }
    NCCL_OK(s, s->api->GroupStart());
    if (is_root) {
        for (int r = 0; r < W; ++r) {
            if (rank_pts[r] == 0 || (r == s->rank && !self_loop)) continue;
            NCCL_OK(s, s->api->Send(d_data + rank_off[r] * 3, (size_t)rank_pts[r] * 3, ncclDouble, r, s->comm, st));
            NCCL_OK(s, s->api->Send(d_lab + rank_off[r], (size_t)rank_pts[r], ncclInt32, r, s->comm, st));
        }
        if (self_loop && mine) {
            NCCL_OK(s, s->api->Recv(loop_data, (size_t)mine * 3, ncclDouble, root, s->comm, st));
            NCCL_OK(s, s->api->Recv(loop_lab, (size_t)mine, ncclInt32, root, s->comm, st));
            my_data = loop_data; my_lab = loop_lab;
        }
    } else if (mine) {
        NCCL_OK(s, s->api->Recv(d_data, (size_t)mine * 3, ncclDouble, root, s->comm, st));
        NCCL_OK(s, s->api->Recv(d_lab, (size_t)mine, ncclInt32, root, s->comm, st));
    }
    NCCL_OK(s, s->api->GroupEnd());
    // ---- 3. install this rank's frames in the context's resident buffers (device-to-device) and its start states
    std::vector<int> lcnt(nloc);
    std::vector<double> lp((size_t)nloc * 3), lq((size_t)nloc * 4 * J), lw((size_t)nloc * K);
    for (int i = 0; i < nloc; ++i) {
        const int f = avt_shard_global_frame(i, s->rank, W);
        lcnt[i] = cnt[f];
        const double* x = &tab[(size_t)B + (size_t)f * xs];
        std::copy(x, x + 3, &lp[(size_t)i * 3]);
        std::copy(x + 3, x + 3 + 4 * J, &lq[(size_t)i * 4 * J]);
        std::copy(x + 3 + 4 * J, x + xs, &lw[(size_t)i * K]);
    }
    if (nloc == 0) {       // a rank without frames (B < W): nothing resident, and a later gather must see exactly that
        c->nframes = 0; c->frames_valid = c->state_valid = false;
        HIP_OK(hipStreamSynchronize(st));
        return 0;
    }
    if (avt_internal_install_frames(c, nloc, lcnt.data(), my_data, my_lab, 1)) return 1;
    if (avt_state_upload(c, nloc, lp.data(), lq.data(), lw.data())) return 1;
    HIP_OK(hipStreamSynchronize(st));
    return 0;
} catch (const std::exception& e) { avt_set_error(std::string("avt_shard_scatter_frames: ") + e.what()); return 1; }

extern "C" int avt_shard_gather_enqueue(avt_shard* s, avt_ctx* c, int B) {
    if (!s || !c || B <= 0) { avt_set_error("avt_shard_gather_enqueue: bad argument"); return 1; }
    const int W = s->world, per = (B + W - 1) / W, stride = gather_stride(c);
    const int nloc = avt_shard_local_count(B, s->rank, W);
    // (a rank whose context does not hold its share still takes part in the all-gather - its peers are on their way into it - with
    // its rows marked faulty, and reports the error afterwards; every rank's download then fails on those rows)
    const bool mismatch = nloc != (c->frames_valid ? c->nframes : 0);
    HIP_OK(hipSetDevice(s->device));
    const size_t blk = (size_t)per * stride;
    if (s->gather_cap < blk) {
        HIP_OK(hipStreamSynchronize(c->stream));
        HIP_OK(hipStreamSynchronize(s->stream));
        if (s->d_send) HIP_OK(hipFree(s->d_send));
        if (s->d_recv) HIP_OK(hipFree(s->d_recv));
        s->d_send = s->d_recv = nullptr; s->gather_cap = 0;
        HIP_OK(hipMalloc((void**)&s->d_send, blk * 8));
        HIP_OK(hipMalloc((void**)&s->d_recv, blk * 8 * W));
        HIP_OK(hipMemsetAsync(s->d_send, 0, blk * 8, c->stream));
        s->gather_cap = blk;
    }
    // Packing kernel and all-gather go behind optimize() on the context's stream.  Measured on one MI355X (bench.py, one frame
    // per step, 0.62 ms): this costs 6 us per step; putting the all-gather on the shard's own stream instead - so that the
    // exchange of step k overlaps step k+1 - costs 25 us, because the event record / cross-stream wait the hand-over needs
    // sit in the context stream's critical path (double-buffering the send block and waiting on the host: 24 us).
    // (avt_shard_set_self_exchange - the dry-run switch that pushes a lone rank's own blocks through the transport - keeps the one-rank all-gather too)
    const bool direct = W == 1 && !mismatch && !s->self_exchange;
    if (mismatch) {
        std::vector<double> rows(blk, 0.0);
        for (int i = 0; i < per; ++i) rows[(size_t)i * stride + c->dm.d.xsize + 7] = (double)AVT_FAULT_NOT_RESIDENT;
        HIP_OK(hipMemcpyAsync(s->d_send, rows.data(), blk * 8, hipMemcpyHostToDevice, c->stream));
        HIP_OK(hipStreamSynchronize(c->stream));
    } else if (nloc && !c->results_fresh) {      // (no optimize() in front: the records are made now)
        launch_pack_results(c, nloc, c->fb.results, stride);
        c->results_fresh = true;
    }
    // The result records (p, q, w, statistics, fault word per resident frame) are written by the k_lbs launch that closes optimize()
    // (FrameBuffers::results): they ARE the send block.  One rank: they are the gathered block as well - an all-gather over one rank is a
    // device-to-device copy kernel on the stream, 4.3 us per step for the identity.
    const double* send = (mismatch || (size_t)per > (size_t)c->fb.max_frames) ? s->d_send : c->fb.results;
    if (!mismatch && send == s->d_send && nloc) HIP_OK(hipMemcpyAsync(s->d_send, c->fb.results, (size_t)nloc * stride * 8, hipMemcpyDeviceToDevice, c->stream));
    //  ... as a SNAPSHOT, like every other world size: enqueue -> optimize -> download returns the rows of the call the gather was enqueued
    // behind, not the later call's (include/avt_shard.h: "into a device buffer owned by the shard"); one rank: a plain device-to-device copy
    if (!direct) NCCL_OK(s, s->api->AllGather(send, s->d_recv, blk, ncclDouble, s->comm, c->stream));
    else if (nloc) HIP_OK(hipMemcpyAsync(s->d_recv, send, (size_t)nloc * stride * 8, hipMemcpyDeviceToDevice, c->stream));
    s->gathered = s->d_recv;
    s->gather_stream = c->stream;
    if (mismatch) { avt_set_error("avt_shard_gather_enqueue: resident frames differ from this rank's share of the batch (its rows were gathered as faulty)"); return 1; }
    return 0;
}

extern "C" int avt_shard_set_self_exchange(avt_shard* s, int on) {
    if (!s) { avt_set_error("avt_shard_set_self_exchange: null argument"); return 1; }
    s->self_exchange = on != 0;
    return 0;
}

extern "C" int avt_shard_gather_wait(avt_shard* s) {
    if (!s) { avt_set_error("avt_shard_gather_wait: null argument"); return 1; }
    HIP_OK(hipSetDevice(s->device));
    if (s->gather_stream) HIP_OK(hipStreamSynchronize(s->gather_stream));
    return 0;
}

extern "C" int avt_shard_gather_download(avt_shard* s, avt_ctx* c, int B, double* p, double* q, double* w, avt_stats* stats) try {
    if (!s || !c || B <= 0) { avt_set_error("avt_shard_gather_download: bad argument"); return 1; }
    const AvtDims& d = c->dm.d;
    const int W = s->world, per = (B + W - 1) / W, stride = gather_stride(c), xs = d.xsize, J = d.J, K = d.K;
    const size_t blk = (size_t)per * stride;
    if (s->gather_cap < blk || !s->d_recv || !s->gathered) { avt_set_error("avt_shard_gather_download: nothing was gathered"); return 1; }
    HIP_OK(hipSetDevice(s->device));
    std::vector<double> host(blk * W);
    HIP_OK(hipMemcpyAsync(host.data(), s->gathered, host.size() * 8, hipMemcpyDeviceToHost, c->stream));   // behind the all-gather
    HIP_OK(hipStreamSynchronize(c->stream));
    int bad = -1;
    for (int f = 0; f < B; ++f) {
        const double* x = &host[(size_t)(f % W) * blk + (size_t)(f / W) * stride];
        if (x[xs + 7] != 0.0 && bad < 0) bad = f;       // the owning rank's device fault bits travelled with the result (k_pack_results)
        if (p) std::copy(x, x + 3, p + 3 * (size_t)f);
        if (q) std::copy(x + 3, x + 3 + 4 * J, q + (size_t)4 * J * f);
        if (w) std::copy(x + 3 + 4 * J, x + xs, w + (size_t)K * f);
        if (stats) {
            const double* t = x + xs;
            stats[f].initial_cost = t[0]; stats[f].final_cost = t[1]; stats[f].lambda = t[2];
            stats[f].num_correspondences = (int)t[3]; stats[f].matched_model_points = (int)t[4];
            stats[f].gn_iterations = (int)t[5]; stats[f].accepted_steps = (int)t[6];
        }
    }
    if (bad >= 0) {
        (void)hipMemsetAsync(c->fb.fault, 0, (size_t)c->fb.max_frames * sizeof(unsigned), c->stream);   // reported once
        c->results_fresh = false;      //  the result records still carry the word that was just cleared: the next gather packs them again
        avt_set_error("avt_shard_gather_download: frame " + std::to_string(bad) + " carries a device fault (rank " + std::to_string(bad % W) + "); its result is not valid");
        return AVT_STATUS_DEVICE_FAULT;
    }
    return 0;
} catch (const std::exception& e) { avt_set_error(std::string("avt_shard_gather_download: ") + e.what()); return 1; }

extern "C" int avt_shard_gather_results(avt_shard* s, avt_ctx* c, int B, double* p, double* q, double* w, avt_stats* stats) {
    if (avt_shard_gather_enqueue(s, c, B)) return 1;
    return avt_shard_gather_download(s, c, B, p, q, w, stats);
}

extern "C" int avt_shard_barrier(avt_shard* s, avt_ctx* c) {
    if (!s) { avt_set_error("avt_shard_barrier: null argument"); return 1; }
    HIP_OK(hipSetDevice(s->device));
    hipStream_t st = c ? c->stream : s->stream;
    if (grow(&s->d_stage2, &s->stage2_cap, (size_t)(s->world + 1) * 8)) return 1;
    double* b = (double*)s->d_stage2;
    NCCL_OK(s, s->api->AllGather(b + s->world, b, 1, ncclDouble, s->comm, st));
    HIP_OK(hipStreamSynchronize(st));
    return 0;
}
