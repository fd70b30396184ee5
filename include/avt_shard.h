/* avt_shard.h — batch split of independent frames over the GPUs of one node, C ABI.
 *
 * The reference is single-process: its only parallelism are per-call std::thread pools (AvatarOptimizer.cpp:337-343,
 * :883-889), and independent subjects are processed by independent processes (smplsynth.cpp:67,89-90 — one Avatar per
 * thread over a shared immutable AvatarModel).  On MI355X the same independence becomes one process per GPU:
 *
 *     frame f of a batch of B  ->  rank f mod W            (avt_shard_owner / avt_shard_local_count)
 *
 * Model constants are replicated, there is NO collective inside optimize().  RCCL (over xGMI) is used for exactly three
 * exchanges, all on device buffers and all on the owning context's stream:
 *   - avt_shard_broadcast_model : ncclBroadcast of the packed model (≈2.3 MB, once per model);
 *   - avt_shard_scatter_frames  : grouped ncclSend / ncclRecv of each rank's clouds + labels (28 B per point) and an
 *                                 ncclBroadcast of the (small) frame table and start states;
 *   - avt_shard_gather_results  : ncclAllGather of (p, q, w) + stats per frame (xsize + 8 doubles).
 * librccl is opened at run time (dlopen: the copy already mapped by the host process if there is one, e.g. the one
 * PyTorch ships), so libavatar_hip.so itself has no link-time dependency on it and single-GPU users never load it.
 *
 * Rendezvous: rank 0 calls avt_shard_unique_id() and hands the 128 opaque bytes to the other ranks by whatever
 * out-of-band channel the host program has (the torch.distributed store in bench.py, a file or MPI in a C++ host);
 * every rank then calls avt_shard_create().  One shard handle per process / GPU; not thread-safe.
 */
#ifndef AVT_SHARD_H_
#define AVT_SHARD_H_

#include <stddef.h>

#include "avt.h"

#ifdef __cplusplus
extern "C" {
#endif

#define AVT_SHARD_ID_BYTES 128            /* == NCCL_UNIQUE_ID_BYTES */
#define AVT_SHARD_STAT_DOUBLES 8          /* stats of a frame as doubles, see avt_shard_gather_results */

typedef struct avt_shard avt_shard;

/* ---- the partition (pure functions; no GPU, no communicator) */
int avt_shard_owner(int frame, int world);                              /* frame mod world */
int avt_shard_local_count(int num_frames, int rank, int world);        /* frames r, r+W, ... < num_frames */
int avt_shard_local_index(int frame, int world);                       /* position of `frame` among its owner's frames */
int avt_shard_global_frame(int local_index, int rank, int world);      /* inverse: rank + world * local_index */

/* ---- packed model: every field of avt_model_desc in one relocatable byte block (what the broadcast ships) */
int avt_model_pack_size(const avt_model_desc* desc, size_t* bytes);
int avt_model_pack(const avt_model_desc* desc, void* buf, size_t bytes);
/* avt_model_create() from a packed block (the block is only read during the call) */
int avt_model_unpack(const void* buf, size_t bytes, avt_model** out);

/* ---- communicator */
int avt_shard_unique_id(char id[AVT_SHARD_ID_BYTES]);                                  /* ncclGetUniqueId */
int avt_shard_create(int device, int rank, int world, const char id[AVT_SHARD_ID_BYTES], avt_shard** out);
/* The same handle over the in-process LOOP-BACK transport instead of RCCL: the `world` ranks are threads of this process
 * (one avt_shard and one avt_ctx each, on any devices), ranks that name the same `group` string belong together, and the
 * exchanges below become device-to-device copies behind host-side rendezvous.  For single-process hosts that drive their GPUs
 * from threads, and for exercising the multi-rank exchange code on a one-GPU box (RCCL refuses two ranks on one GPU).  Every
 * exchange is collective: each rank's thread must make the call.  A rank that does not arrive within
 * AVT_SHARD_LOOPBACK_TIMEOUT_S (default 20 s) makes its peers' calls FAIL instead of hang.  A group that broke this way stays broken for its
 * members; creating the ranks again under the same name starts a fresh group, and a rank can be a member only once.
 * Note for thread-per-GPU hosts: every hipGraphLaunch of the process goes through ONE mutex inside the library (HIP 7.0's hipGraphLaunch is not
 * safe against a hipGraphLaunch from another thread, profiles/r03_hipgraphlaunch_thread_crash.txt), and a two-branch graph costs ~230 us of host
 * time to enqueue: the optimize() calls of the threads serialise on it.  One process per GPU (the RCCL transport) does not share that lock. */
int avt_shard_create_loopback(int device, int rank, int world, const char* group, avt_shard** out);
/* The same handle over a SHARED-MEMORY transport: the `world` ranks are PROCESSES of one node (one per rank, like RCCL's), every
 * exchange is staged through a POSIX shared-memory segment named after the 128 rendezvous bytes (any bytes all ranks agree on; rank 0
 * creates the segment, creation is collective).  For nodes without a usable RCCL and - what it was written for - for running the whole
 * multi-process launch path (launcher -> ranks -> scatter -> optimize -> all-gather) with N > 1 ranks on ONE GPU, which RCCL refuses.
 * Every byte crosses the host twice: not a fast path.  Exchanges complete before they return; a rank that does not arrive within
 * AVT_SHARD_LOOPBACK_TIMEOUT_S (default 60 s here) makes its peers' calls fail, and the segment stays broken. */
int avt_shard_create_shm(int device, int rank, int world, const char id[AVT_SHARD_ID_BYTES], avt_shard** out);
void avt_shard_destroy(avt_shard* s);
int avt_shard_rank(const avt_shard* s);
int avt_shard_world(const avt_shard* s);
/* "rccl <version>, <path of the library that was opened>" (diagnostics; valid until avt_shard_destroy) */
const char* avt_shard_backend(const avt_shard* s);

/* ---- model broadcast.  `desc` is read on `root` only (NULL elsewhere); every rank, root included, receives a model
 * built from the broadcast bytes, so that all ranks provably hold the same constants. */
int avt_shard_broadcast_model(avt_shard* s, int root, const avt_model_desc* desc, avt_model** out);

/* ---- cloud scatter.  On `root`: the whole batch in avt_optimize_batch's host layout (frame f owns points
 * [frame_offsets[f], frame_offsets[f+1]) of data/labels; p/q/w frame-major start states).  Other ranks pass NULL for the
 * five arrays.  On return every rank's context holds ITS frames (in avt_shard_global_frame order) resident exactly as
 * after avt_frames_upload + avt_state_upload, ready for avt_optimize_resident.  num_frames is read on every rank and
 * must agree; local frame count must fit ctx's max_frames.  Errors are COLLECTIVE: what only one rank can see (the root's
 * arguments, a context too small for its share) is agreed on before any cloud moves, and then every rank returns non-zero -
 * no rank is left waiting in an exchange its peer never entered.  A rank that owns no frame (num_frames < world) ends with no
 * resident frames. */
int avt_shard_scatter_frames(avt_shard* s, avt_ctx* ctx, int root, int num_frames, const double* data, const int* labels,
                             const int* frame_offsets, const double* p, const double* q, const double* w);

/* ---- result gather.  Enqueues, on ctx's stream and without host synchronisation, the packing of this rank's resident
 * frames and one ncclAllGather into a device buffer owned by the shard (so it can sit inside a timed loop behind
 * avt_optimize_resident).  avt_shard_gather_wait blocks the host until the last enqueued all-gather is complete;
 * avt_shard_gather_download waits likewise and copies out, for ALL num_frames frames in global frame order: p (3), q (4J),
 * w (K) and stats (any pointer may be NULL).  avt_shard_gather_results = enqueue + download. */
int avt_shard_gather_enqueue(avt_shard* s, avt_ctx* ctx, int num_frames);
int avt_shard_gather_wait(avt_shard* s);
int avt_shard_gather_download(avt_shard* s, avt_ctx* ctx, int num_frames, double* p, double* q, double* w, avt_stats* stats);
int avt_shard_gather_results(avt_shard* s, avt_ctx* ctx, int num_frames, double* p, double* q, double* w, avt_stats* stats);

/* ---- dry runs.  With `on` != 0 a rank's OWN blocks travel through the transport as well: the scatter sends the root's share to itself
 * (grouped ncclSend / ncclRecv) and a one-rank gather keeps its ncclAllGather, so that a single GPU exercises the calls an N-rank run makes.
 * Off by default (a rank's own share is a device-to-device copy).  (Rounds 2-5 read the environment variable AVT_SHARD_SELF_SENDRECV at
 * every call; nothing on the exchange path reads the environment now.)
 * What IS read from the environment, ONCE, when a handle is created: AVT_RCCL_LIB (avt_shard_create: the first library name tried
 * before librccl.so.1 / librccl.so and torch's bundled copy) and AVT_SHARD_LOOPBACK_TIMEOUT_S (avt_shard_create_loopback /
 * avt_shard_create_shm: how long an exchange may see no progress before the group is declared broken; 60 s). */
int avt_shard_set_self_exchange(avt_shard* s, int on);

/* ---- barrier on the device (a 1-double all-gather), for hosts without another rendezvous */
int avt_shard_barrier(avt_shard* s, avt_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* AVT_SHARD_H_ */
