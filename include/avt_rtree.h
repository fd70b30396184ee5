/* avt_rtree.h — C ABI of the body-part forest inference, part of libavatar_hip.so.
 *
 * The stage immediately before the fitting path: `ark::RTree` turns a foreground depth image into the per-pixel
 * body-part labels AvatarOptimizer::optimize() consumes (demo.cpp:196-268).  Each entry point names the reference
 * interface it replaces (file:line relative to the reference tree); include/ark/RTree.h re-creates the class on top.
 *
 * Conventions: images are row-major, depth is float32 metres with 0 = background (demo.cpp:185-191), labels are
 * uint8 with 255 = none; a region of interest is (top_left, bot_right) inclusive, bot_right.x == -1 means the whole
 * image (RTree.cpp:3190-3193).  Functions return 0 on success; avt_last_error() (avt.h) describes a failure.
 */
#ifndef AVT_RTREE_H_
#define AVT_RTREE_H_

#ifdef __cplusplus
extern "C" {
#endif

typedef struct avt_rtree avt_rtree;

/* The members of `class RTree` (RTree.h:171-184): nodes, leafData, numParts, partMap (+ its type). */
typedef struct avt_rtree_desc {
    int n_nodes;               /* nodes.size(); node 0 is the root                           (RTree.h:171) */
    int n_leafs;               /* leafData.size()                                            (RTree.h:172) */
    int num_parts;             /* numParts, < 128 (post-processing marks visited labels +128) (RTree.h:175) */
    const float* feature;      /* n_nodes x 5: u.x u.y v.x v.y thresh                        (RTree.h:28-41) */
    const int* links;          /* n_nodes x 3: lnode rnode leafid (leafid == -1: internal)    (RTree.h:36-40) */
    const float* leaf_data;    /* n_leafs x num_parts row-major distributions                 (RTree.h:172) */
    int part_map_len;          /* 0: no part map                                              (RTree.h:177) */
    const int* part_map;       /* SMPL joint -> part                                          (RTree.cpp:3465-3509) */
    int part_map_type;         /* 0 contiguous, 1 disjoint                                    (RTree.cpp:3471-3476) */
} avt_rtree_desc;

/* RTree(int num_parts) + filled members.  Copies everything, uploads the packed tree to `device`; device < 0 makes a
 * host-only tree (file formats, members, post-processing) whose inference calls fail. */
int avt_rtree_create(const avt_rtree_desc* desc, int device, avt_rtree** out);
/* RTree::loadFile (RTree.cpp:2967-3064): binary 'R'..'T' or legacy text format, plus "<path>.partmap" if present. */
int avt_rtree_load(const char* path, int device, avt_rtree** out);
/* RTree::exportFile (RTree.cpp:3066-3120), binary format. */
int avt_rtree_export(const avt_rtree* rt, const char* path);
void avt_rtree_destroy(avt_rtree* rt);
int avt_rtree_info(const avt_rtree* rt, int* n_nodes, int* n_leafs, int* num_parts, int* part_map_len, int* part_map_type);
/* copies the members out; any pointer may be NULL.  leaf_best = leafBestMatch (RTree.cpp:3451-3463). */
int avt_rtree_get(const avt_rtree* rt, float* feature, int* links, float* leaf_data, unsigned char* leaf_best, int* part_map);

/* cv::Mat RTree::predictBest(depth, num_threads, interval, top_left, bot_right, fill_in_gaps) (RTree.cpp:3184-3262):
 * host image in, host labels out (rows x cols bytes).  Runs on the GPU; num_threads has no counterpart. */
int avt_rtree_predict_best(avt_rtree* rt, const float* depth, int rows, int cols, int interval, int tl_x, int tl_y, int br_x, int br_y,
                           int fill_in_gaps, unsigned char* labels_out);

/* std::vector<cv::Mat> RTree::predict(depth) (RTree.cpp:3156-3182): the leaf distribution of every pixel with depth > 0,
 * probes bounded by the image; out = num_parts planes of rows x cols float32 (0 where depth <= 0). */
int avt_rtree_predict(avt_rtree* rt, const float* depth, int rows, int cols, float* dist_out);

/* Batch form for resident images (throughput use, bench.py): upload n images once, label them all with one launch
 * sequence on the tree's stream, download what is needed.  avt_rtree_sync waits for the stream. */
int avt_rtree_images_upload(avt_rtree* rt, int n_images, int rows, int cols, const float* depth);
int avt_rtree_predict_best_resident(avt_rtree* rt, int interval, int tl_x, int tl_y, int br_x, int br_y, int fill_in_gaps);
int avt_rtree_labels_download(avt_rtree* rt, int image, unsigned char* labels_out);
int avt_rtree_sync(avt_rtree* rt);

/* void RTree::postProcess(image, com_pre, interval, num_threads, top_left, bot_right, dist_to_pre_weight) const
 * (RTree.cpp:3422-3449): largest-component selection per part ('contiguous' part maps) or small-piece removal
 * ('disjoint'), then up-scaling of the interval grid.  Host code, as in the reference (a sequential flood fill whose
 * scan order defines the result).  com_pre: 2 x num_parts column-major; com_pre_valid == 0 means "not sized yet"
 * (the resize branch of :3431-3435). */
int avt_rtree_post_process(const avt_rtree* rt, unsigned char* image, int rows, int cols, double* com_pre, int com_pre_valid, int interval,
                           int tl_x, int tl_y, int br_x, int br_y, double dist_to_pre_weight);

#ifdef __cplusplus
}
#endif
#endif
