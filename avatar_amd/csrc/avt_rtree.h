// avt_rtree.h (private) — host-side tree and the device image of it
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "../../include/avt_rtree.h"

// one tree node as the kernel reads it: two 16-byte loads
struct RtNodeDev {
    float ux, uy, vx, vy;
    float thresh;
    int lnode;      // internal: left child; leaf: best-match label
    int rnode;      // internal: right child; leaf: leaf id (row of the distribution table)
    int leaf;       // 1: leaf
};

struct avt_rtree {
    int device = 0;
    int num_parts = 0;
    std::vector<float> feature;      // n x 5
    std::vector<int> links;          // n x 3
    std::vector<float> leaf_data;    // nl x num_parts
    std::vector<unsigned char> leaf_best;
    std::vector<int> part_map;
    int part_map_type = 0;
    // device
    hipStream_t stream = nullptr;
    RtNodeDev* d_nodes = nullptr;
    float* d_leaf = nullptr;         // [n_leafs][num_parts] distributions
    float* d_depth = nullptr;
    unsigned char* d_labels = nullptr;
    size_t cap_pixels = 0;           // capacity of d_depth / d_labels in pixels
    int n_images = 0, rows = 0, cols = 0;
};

int avt_rtree_launch_predict_dist(avt_rtree* rt, int rows, int cols, float* d_out);
int avt_rtree_launch_predict(avt_rtree* rt, int n_images, int rows, int cols, int interval, int tlx, int tly, int brx, int bry, int fill);
