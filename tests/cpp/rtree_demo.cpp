// rtree_demo.cpp — the labelling step of the reference's tracker loop through the C++ facade (demo.cpp:133, :196-204):
//   argv[1] tree file, argv[2] depth.bin (int rows, cols, tl.x, tl.y, br.x, br.y; rows*cols floats), argv[3] out.bin
//   (rows*cols label bytes after predictBest + postProcess, then 2*numParts doubles com_pre).
#include <cstdio>

#include "ark/RTree.h"

int main(int argc, char** argv) {
    if (argc < 4) { std::fprintf(stderr, "usage: rtree_demo tree depth.bin out.bin\n"); return 2; }
    ark::RTree rtree(argv[1]);
    if (rtree.numParts <= 0) return 2;
    FILE* f = std::fopen(argv[2], "rb");
    if (!f) { std::perror("depth"); return 2; }
    int hdr[6];
    if (std::fread(hdr, sizeof(int), 6, f) != 6) return 2;
    ark::ImageF depth(hdr[0], hdr[1]);
    if (std::fread(depth.data(), sizeof(float), depth.a.size(), f) != depth.a.size()) return 2;
    std::fclose(f);
    const ark::Point topLeft(hdr[2], hdr[3]), botRight(hdr[4], hdr[5]);
    ark::MatrixNX<2> comPre;                                            // demo.cpp:148
    ark::Image8 result = rtree.predictBest(depth, 8, 2, topLeft, botRight);
    rtree.postProcess(result, comPre, 2, 8, topLeft, botRight);
    FILE* o = std::fopen(argv[3], "wb");
    std::fwrite(result.data(), 1, result.a.size(), o);
    std::fwrite(comPre.data(), sizeof(double), comPre.size(), o);
    std::fclose(o);
    size_t labelled = 0;
    for (uint8_t v : result.a) labelled += v != 255;
    std::printf("rtree_demo: %d parts, %zu nodes, %zu labelled pixels\n", rtree.numParts, rtree.nodes.size(), labelled);
    return 0;
}
