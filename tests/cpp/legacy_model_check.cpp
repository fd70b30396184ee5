// legacy_model_check.cpp — CPU-only check of ark::AvatarModel's loaders (include/ark/Avatar.h): loads a model directory (model.npz
// or the reference's legacy text format, AvatarModel.cpp:128-288) and prints what the tests compare: dimensions, the main joint
// of every point, initialJointPos and jointShapeReg.  Usage: legacy_model_check <model_dir> [limit_one_joint_per_point]
#include <cstdio>
#include <cstdlib>

#include "ark/Avatar.h"

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    ark::AvatarModel m(argv[1], argc > 2 && std::atoi(argv[2]) != 0);
    std::printf("dims %d %d %d %d\n", m.numPoints(), m.numJoints(), m.numShapeKeys(), m.numFaces());
    std::printf("parent");
    for (int j = 0; j < m.numJoints(); ++j) std::printf(" %d", m.parent[j]);
    std::printf("\nmain_joint");
    for (int v = 0; v < m.numPoints(); ++v) std::printf(" %d", m.mainJoint[v]);
    std::printf("\ninitial_joint_pos");
    for (int i = 0; i < 3 * m.numJoints(); ++i) std::printf(" %.17g", m.initialJointPos.data()[i]);
    std::printf("\njoint_shape_reg");
    for (double v : m.jointShapeReg) std::printf(" %.17g", v);
    std::printf("\nuse_jsr %d\n", m.useJointShapeRegressor ? 1 : 0);
    return 0;
}
