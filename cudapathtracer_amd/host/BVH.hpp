// BVHAccel: host BVH builder that emits a flattened, pre-order ("threaded")
// node array instead of the reference's pointer graph of BVHBuildNode
// (common/BVH.hpp:55-80).  The *topology* is the reference's: recursive median
// split on the longest axis of the centroid bounds with std::sort, one
// primitive per leaf, two primitives split in their current order
// (common/BVH.cpp:81-168), because closest-hit ids are only reproducible if the
// same set of boxes is tested for every ray.
//
// Layout of one level (FlatNode array, pre-order):
//   node visited after a box hit  = index + 1   (the reference's nextIfHit, BVH.cpp:53-62)
//   node visited after a box miss = miss        (the reference's nextIfMiss, BVH.cpp:22-37)
//   leaf: object >= 0 (index into the primitive list the BVH was built over)
// Scene::buildBVH splices the per-mesh levels into the top level and writes the
// 32-byte node / 64-byte triangle records the device consumes.
#pragma once
#include <cstdint>
#include <vector>

#include "Bounds3.hpp"
#include "Triangle.hpp"

struct FlatNode {
    Bounds3 bounds;
    int32_t miss = 0;     // pre-order index of the node to visit after a miss; node count = end
    int32_t object = -1;  // leaf: primitive index; interior: -1
    float area = 0.f;     // sum of child areas in tree order (BVH.cpp:121,164)
};

class BVHAccel {
public:
    enum class SplitMethod { NAIVE, SAH };

    // primitives are borrowed for the duration of the constructor
    // leaf_need (SAH only, optional): per primitive, the stack entries of the subtree that will stand in its leaf's
    // place once the tree is flattened (a mesh under the top level); -1 = the leaf stays a leaf.  The child order
    // takes it into account.
    BVHAccel(const std::vector<Object*>& p, int maxPrimsInNode = 1, SplitMethod splitMethod = SplitMethod::NAIVE,
             const std::vector<int32_t>* leaf_need = nullptr);
    // stack entries a depth-first walk needs that descends into a node's first child and keeps the second, if it is
    // not a leaf, on its stack
    int32_t stackNeed() const;

    Bounds3 WorldBound() const { return nodes.empty() ? Bounds3() : nodes[0].bounds; }
    float rootArea() const { return nodes.empty() ? 0.f : nodes[0].area; }
    // primitive indices in the order the traversal meets the leaves
    std::vector<int32_t> leafOrder() const;

    std::vector<FlatNode> nodes;  // pre-order
    const int maxPrimsInNode;
    const SplitMethod splitMethod;

private:
    int32_t recursiveBuild(std::vector<int32_t>& order, size_t lo, size_t hi, const std::vector<Bounds3>& bounds,
                           const std::vector<Vec3>& centroids, const std::vector<float>& areas);
    struct SahNode {
        Bounds3 bounds;
        float area = 0.f;
        int32_t object = -1, first = -1, second = -1;
        int32_t need = 0;  // stack entries a depth-first walk of the subtree needs
        bool heavy = false;  // an interior node, or a leaf a subtree will replace: occupies a stack entry while it waits
    };
    int32_t sahBuild(std::vector<SahNode>& tmp, std::vector<int32_t>& order, size_t lo, size_t hi, const std::vector<Bounds3>& bounds,
                     const std::vector<Vec3>& centroids, const std::vector<float>& areas, const std::vector<int32_t>* leaf_need);
    void sahEmit(const std::vector<SahNode>& tmp, int32_t root);
    size_t sahSplit(std::vector<int32_t>& order, size_t lo, size_t hi, const std::vector<Bounds3>& bounds, const std::vector<Vec3>& centroids);
    void setMiss(int32_t node, int32_t val);
    std::vector<int32_t> right_child_;  // build-time only
};
