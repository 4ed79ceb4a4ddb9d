#include "BVH.hpp"

#include <algorithm>
#include <stdexcept>

BVHAccel::BVHAccel(const std::vector<Object*>& p, int maxPrims, SplitMethod method)
    : maxPrimsInNode(std::min(255, maxPrims)), splitMethod(method) {
    if (method != SplitMethod::NAIVE) throw std::invalid_argument("BVHAccel: only SplitMethod::NAIVE is implemented");
    if (p.empty()) return;
    const size_t n = p.size();
    std::vector<Bounds3> bounds(n);
    std::vector<Vec3> centroids(n);
    std::vector<float> areas(n);
    for (size_t i = 0; i < n; ++i) {
        bounds[i] = p[i]->getBounds();
        centroids[i] = bounds[i].Centroid();
        areas[i] = p[i]->area;
    }
    std::vector<int32_t> order(n);
    for (size_t i = 0; i < n; ++i) order[i] = (int32_t)i;
    nodes.reserve(2 * n);
    right_child_.reserve(2 * n);
    recursiveBuild(order, 0, n, bounds, centroids, areas);
    setMiss(0, (int32_t)nodes.size());
    right_child_.clear();
    right_child_.shrink_to_fit();
}

// Emits the subtree over order[lo, hi) in pre-order and returns its root index.
// The split rules are those of common/BVH.cpp:89-165: 1 primitive -> leaf;
// 2 -> left = first, right = second, *unsorted*; >= 3 -> sort the range by
// centroid along the widest axis of the centroid bounds, split at size/2.
int32_t BVHAccel::recursiveBuild(std::vector<int32_t>& order, size_t lo, size_t hi, const std::vector<Bounds3>& bounds,
                                 const std::vector<Vec3>& centroids, const std::vector<float>& areas) {
    const int32_t self = (int32_t)nodes.size();
    nodes.emplace_back();
    right_child_.push_back(-1);
    const size_t count = hi - lo;
    if (count == 1) {
        const int32_t obj = order[lo];
        nodes[self].bounds = bounds[obj];
        nodes[self].object = obj;
        nodes[self].area = areas[obj];
        return self;
    }
    size_t mid = lo + 1;
    if (count > 2) {
        Bounds3 centroidBounds;
        for (size_t i = lo; i < hi; ++i) centroidBounds = Union(centroidBounds, centroids[order[i]]);
        const int dim = centroidBounds.maxExtent();
        auto first = order.begin() + (std::ptrdiff_t)lo, last = order.begin() + (std::ptrdiff_t)hi;
        switch (dim) {
        case 0: std::sort(first, last, [&](int32_t a, int32_t b) { return centroids[a].x < centroids[b].x; }); break;
        case 1: std::sort(first, last, [&](int32_t a, int32_t b) { return centroids[a].y < centroids[b].y; }); break;
        default: std::sort(first, last, [&](int32_t a, int32_t b) { return centroids[a].z < centroids[b].z; }); break;
        }
        mid = lo + count / 2;
    }
    const int32_t left = recursiveBuild(order, lo, mid, bounds, centroids, areas);
    const int32_t right = recursiveBuild(order, mid, hi, bounds, centroids, areas);
    right_child_[self] = right;
    nodes[self].bounds = Union(nodes[left].bounds, nodes[right].bounds);
    nodes[self].area = nodes[left].area + nodes[right].area;
    return self;
}

// left child's miss link = right sibling, right child's = the parent's
// (common/BVH.cpp:22-37).  Iterative: the tree can be deep for large meshes.
void BVHAccel::setMiss(int32_t root, int32_t val) {
    std::vector<std::pair<int32_t, int32_t>> stack{{root, val}};
    while (!stack.empty()) {
        auto [node, miss] = stack.back();
        stack.pop_back();
        nodes[node].miss = miss;
        if (nodes[node].object >= 0) continue;
        const int32_t left = node + 1, right = right_child_[node];
        stack.emplace_back(left, right);
        stack.emplace_back(right, miss);
    }
}

std::vector<int32_t> BVHAccel::leafOrder() const {
    std::vector<int32_t> out;
    for (const FlatNode& n : nodes)
        if (n.object >= 0) out.push_back(n.object);
    return out;
}
