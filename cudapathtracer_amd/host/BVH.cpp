#include "BVH.hpp"

#include <algorithm>
#include <stdexcept>

BVHAccel::BVHAccel(const std::vector<Object*>& p, int maxPrims, SplitMethod method, const std::vector<int32_t>* leaf_need)
    : maxPrimsInNode(std::min(255, maxPrims)), splitMethod(method) {
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
    if (method == SplitMethod::SAH) {
        std::vector<SahNode> tmp;
        tmp.reserve(2 * n);
        const int32_t root = sahBuild(tmp, order, 0, n, bounds, centroids, areas, leaf_need);
        sahEmit(tmp, root);
    } else {
        recursiveBuild(order, 0, n, bounds, centroids, areas);
    }
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

// SplitMethod::SAH (declared by the reference, common/BVH.hpp:28, without an implementation): the split of
// order[lo, hi) that minimises area(left) * count(left) + area(right) * count(right) over 32 centroid bins on each
// axis; the range is partitioned in place and the first index of the right part returned.  Centroids that all fall
// into one bin on every axis (coincident primitives) are split in the middle.  Deterministic.
size_t BVHAccel::sahSplit(std::vector<int32_t>& order, size_t lo, size_t hi, const std::vector<Bounds3>& bounds,
                          const std::vector<Vec3>& centroids) {
    constexpr int kBins = 32;
    Bounds3 cb;
    for (size_t i = lo; i < hi; ++i) cb = Union(cb, centroids[order[i]]);
    const Vec3 ext = cb.Diagonal();
    auto half_area = [](const Bounds3& b) {
        const Vec3 d = b.Diagonal();
        return (double)d.x * d.y + (double)d.y * d.z + (double)d.z * d.x;
    };
    auto coord = [](const Vec3& v, int axis) { return axis == 0 ? v.x : axis == 1 ? v.y : v.z; };
    double best_cost = 0.0;
    int best_axis = -1, best_bin = 0;
    for (int axis = 0; axis < 3; ++axis) {
        const float e = coord(ext, axis), c0 = coord(cb.pMin, axis);
        if (!(e > 0.f)) continue;
        const float scale = (float)kBins / e;
        Bounds3 bin_bounds[kBins];
        size_t bin_count[kBins] = {};
        for (size_t i = lo; i < hi; ++i) {
            const int b = std::min(kBins - 1, (int)((coord(centroids[order[i]], axis) - c0) * scale));
            bin_bounds[b] = Union(bin_bounds[b], bounds[order[i]]);
            ++bin_count[b];
        }
        // cost of splitting after bin s: left = bins [0, s], right = bins (s, kBins)
        double right_area[kBins];
        size_t right_count[kBins];
        Bounds3 acc;
        size_t cnt = 0;
        for (int b = kBins - 1; b > 0; --b) {
            if (bin_count[b]) acc = Union(acc, bin_bounds[b]);
            cnt += bin_count[b];
            right_area[b] = cnt ? half_area(acc) : 0.0;
            right_count[b] = cnt;
        }
        acc = Bounds3();
        cnt = 0;
        for (int s = 0; s + 1 < kBins; ++s) {
            if (bin_count[s]) acc = Union(acc, bin_bounds[s]);
            cnt += bin_count[s];
            if (cnt == 0 || right_count[s + 1] == 0) continue;
            const double cost = half_area(acc) * (double)cnt + right_area[s + 1] * (double)right_count[s + 1];
            if (best_axis < 0 || cost < best_cost) {
                best_cost = cost;
                best_axis = axis;
                best_bin = s;
            }
        }
    }
    if (best_axis < 0) return lo + (hi - lo) / 2;
    const float e = coord(ext, best_axis), c0 = coord(cb.pMin, best_axis), scale = (float)kBins / e;
    auto first = order.begin() + (std::ptrdiff_t)lo, last = order.begin() + (std::ptrdiff_t)hi;
    auto midit = std::stable_partition(first, last, [&](int32_t a) {
        return std::min(kBins - 1, (int)((coord(centroids[a], best_axis) - c0) * scale)) <= best_bin;
    });
    return lo + (size_t)(midit - first);
}

// The SAH tree is built into a scratch array first, because the order of a node's children is chosen afterwards:
// a depth-first walk that descends into one child and keeps the other on its stack needs
// max(1 + need(first), need(second)) entries, so the child whose subtree needs fewer goes first.  With that order
// no tree over n leaves needs more than log2(n) entries -- never more than the balanced median-split tree, however
// uneven the SAH splits are (the per-lane stacks of the traversal kernels live in LDS and bound the occupancy).
int32_t BVHAccel::sahBuild(std::vector<SahNode>& tmp, std::vector<int32_t>& order, size_t lo, size_t hi, const std::vector<Bounds3>& bounds,
                           const std::vector<Vec3>& centroids, const std::vector<float>& areas, const std::vector<int32_t>* leaf_need) {
    const int32_t self = (int32_t)tmp.size();
    tmp.emplace_back();
    const size_t count = hi - lo;
    if (count == 1) {
        const int32_t obj = order[lo];
        tmp[self].bounds = bounds[obj];
        tmp[self].object = obj;
        tmp[self].area = areas[obj];
        if (leaf_need && (*leaf_need)[(size_t)obj] >= 0) {
            tmp[self].need = (*leaf_need)[(size_t)obj];
            tmp[self].heavy = true;
        }
        return self;
    }
    const size_t mid = count == 2 ? lo + 1 : sahSplit(order, lo, hi, bounds, centroids);
    const int32_t a = sahBuild(tmp, order, lo, mid, bounds, centroids, areas, leaf_need);
    const int32_t b = sahBuild(tmp, order, mid, hi, bounds, centroids, areas, leaf_need);
    const bool a_first = tmp[a].need <= tmp[b].need;
    SahNode& n = tmp[self];
    n.first = a_first ? a : b;
    n.second = a_first ? b : a;
    n.bounds = Union(tmp[n.first].bounds, tmp[n.second].bounds);
    n.area = tmp[n.first].area + tmp[n.second].area;
    // leaves are tested, not pushed: only an interior second child occupies a stack entry
    n.heavy = true;
    n.need = std::max(tmp[n.first].need + (tmp[n.second].heavy ? 1 : 0), tmp[n.second].need);
    return self;
}

// pre-order emission of the scratch tree (iterative: first child right after its parent)
void BVHAccel::sahEmit(const std::vector<SahNode>& tmp, int32_t root) {
    struct Item {
        int32_t node, parent;  // parent = index in `nodes` whose right child this is, or -1
    };
    std::vector<Item> stack{{root, -1}};
    while (!stack.empty()) {
        const Item it = stack.back();
        stack.pop_back();
        const SahNode& t = tmp[it.node];
        const int32_t self = (int32_t)nodes.size();
        nodes.emplace_back();
        right_child_.push_back(-1);
        if (it.parent >= 0) right_child_[it.parent] = self;
        nodes[self].bounds = t.bounds;
        nodes[self].area = t.area;
        nodes[self].object = t.object;
        if (t.object >= 0) continue;
        stack.push_back({t.second, self});
        stack.push_back({t.first, -1});
    }
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

int32_t BVHAccel::stackNeed() const {
    std::vector<int32_t> need(nodes.size(), 0);
    for (size_t i = nodes.size(); i-- > 0;) {  // children follow their parent in pre-order
        if (nodes[i].object >= 0) continue;
        const size_t left = i + 1, right = (size_t)nodes[left].miss;
        need[i] = std::max(need[left] + (nodes[right].object >= 0 ? 0 : 1), need[right]);
    }
    return nodes.empty() ? 0 : need[0];
}

std::vector<int32_t> BVHAccel::leafOrder() const {
    std::vector<int32_t> out;
    for (const FlatNode& n : nodes)
        if (n.object >= 0) out.push_back(n.object);
    return out;
}
