// The static ray tree of the reference's recursion (reference raytrace.py:103-133), enumerated
// breadth first.  Pure host code: no device call, usable (and checked under sanitizers) without a GPU.
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

namespace rt {

constexpr int HOST_TREE_MAX_NODES = 65536;

struct HostTree {
    std::vector<int32_t> a, b, parent, branch, child, level, level_start;
    int n_nodes() const { return (int)a.size(); }
    int n_levels() const { return (int)level_start.size() - 1; }
};

// A call with remaining depths (a, b) may recurse into transmit (a, b-1), total internal reflection
// (a-1, b-1) and reflect (a-1, b).  Returns an empty string, or what is wrong with the arguments.
inline std::string build_host_tree(int refl, int refr, HostTree& t) {
    if (refl < 0 || refr < 0) return "depth limits must be non-negative";
    t = HostTree();
    t.a.push_back(refl); t.b.push_back(refr); t.parent.push_back(-1); t.branch.push_back(-1);
    t.level.push_back(0); t.child.insert(t.child.end(), {-1, -1, -1});
    t.level_start.push_back(0);
    size_t begin = 0, end = 1;
    while (begin < end) {
        t.level_start.push_back((int32_t)end);
        for (size_t node = begin; node < end; ++node) {
            const int na = t.a[node], nb = t.b[node];
            const bool allowed[3] = {nb != 0, na != 0 && nb != 0, na != 0};
            const int ca[3] = {na, na - 1, na - 1}, cb[3] = {nb - 1, nb - 1, nb};
            for (int c = 0; c < 3; ++c) {
                if (!allowed[c]) continue;
                if (t.a.size() >= (size_t)HOST_TREE_MAX_NODES)
                    return "ray tree for depths (" + std::to_string(refl) + ", " + std::to_string(refr) + ") has more than " +
                           std::to_string(HOST_TREE_MAX_NODES) + " nodes";
                t.child[node * 3 + c] = (int32_t)t.a.size();
                t.a.push_back(ca[c]); t.b.push_back(cb[c]); t.parent.push_back((int32_t)node);
                t.branch.push_back(c); t.level.push_back(t.level[node] + 1);
                t.child.insert(t.child.end(), {-1, -1, -1});
            }
        }
        begin = end;
        end = t.a.size();
    }
    return std::string();
}

}  // namespace rt
