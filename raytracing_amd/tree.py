"""Static shape of the reference's ray tree and the order in which its nodes
consume the global random stream.

A *node* is one call of the reference's recursive tracer (reference
raytrace.py:19): all rays of the image that followed the same sequence of
branches.  From a node with remaining depths ``(a, b)`` = (reflection,
refraction) the reference may recurse into

* branch 0, transmitted rays, depths ``(a, b-1)``, when ``b != 0``
  (reference raytrace.py:103-107);
* branch 1, totally internally reflected rays, ``(a-1, b-1)``, when
  ``a != 0 and b != 0`` (reference raytrace.py:122-127);
* branch 2, mirror-reflected rays, ``(a-1, b)``, when ``a != 0``
  (reference raytrace.py:130-133);

in that order, and only afterwards evaluates the lights of the node itself
(reference raytrace.py:140-147) - so disk-light samples are drawn in
*post-order*.  Nodes are numbered breadth first so that every level of the tree
is a contiguous id range, which is the layout the device queues use.
"""
from dataclasses import dataclass

import numpy as np

BRANCH_TRANSMIT, BRANCH_TIR, BRANCH_REFLECT = 0, 1, 2


@dataclass
class RayTree:
    a: np.ndarray            # (K,) remaining reflection depth
    b: np.ndarray            # (K,) remaining refraction depth
    parent: np.ndarray       # (K,) parent node id, -1 for the root
    branch: np.ndarray       # (K,) branch taken from the parent, -1 for the root
    child: np.ndarray        # (K, 3) child node ids, -1 where the depth forbids the branch
    level: np.ndarray        # (K,) bounce level
    level_start: np.ndarray  # (L + 1,) first node id of each level

    @property
    def n_nodes(self):
        return self.a.shape[0]

    @property
    def n_levels(self):
        return self.level_start.shape[0] - 1

    def post_order(self):
        """Node ids in the order the reference finishes them (children in branch
        order first, then the node)."""
        order, stack = [], [(0, 0)]
        while stack:
            node, nxt = stack.pop()
            while nxt < 3 and self.child[node, nxt] < 0:
                nxt += 1
            if nxt < 3:
                stack.append((node, nxt + 1))
                stack.append((int(self.child[node, nxt]), 0))
            else:
                order.append(node)
        return np.array(order, dtype=np.int64)

    def pre_order(self):
        order, stack = [], [0]
        while stack:
            node = stack.pop()
            order.append(node)
            for c in (2, 1, 0):
                if self.child[node, c] >= 0:
                    stack.append(int(self.child[node, c]))
        return np.array(order, dtype=np.int64)


def build_tree(max_reflection_depth, max_refraction_depth):
    a, b, parent, branch, level = [max_reflection_depth], [max_refraction_depth], [-1], [-1], [0]
    child = [[-1, -1, -1]]
    starts = [0]
    frontier = [0]
    while frontier:
        starts.append(len(a))          # end of this level = start of the next
        nxt = []
        for node in frontier:
            na, nb = a[node], b[node]
            options = ((nb != 0, na, nb - 1), (na != 0 and nb != 0, na - 1, nb - 1), (na != 0, na - 1, nb))
            for c, (allowed, ca, cb) in enumerate(options):
                if not allowed:
                    continue
                cid = len(a)
                a.append(ca); b.append(cb); parent.append(node); branch.append(c)
                level.append(level[node] + 1); child.append([-1, -1, -1])
                child[node][c] = cid
                nxt.append(cid)
        frontier = nxt
    return RayTree(a=np.array(a, dtype=np.int32), b=np.array(b, dtype=np.int32),
                   parent=np.array(parent, dtype=np.int32), branch=np.array(branch, dtype=np.int32),
                   child=np.array(child, dtype=np.int32), level=np.array(level, dtype=np.int32),
                   level_start=np.array(starts, dtype=np.int32))


def sample_offsets(tree, lights, node_hits, node_beam):
    """Offsets into the random stream of every (node, light) sample block.

    ``node_hits[k]`` is the number of rays of node ``k`` that hit a sphere and
    ``node_beam[k, l]`` the number of those hits inside spotlight ``l``'s beam
    (unused for other kinds of light).  A disk light draws ``2 * n_points * h``
    uniforms per node - first all angles, then all radii (reference
    objects.py:132-133) - with ``h`` the hit count, or the in-beam count for a
    spotlight (reference objects.py:216-217).  Nodes are visited in post-order,
    lights in scene order.  Returns ``(offsets (K, L) int64, counts (K, L) int64,
    total)``; offsets of point lights and of empty blocks are -1.
    """
    K, L = tree.n_nodes, len(lights)
    offsets = np.full((K, max(L, 1)), -1, dtype=np.int64)
    counts = np.zeros((K, max(L, 1)), dtype=np.int64)
    if L == 0:
        return offsets, counts, 0
    order = getattr(tree, "_post_order", None)
    if order is None:
        order = tree._post_order = tree.post_order()
    hits = np.asarray(node_hits, dtype=np.int64).reshape(K)
    for l, light in enumerate(lights):
        if light.draws_samples:
            h = np.asarray(node_beam, dtype=np.int64).reshape(K, -1)[:, l] if light.kind == 2 else hits
            counts[:, l] = np.where(hits != 0, h, 0)
    n_points = np.array([light.n_points if light.draws_samples else 0 for light in lights], dtype=np.int64)
    sizes = (2 * n_points[None, :] * counts[:, :L])[order]            # in drawing order: node by node, light by light
    ends = np.cumsum(sizes.ravel())
    starts = (ends - sizes.ravel()).reshape(sizes.shape)
    offsets[order, :L] = np.where(sizes > 0, starts, -1)
    return offsets, counts, int(ends[-1]) if ends.size else 0
