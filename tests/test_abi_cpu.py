"""Host-only checks of the C ABI: the library loads, exports every symbol the
header declares, and its pure-host entry points agree with the Python host
layer.  No kernel is launched here."""
import os
import re

import numpy as np
import pytest

from raytracing_amd import _native
from raytracing_amd.tree import build_tree, sample_offsets
from raytracing_amd.flatten import FlatLight, LIGHT_DISK, LIGHT_POINT, LIGHT_SPOT

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(REPO, "include", "raytracing_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _native.load_library()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(_native.SIGNATURES) == declared


def test_struct_sizes_match_the_header_layout():
    import ctypes as C
    assert C.sizeof(_native.RtLight) == 16 + 8 * (3 + 1 + 3 + 3 + 1 + 3 + 3 + 3 + 2)
    assert C.sizeof(_native.RtParams) == 24 + 8 * 5
    assert C.sizeof(_native.RtCounters) == 10 * 8 + 4 * 4 + 2 * 8
    assert C.sizeof(_native.RtScene) == 8 + 3 * 8 + 8 * (5 * 3 + 4)


@pytest.mark.parametrize("depths", [(3, 5), (0, 0), (1, 0), (0, 2), (2, 2), (4, 4)])
def test_native_tree_equals_python_tree(depths):
    t = build_tree(*depths)
    n = _native.build_tree_native(*depths)
    for key in ("a", "b", "parent", "branch", "child", "level", "level_start"):
        assert np.array_equal(getattr(t, key), n[key]), key


def test_tree_of_default_depths():
    t = build_tree(3, 5)
    assert t.n_nodes == 644 and t.n_levels == 9
    assert list(np.diff(t.level_start)) == [1, 3, 9, 27, 65, 131, 191, 161, 56]
    post = t.post_order()
    assert sorted(post) == list(range(644)) and post[-1] == 0
    # children finish before their parent
    where = np.empty(644, dtype=int)
    where[post] = np.arange(644)
    for k in range(1, 644):
        assert where[k] < where[t.parent[k]]


def test_tree_too_large_is_an_error():
    lib = _native.load_library()
    import ctypes as C
    nk, nl = C.c_int32(), C.c_int32()
    none = C.cast(None, C.POINTER(C.c_int32))
    rc = lib.rt_build_tree(9, 9, C.byref(nk), C.byref(nl), none, none, none, none, none, none, none)
    assert rc == _native.RT_ERR_INVALID
    assert b"nodes" in lib.rt_last_error()


def test_device_count_never_fails():
    assert _native.load_library().rt_device_count() >= 0


def test_sample_offsets_equal_the_plain_loop():
    """The offsets as a loop in the order the reference draws: nodes in post-order, lights in scene order."""
    from types import SimpleNamespace as L
    rs = np.random.RandomState(11)
    for depths in ((3, 5), (2, 0), (0, 0), (1, 4)):
        t = build_tree(*depths)
        K = t.n_nodes
        for lights in ([], [L(kind=0, draws_samples=False, n_points=1)],
                       [L(kind=1, draws_samples=True, n_points=30)],
                       [L(kind=2, draws_samples=True, n_points=7), L(kind=0, draws_samples=False, n_points=1), L(kind=1, draws_samples=True, n_points=3)]):
            hits = rs.randint(0, 50, K) * (rs.rand(K) < 0.7)
            beam = np.minimum(rs.randint(0, 50, (K, 8)), hits[:, None]) * (rs.rand(K, 8) < 0.8)
            off, cnt, total = sample_offsets(t, lights, hits, beam)
            want_off = np.full((K, max(len(lights), 1)), -1, dtype=np.int64)
            want_cnt = np.zeros_like(want_off)
            cursor = 0
            for node in t.post_order():
                if hits[node] == 0:
                    continue
                for l, light in enumerate(lights):
                    h = int(beam[node, l]) if light.kind == 2 else int(hits[node])
                    if not light.draws_samples or h == 0:
                        continue
                    want_off[node, l], want_cnt[node, l] = cursor, h
                    cursor += 2 * light.n_points * h
            assert total == cursor and np.array_equal(off, want_off) and np.array_equal(cnt, want_cnt)


def test_sample_offsets_follow_post_order():
    t = build_tree(1, 1)   # root, t-child, i-child, r-child, and the r-child's t-child ...
    K = t.n_nodes
    hits = np.arange(1, K + 1) * 2
    beam = np.zeros((K, 2), dtype=np.int64)
    beam[:, 1] = np.arange(K)
    lights = [FlatLight(LIGHT_POINT, np.zeros(3), 1.0, np.ones(3), np.ones(3)),
              FlatLight(LIGHT_SPOT, np.zeros(3), 1.0, np.ones(3), np.ones(3), n_points=5)]
    off, cnt, total = sample_offsets(t, lights, hits, beam)
    cursor = 0
    for k in t.post_order():
        assert off[k, 0] == -1
        if beam[k, 1]:
            assert off[k, 1] == cursor and cnt[k, 1] == beam[k, 1]
            cursor += 2 * 5 * beam[k, 1]
        else:
            assert off[k, 1] == -1
    assert total == cursor
    disk = [FlatLight(LIGHT_DISK, np.zeros(3), 1.0, np.ones(3), np.ones(3), n_points=3)]
    off, cnt, total = sample_offsets(t, disk, hits, beam)
    assert total == 2 * 3 * hits.sum()


@pytest.mark.parametrize("k", [5, 16])
def test_mt19937_jump_polynomial_advances_the_window(k):
    """g = x^(2^k) mod phi applied to a window of the word sequence (XOR of the windows
    at the set coefficients) is the window 2^k words later."""
    g = _native.mt19937_jump_poly(k)
    bits = np.flatnonzero(np.unpackbits(g.view(np.uint8), bitorder="little"))
    assert bits.max() < 19937
    if k == 5:
        assert list(bits) == [32]             # below the degree the power of x is not reduced
    # an independent word sequence: MT19937 recurrence in NumPy
    x = np.empty(2 ** k + 19937 + 2 * 624 + 100, dtype=np.uint32)
    x[:624] = np.random.RandomState(99).get_state()[1]
    for n in range(624, x.size):
        y = (x[n - 624] & np.uint32(0x80000000)) | (x[n - 623] & np.uint32(0x7FFFFFFF))
        x[n] = x[n - 227] ^ (y >> np.uint32(1)) ^ (np.uint32(0x9908B0DF) if y & np.uint32(1) else np.uint32(0))
    n0 = 37
    acc = np.zeros(624, dtype=np.uint32)
    for i in bits:
        acc ^= x[n0 + i:n0 + i + 624]
    want = x[n0 + 2 ** k:n0 + 2 ** k + 624]
    assert np.array_equal(acc[1:], want[1:])
    assert (acc[0] ^ want[0]) & np.uint32(0x80000000) == 0     # only the top bit of word 0 is state


def test_mt19937_jump_polynomial_range():
    with pytest.raises(_native.RtError):
        _native.mt19937_jump_poly(48)


# ---------------------------------------------------------------------------------------------
# The sphere grouping of many-sphere scenes (host code of the library): every sphere has exactly one
# place, and every bounding sphere contains what it bounds - the property the device's "the ray misses
# the bound, so it misses every member" rests on.
@pytest.mark.parametrize("n,seed", [(65, 0), (1000, 1), (4096, 2), (5003, 3)])
def test_hierarchy_partition_and_bounds(n, seed):
    from raytracing_amd._native import build_hierarchy_native
    rs = np.random.RandomState(seed)
    geom = np.column_stack([rs.uniform(2, 60, n), rs.uniform(-25, 25, n), rs.uniform(0.3, 12, n), rs.uniform(0.15, 0.6, n)])
    geom[0] = [4.0, 0.0, -1.0e7, 1.0e7]                       # a ground sphere: far larger than the rest
    if n > 500:
        geom[7] = geom[8]                                     # two identical spheres
        geom[9, 3] = 0.0                                      # a point
    slots, clusters, blocks, loose = build_hierarchy_native(geom)
    assert list(loose) == [0]
    placed = slots[slots != 0xFFFF]
    assert sorted(placed.tolist() + list(loose)) == list(range(n))          # one place each
    assert blocks.shape[0] == -(-(n - 1) // 64)
    c, r = geom[:, :3], np.abs(geom[:, 3])
    for b in range(slots.shape[0]):
        in_block = []
        for k in range(8):
            members = slots[b, k][slots[b, k] != 0xFFFF].astype(int)
            if len(members) == 0:
                assert clusters[b, k, 3] == 0 and clusters[b, k, 0] > 1e14   # padding: far away
                continue
            in_block += members.tolist()
            reach = np.linalg.norm(c[members] - clusters[b, k, :3].astype(float), axis=1) + r[members]
            assert reach.max() <= float(clusters[b, k, 3])
        reach = np.linalg.norm(c[in_block] - blocks[b, :3].astype(float), axis=1) + r[in_block]
        assert reach.max() <= float(blocks[b, 3])
    # the groups are compact: a block's bound is far smaller than the scene
    assert np.median(blocks[:, 3]) < (30.0 if n >= 1000 else 60.0)


def test_hierarchy_degenerate_inputs():
    from raytracing_amd._native import build_hierarchy_native
    slots, clusters, blocks, loose = build_hierarchy_native(np.zeros((0, 4)))
    assert slots.shape[0] == 0 and len(loose) == 0
    same = np.tile([1.0, 2.0, 3.0, 0.5], (130, 1))           # all spheres identical
    slots, clusters, blocks, loose = build_hierarchy_native(same)
    assert len(loose) == 0 and sorted(slots[slots != 0xFFFF].tolist()) == list(range(130))
    bad = same.copy()
    bad[5, 0] = np.nan
    bad[6, 3] = np.inf
    slots, clusters, blocks, loose = build_hierarchy_native(bad)
    assert sorted(loose.tolist()) == [5, 6]                  # non-finite spheres are never grouped
