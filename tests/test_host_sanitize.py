"""The library's pure-host code (static ray tree, sphere grouping of many-sphere scenes, MT19937
jump polynomials) under AddressSanitizer and UndefinedBehaviorSanitizer, on the CPU: the same
headers the library is built from are compiled into a small program with the sanitizers on the host
side only, and what it computes is compared with the Python host layer and with the exports of the
ordinary build.  A sanitizer report ends the program with a non-zero exit code."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from raytracing_amd import _native
from raytracing_amd.Scenes import synthetic
from raytracing_amd.flatten import flatten_scene
from raytracing_amd.tree import build_tree

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    if not (os.path.exists(HIPCC) or shutil.which(HIPCC)):
        pytest.skip("hipcc not available")
    exe = str(tmp_path_factory.mktemp("host_sanitize") / "host_checks")
    cmd = [HIPCC, "-O1", "-g", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off",
           "-Xarch_host", "-fsanitize=address", "-Xarch_host", "-fsanitize=undefined",
           "-Xarch_host", "-fno-sanitize-recover=all", "-Xarch_host", "-fno-omit-frame-pointer",
           os.path.join(REPO, "tests", "host_sanitize", "host_checks.hip"), "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)

    def run(*args):
        env = dict(os.environ, ASAN_OPTIONS="verify_asan_link_order=0:detect_leaks=0:abort_on_error=0",
                   UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
        res = subprocess.run([exe, *map(str, args)], capture_output=True, text=True, env=env, timeout=300)
        assert res.returncode == 0, res.stderr[-4000:]
        assert "runtime error" not in res.stderr and "AddressSanitizer" not in res.stderr, res.stderr[-4000:]
        out = {}
        for line in res.stdout.splitlines():
            name, _, rest = line.partition(" ")
            out[name] = rest.split()
        return out
    return run


@pytest.mark.parametrize("depths", [(3, 5), (0, 0), (1, 0), (0, 2), (4, 4)])
def test_tree_under_sanitizers(checker, depths):
    out = checker("tree", *depths)
    t = build_tree(*depths)
    for key in ("a", "b", "parent", "branch", "level", "level_start"):
        assert [int(v) for v in out[key]] == list(getattr(t, key)), key
    assert [int(v) for v in out["child"]] == list(np.asarray(t.child).reshape(-1))


def test_tree_limits_under_sanitizers(checker):
    assert "non-negative" in " ".join(checker("tree", -1, 2)["error"])
    assert "more than" in " ".join(checker("tree", 12, 12)["error"])


def _compare_grouping(out, geom):
    slots, clusters, blocks, loose = _native.build_hierarchy_native(geom)
    assert int(out["blocks"][0]) == slots.shape[0] and int(out["loose"][0]) == loose.size
    assert [int(v) for v in out["ids"]] == list(slots.reshape(-1))
    assert np.array_equal(np.array(out["cluster_bound"], dtype=np.float32), clusters.reshape(-1))
    assert np.array_equal(np.array(out["block_bound"], dtype=np.float32), blocks.reshape(-1))
    assert [int(v) for v in out["loose_spheres"]] == list(loose)
    used = slots[slots != 0xFFFF]
    assert sorted(list(used) + list(loose)) == list(range(geom.shape[0]))     # every sphere exactly once


@pytest.mark.parametrize("n", [65, 200, 4096])
def test_grouping_of_the_benchmark_scene_under_sanitizers(checker, tmp_path, n):
    geom = np.ascontiguousarray(flatten_scene(synthetic.many_spheres_scene(n, 64, 36)).sphere_geom)
    path = tmp_path / "geom.bin"
    geom.tofile(path)
    _compare_grouping(checker("hierfile", path, geom.shape[0]), geom)


@pytest.mark.parametrize("n,seed", [(0, 1), (1, 1), (2, 3), (9, 4), (64, 5), (513, 6), (5000, 7)])
def test_grouping_of_odd_sizes_under_sanitizers(checker, n, seed):
    out = checker("hier", n, seed)
    blocks = int(out["blocks"][0])
    ids = [int(v) for v in out["ids"]]
    loose = [int(v) for v in out.get("loose_spheres", [])]
    assert len(ids) == blocks * 64
    assert sorted([i for i in ids if i != 0xFFFF] + loose) == list(range(n))


@pytest.mark.parametrize("k", [0, 1, 16, 20, 33, 47])
def test_jump_polynomials_under_sanitizers(checker, k):
    out = checker("jump", k)
    assert np.array_equal(np.array(out["poly"], dtype=np.uint64).astype(np.uint32), _native.mt19937_jump_poly(k))


def test_jump_polynomial_range_under_sanitizers(checker):
    assert "error" in checker("jump", 48) and "error" in checker("jump", -1)


@pytest.mark.parametrize("n", [0, 1, 70, 1500])
def test_grid_and_light_map_under_sanitizers(checker, n):
    out = checker("grid", n, 11)
    dims = [int(v) for v in out["dims"]]
    if n <= 1:
        assert int(out["items"][0]) >= 0      # (a scene of one sphere: nothing to check but that it ran clean)
    else:
        assert min(dims) >= 1 and int(out["loose"][0]) == 1 and int(out["items"][0]) >= n - 1
        assert int(out["map_usable"][0]) == 1 and int(out["map_always"][0]) == 1 and int(out["map_items"][0]) >= n - 1
