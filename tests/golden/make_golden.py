"""Generate the golden fixtures by running the REFERENCE implementation.

Run once in the development container (the reference checkout is read-only and
does not travel to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [--ref /root/reference] [--only NAME ...]

The script imports the reference's own modules from its checkout, patches
nothing in them, and records inputs and outputs only:

* ``scenes_flat.npz``   flattened parameters of the reference's example scenes
* ``images_160x90.npz`` example_1-4 images (seed 0), per-call traces, counters
* ``depths_96x54.npz``  example_2 under other depth limits
* ``spot_128x72.npz``   spotlight scene, five easing modes, 64 samples (seed 0)
* ``many_257_64x36.npz`` / ``many_4096_64x36.npz``  many-sphere scene
* ``strips_160x90.npz`` example_3 rendered in five row strips
* ``traces_500.npz``    per-call traces + checksums of example_1-4 at 500x500
* ``unit_kats.npz``     function-level input/output vectors
* ``rgb8.npz``          the 8-bit images the reference's output step (``plt.imsave``, main.py:32)
                        writes: example_1-3 at 1000x1000, example_4 at 500x500 (seed 0)
* ``disk_colour_48x27.npz``  example_4 with and without colours set on its disk light
* ``tir_160x90.npz``    scene with thousands of rays in the total-internal-reflection branch
* ``clip_160x90.npz``   scene (EPSILON = 1e-2) where the attenuation factor exceeds 1 and the
                        per-call clip of raytrace.py:170 is active below the top level; the largest
                        value entering that clip is recorded per recursion depth
* ``full_1080.npz``     example_1-4 at 1920x1080 (seed 0): pixels ``[::30, ::32]``, traces, sums
* ``full_4k.npz``       example_1-3 at 3840x2160: pixels ``[::60, ::64]``, traces, sums (example_4
                        needs ~78 GB in the reference at this size and is not generated)

The reference binds its image size at import time, so every configuration is
produced from a fresh import with ``constants.WIDTH/HEIGHT/SIZE`` set first.
"""
import argparse
import importlib
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)

REF_MODULES = ("constants", "materials", "objects", "raytrace", "Scenes", "Scenes.Scene",
               "Scenes.scenes", "Scenes.example_1", "Scenes.example_2", "Scenes.example_3",
               "Scenes.example_4")


class Ref:
    """A fresh import of the reference at a given image size / depth limits."""

    def __init__(self, ref_dir, width, height, refl_depth=3, refr_depth=5, epsilon=None):
        for name in REF_MODULES:
            sys.modules.pop(name, None)
        if ref_dir not in sys.path:
            sys.path.insert(0, ref_dir)
        constants = importlib.import_module("constants")
        constants.WIDTH, constants.HEIGHT, constants.SIZE = width, height, width * height
        constants.MAX_REFLECTION_DEPTH = refl_depth
        constants.MAX_REFRACTION_DEPTH = refr_depth
        if epsilon is not None:
            constants.EPSILON = epsilon
        self.constants = constants
        self.materials = importlib.import_module("materials")
        self.objects = importlib.import_module("objects")
        self.raytrace = importlib.import_module("raytrace")
        self.Scene = importlib.import_module("Scenes.Scene").Scene
        self.scenes = importlib.import_module("Scenes.scenes").scenes
        self._instrument()

    def _instrument(self):
        """Record every recursive call (depths, batch, hits, in-beam counts,
        branch path) without touching the reference's code."""
        rt, ob = self.raytrace, self.objects
        self.calls, self.shadow_rays, self.uniforms = [], 0, 0
        stack = []
        inner = rt.recursive_function
        inner_closest = ob.find_closest_intersected_object
        inner_shadow = ob.shadow_objects_multipliers
        inner_disk = ob.DiskSource.compute_light_intensity
        me = self

        def recursive_function(scene, starts, dirs, a, b):
            if stack:
                parent = stack[-1]
                # which branch: the reference recurses transmit (a, b-1), TIR (a-1, b-1), reflect (a-1, b)
                da, db = parent["a"] - a, parent["b"] - b
                sym = {(0, 1): "t", (1, 1): "i", (1, 0): "r"}[(da, db)]
                path = parent["path"] + sym
            else:
                path = ""
            rec = dict(a=a, b=b, rays=int(starts.shape[0]), hits=0, beam=[], path=path)
            me.calls.append(rec)
            stack.append(rec)
            try:
                return inner(scene, starts, dirs, a, b)
            finally:
                stack.pop()

        def find_closest(starts, dirs, objs):
            seen, dist = inner_closest(starts, dirs, objs)
            if stack:
                stack[-1]["hits"] = int((seen != -1).sum())
            return seen, dist

        def shadow(starts, dirs, objs):
            me.shadow_rays += int(dirs.shape[0])
            return inner_shadow(starts, dirs, objs)

        def disk_intensity(light, points, objs):
            if isinstance(light, ob.DirectionalDiskSource) and stack:
                stack[-1]["beam"].append(int(points.shape[0]))
            me.uniforms += 2 * light.n_points * int(points.shape[0])
            return inner_disk(light, points, objs)

        rt.recursive_function = recursive_function
        ob.find_closest_intersected_object = find_closest
        ob.shadow_objects_multipliers = shadow
        ob.DiskSource.compute_light_intensity = disk_intensity

    def reset_counters(self):
        self.calls, self.shadow_rays, self.uniforms = [], 0, 0

    def render(self, scene, seed=0):
        self.reset_counters()
        np.random.seed(seed)
        img = self.raytrace.raytrace(scene)
        return img

    def trace_arrays(self, prefix):
        c = self.calls
        nbeam = max([len(r["beam"]) for r in c] + [0])
        beam = np.zeros((len(c), max(nbeam, 1)), dtype=np.int64)
        for i, r in enumerate(c):
            beam[i, :len(r["beam"])] = r["beam"]
        return {
            prefix + "_a": np.array([r["a"] for r in c], dtype=np.int32),
            prefix + "_b": np.array([r["b"] for r in c], dtype=np.int32),
            prefix + "_rays": np.array([r["rays"] for r in c], dtype=np.int64),
            prefix + "_hits": np.array([r["hits"] for r in c], dtype=np.int64),
            prefix + "_beam": beam,
            prefix + "_path": np.array([r["path"] for r in c]),
            prefix + "_shadow_rays": np.int64(self.shadow_rays),
            prefix + "_uniforms": np.int64(self.uniforms),
        }


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.0f} KiB", flush=True)


def gen_scenes_flat(ref_dir):
    from raytracing_amd.flatten import flatten_scene
    ref = Ref(ref_dir, 160, 90)
    out = {}
    for name, scene in ref.scenes.items():
        f = flatten_scene(scene)
        out[name + "_geom"] = f.sphere_geom
        out[name + "_mat"] = f.sphere_mat
        out[name + "_ambient"] = f.ambient
        out[name + "_camera"] = f.camera_position
        out[name + "_screen"] = np.concatenate([f.screen_position, f.screen_x, f.screen_y,
                                                [f.screen_width, f.screen_height, f.pixels_x, f.pixels_y]])
        l = f.lights[0]
        out[name + "_light"] = np.concatenate([[l.kind], l.position, [l.intensity, l.radius, l.n_points]])
    save("scenes_flat.npz", **out)


def gen_images(ref_dir):
    ref = Ref(ref_dir, 160, 90)
    out = {}
    for name, scene in ref.scenes.items():
        t0 = time.time()
        out[name + "_image"] = ref.render(scene, seed=0)
        out.update(ref.trace_arrays(name))
        print(name, f"{time.time() - t0:.1f}s sum={out[name + '_image'].sum():.9f}", flush=True)
    save("images_160x90.npz", **out)


def gen_depths(ref_dir):
    out = {}
    for (a, b) in [(0, 0), (1, 0), (0, 2), (2, 2), (1, 3)]:
        ref = Ref(ref_dir, 96, 54, a, b)
        for name in ("example_2", "example_3"):
            key = f"{name}_{a}_{b}"
            out[key + "_image"] = ref.render(ref.scenes[name], seed=0)
            out.update(ref.trace_arrays(key))
    save("depths_96x54.npz", **out)


def gen_spot(ref_dir):
    from raytracing_amd.Scenes import synthetic
    out = {}
    for mode in ("none", "linear", "quadratic", "cubic", "exponential"):
        ref = Ref(ref_dir, 128, 72)
        scene = synthetic.spotlight_scene(mode, 64, _objects=ref.objects, _materials=ref.materials,
                                          _scene_cls=ref.Scene)
        out[mode + "_image"] = ref.render(scene, seed=0)
        out.update(ref.trace_arrays(mode))
        print("spot", mode, out[mode + "_image"].mean(axis=(0, 1)), flush=True)
    save("spot_128x72.npz", **out)


def gen_many(ref_dir, n_spheres):
    from raytracing_amd.Scenes import synthetic
    ref = Ref(ref_dir, 64, 36)
    scene = synthetic.many_spheres_scene(n_spheres, _objects=ref.objects, _materials=ref.materials,
                                         _scene_cls=ref.Scene)
    t0 = time.time()
    out = {"image": ref.render(scene, seed=0)}
    out.update(ref.trace_arrays("trace"))
    print("many", n_spheres, f"{time.time() - t0:.1f}s mean", out["image"].mean(axis=(0, 1)), flush=True)
    save(f"many_{n_spheres}_64x36.npz", **out)


def gen_strips(ref_dir):
    ref = Ref(ref_dir, 160, 90)
    out = {}
    for name in ("example_1", "example_3"):
        scene = ref.scenes[name]
        rows = []
        bounds = [0, 18, 36, 54, 72, 90]
        for r0, r1 in zip(bounds[:-1], bounds[1:]):
            Y, X = np.indices((r1 - r0, 160), dtype=float)
            Y += r0
            np.random.seed(0)
            rows.append(ref.raytrace.get_pixel_color(X.reshape(-1, 1), Y.reshape(-1, 1), scene).reshape(r1 - r0, 160, 3))
        out[name + "_strips"] = np.concatenate(rows, axis=0)
        out[name + "_bounds"] = np.array(bounds)
    save("strips_160x90.npz", **out)


def gen_traces_500(ref_dir):
    ref = Ref(ref_dir, 500, 500)
    out = {}
    for name, scene in ref.scenes.items():
        t0 = time.time()
        img = ref.render(scene, seed=0)
        out.update(ref.trace_arrays(name))
        out[name + "_sum"] = img.sum()
        out[name + "_channel_mean"] = img.mean(axis=(0, 1))
        out[name + "_rows"] = img[::50, ::25].copy()
        print(name, "500x500", f"{time.time() - t0:.1f}s", flush=True)
    save("traces_500.npz", **out)


def gen_unit(ref_dir):
    ref = Ref(ref_dir, 160, 90)
    ob, rt = ref.objects, ref.raytrace
    rs = np.random.RandomState(12345)
    out = {}

    # quadratic roots, both modes
    B = rs.uniform(-6, 6, 4000)
    C = rs.uniform(-6, 6, 4000)
    B[:4] = [0.0, 2.0, -2.0, 0.0]
    C[:4] = [0.0, 1.0, 1.0, -1.0]
    out["quad_B"], out["quad_C"] = B, C
    out["quad_closest"] = ob.solve_quadratic(B.copy(), C.copy(), "closest")
    out["quad_furthest"] = ob.solve_quadratic(B.copy(), C.copy(), "furthest")

    # sphere intersection on every example sphere (incl. the huge grounds)
    geom, o_list, d_list, cl, fu = [], [], [], [], []
    for name, scene in ref.scenes.items():
        for s in scene.objects:
            o = scene.camera.position + rs.uniform(-2, 2, (256, 3))
            o[:, 2] = np.abs(o[:, 2]) + 1e-3
            d = rs.normal(size=(256, 3))
            d[:128, 2] = -np.abs(d[:128, 2])
            d /= np.linalg.norm(d, axis=-1, keepdims=True)
            geom.append(np.concatenate([s.position, [s.radius]]))
            o_list.append(o); d_list.append(d)
            cl.append(s.intersection(o, d)); fu.append(s.intersection(o, d, "furthest"))
    out["isect_geom"] = np.array(geom)
    out["isect_o"], out["isect_d"] = np.array(o_list), np.array(d_list)
    out["isect_closest"], out["isect_furthest"] = np.array(cl), np.array(fu)

    # nearest hit incl. an exact tie (two identical spheres: the later one wins)
    spheres = [ob.Sphere(4, 0, 1, 1.0), ob.Sphere(4, 0, 1, 1.0), ob.Sphere(6, 1, 1, 0.5), ob.Sphere(2, -1, 0.5, 0.5)]
    o = np.tile(np.array([0.0, 0.0, 1.0]), (512, 1))
    d = np.concatenate([np.ones((512, 1)), rs.uniform(-0.5, 0.5, (512, 2))], axis=1)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    seen, tt = ob.find_closest_intersected_object(o, d, spheres)
    out["near_geom"] = np.array([np.concatenate([s.position, [s.radius]]) for s in spheres])
    out["near_o"], out["near_d"], out["near_id"], out["near_t"] = o, d, seen, tt[:, 0]

    # shadow multipliers through the example_2 spheres (two of them transparent)
    sc = ref.scenes["example_2"]
    o = rs.uniform(-1, 6, (1024, 3)); o[:, 2] = rs.uniform(0.01, 3, 1024)
    d = np.array([4.0, 0.0, 10.0]) - o
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    out["shadow_o"], out["shadow_d"] = o, d
    out["shadow_mult"] = ob.shadow_objects_multipliers(o, d, sc.objects)

    # refraction / Fresnel / reflection incl. total internal reflection and n1 == n2
    nrm = rs.normal(size=(2048, 3)); nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    inc = rs.normal(size=(2048, 3)); inc /= np.linalg.norm(inc, axis=-1, keepdims=True)
    flip = np.sum(nrm * inc, axis=-1) < 0
    nrm[flip] *= -1
    n1 = rs.choice([1.0, 1.05, 1.5, 2.4], 2048)
    n2 = rs.choice([1.0, 1.05, 1.5], 2048)
    tv, tmask = rt.refract_vectors(nrm, inc, n1, n2)
    tfull = np.zeros((2048, 3)); tfull[tmask] = tv
    out["opt_normal"], out["opt_incident"], out["opt_n1"], out["opt_n2"] = nrm, inc, n1, n2
    out["opt_transmitted"], out["opt_mask"] = tfull, tmask
    out["opt_fresnel"] = rt.fresnel_multiplier(inc, tfull, nrm, n1, n2)
    out["opt_reflect"] = rt.reflect_vectors(inc, -nrm)

    # easing functions on a grid (a, d > 0) and the d == 0 guard of ease_fall_off_beam
    x = rs.uniform(0, 4, 3000); a = rs.uniform(0.2, 2, 3000); dd = rs.uniform(0.05, 2, 3000)
    x[:6] = [0.5, 1, 1.2, 1.5, 2, 2.5]; a[:6] = 1; dd[:6] = 1
    out["ease_x"], out["ease_a"], out["ease_d"] = x, a, dd
    out["ease_linear"] = ob.linear_easing(x, a, dd)
    out["ease_quadratic"] = ob.quadratic_easing(x, a, dd)
    out["ease_cubic"] = ob.cubic_easing(x, a, dd)
    out["ease_exponential"] = ob.exponential_easing(x, a, dd)
    dz = dd.copy(); dz[::7] = 0.0
    for mode in ("none", "linear", "quadratic", "cubic", "exponential"):
        light = ob.DirectionalDiskSource(easing_mode=mode)
        out["beam_" + mode] = light.ease_fall_off_beam(x, a, dz)
    out["ease_dz"] = dz

    # screen mapping / camera frames of the four example cameras
    frames, pix = [], []
    X = rs.uniform(0, 160, (64, 1)); Y = rs.uniform(0, 90, (64, 1))
    for name, scene in ref.scenes.items():
        cam, scr = scene.camera, scene.screen
        frames.append(np.concatenate([cam.position, cam.viewing_direction, cam.y_vector, scr.position,
                                      scr.x_vector, scr.y_vector, [scr.width, scr.height]]))
        pix.append(scr.index_to_position(X, Y))
    out["screen_X"], out["screen_Y"] = X, Y
    out["screen_frames"], out["screen_points"] = np.array(frames), np.array(pix)
    save("unit_kats.npz", **out)


def gen_rgb8(ref_dir):
    """The reference's render followed by its own output step: plt.imsave into memory, then the
    PNG's pixels (decoded with our zlib-only decoder)."""
    import io
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    from raytracing_amd.imageio import decode_png
    out = {}
    for names, size in ((("example_1", "example_2", "example_3"), 1000), (("example_4",), 500)):
        ref = Ref(ref_dir, size, size)
        for name in names:
            t0 = time.time()
            img = ref.render(ref.scenes[name], seed=0)
            buf = io.BytesIO()
            plt.imsave(buf, img, format="png")
            px = decode_png(buf.getvalue())
            assert px.shape == (size, size, 4) and (px[..., 3] == 255).all()
            out[name] = np.ascontiguousarray(px[..., :3])
            print(name, size, f"{time.time() - t0:.1f}s", flush=True)
    save("rgb8.npz", **out)


def gen_disk_colour(ref_dir):
    """A DiskSource whose colour attributes are set: the reference ignores them (its samples go
    through a fresh white PointSource, objects.py:140-144)."""
    ref = Ref(ref_dir, 48, 27)
    scene = ref.scenes["example_4"]
    plain = ref.render(scene, seed=0)
    scene.light_sources[0].diffuse_color = np.array([1.0, 0.0, 0.0])
    scene.light_sources[0].specular_color = np.array([0.0, 0.0, 1.0])
    coloured = ref.render(scene, seed=0)
    print("disk colour: max difference", np.abs(plain - coloured).max(), flush=True)
    save("disk_colour_48x27.npz", plain=plain, coloured=coloured)


class ClipSpy:
    """Stands in for the ``np`` name inside the reference's raytrace module: every attribute is
    NumPy's, except that ``clip`` also records, per recursion depth, the largest value handed to
    the clip that ends ``recursive_function`` (reference raytrace.py:170)."""

    def __init__(self, ref):
        self._ref = ref
        self.max_by_depth = {}

    def __getattr__(self, name):
        return getattr(np, name)

    def clip(self, a, lo, hi, *args, **kwargs):
        frame = sys._getframe(1)
        if frame.f_code.co_name == "recursive_function" and frame.f_lineno == 170:
            # the call this clip ends is identified by the depth limits it was given
            a_, b_ = frame.f_locals["reflection_depth"], frame.f_locals["refraction_depth"]
            depth = (3 - a_, 5 - b_)
            self.max_by_depth[depth] = max(self.max_by_depth.get(depth, -np.inf), float(np.max(a)))
        return np.clip(a, lo, hi, *args, **kwargs)


def gen_tir(ref_dir):
    from raytracing_amd.Scenes import synthetic
    ref = Ref(ref_dir, 160, 90)
    scene = synthetic.tir_scene(_objects=ref.objects, _materials=ref.materials, _scene_cls=ref.Scene)
    out = {"image": ref.render(scene, seed=0)}
    out.update(ref.trace_arrays("trace"))
    tir = [(r["path"], r["rays"]) for r in ref.calls if r["path"].endswith("i") and r["rays"] >= 3]
    print("tir: rays in total-internal-reflection calls with batch >= 3:", sum(n for _, n in tir), "in", len(tir), "calls", flush=True)
    save("tir_160x90.npz", **out)


def gen_clip(ref_dir):
    from raytracing_amd.Scenes import synthetic
    ref = Ref(ref_dir, 160, 90, epsilon=1e-2)
    spy = ClipSpy(ref)
    ref.raytrace.np = spy
    scene = synthetic.clip_scene(_objects=ref.objects, _materials=ref.materials, _scene_cls=ref.Scene)
    out = {"image": ref.render(scene, seed=0), "epsilon": np.float64(1e-2)}
    out.update(ref.trace_arrays("trace"))
    keys = sorted(spy.max_by_depth)
    out["preclip_depth_used"] = np.array(keys, dtype=np.int32)          # (reflections used, refractions used)
    out["preclip_max"] = np.array([spy.max_by_depth[k] for k in keys])
    print("clip: largest value entering the clip of raytrace.py:170, by (reflections, refractions) used:", flush=True)
    for k in keys:
        print("   ", k, f"{spy.max_by_depth[k]:.4f}", flush=True)
    save("clip_160x90.npz", **out)


def _gen_full(ref_dir, width, height, names, step_y, step_x, fname):
    path = os.path.join(HERE, fname)
    out = dict(np.load(path)) if os.path.exists(path) else {}
    for name in names:
        ref = Ref(ref_dir, width, height)
        t0 = time.time()
        img = ref.render(ref.scenes[name], seed=0)
        out.update(ref.trace_arrays(name))
        out[name + "_sum"] = img.sum()
        out[name + "_channel_mean"] = img.mean(axis=(0, 1))
        out[name + "_pixels"] = img[::step_y, ::step_x].copy()
        out[name + "_seconds"] = np.float64(time.time() - t0)
        print(name, f"{width}x{height}", f"{time.time() - t0:.1f}s", flush=True)
        del img, ref
        save(fname, **out)     # after every scene: the long runs may be cut short
    out["steps"] = np.array([step_y, step_x])
    save(fname, **out)


def gen_full_1080(ref_dir, names=("example_1", "example_2", "example_3", "example_4")):
    _gen_full(ref_dir, 1920, 1080, names, 30, 32, "full_1080.npz")


def gen_full_4k(ref_dir, names=("example_1", "example_2", "example_3")):
    _gen_full(ref_dir, 3840, 2160, names, 60, 64, "full_4k.npz")


GENERATORS = {
    "scenes_flat": gen_scenes_flat, "images": gen_images, "depths": gen_depths, "spot": gen_spot,
    "many_257": lambda r: gen_many(r, 257), "many_4096": lambda r: gen_many(r, 4096),
    "strips": gen_strips, "traces_500": gen_traces_500, "unit": gen_unit, "rgb8": gen_rgb8,
    "disk_colour": gen_disk_colour, "tir": gen_tir, "clip": gen_clip, "full_1080": gen_full_1080, "full_4k": gen_full_4k,
}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--only", nargs="*", default=None)
    args = ap.parse_args()
    for key, fn in GENERATORS.items():
        if args.only and key not in args.only:
            continue
        t0 = time.time()
        fn(args.ref)
        print(f"[{key}] done in {time.time() - t0:.1f}s", flush=True)
