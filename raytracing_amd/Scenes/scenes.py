"""Name -> scene table (reference Scenes/scenes.py:6-9) plus ``build_scene``
for rendering an example at another resolution."""
from . import example_1, example_2, example_3, example_4

_modules = {"example_1": example_1, "example_2": example_2,
            "example_3": example_3, "example_4": example_4}

scenes = {name: mod.scene for name, mod in _modules.items()}


def build_scene(name, width=None, height=None):
    """Fresh copy of a scene whose screen is ``width`` x ``height``: an example by name, or one of
    the synthetic benchmark scenes as ``"spot:MODE[:SAMPLES]"`` (spotlight with that easing mode,
    64 samples by default) or ``"many:SPHERES"`` (see :mod:`.synthetic`)."""
    if name.startswith("spot:"):
        from . import synthetic
        parts = name.split(":")
        return synthetic.spotlight_scene(parts[1], int(parts[2]) if len(parts) > 2 else 64, width, height)
    if name.startswith("many:"):
        from . import synthetic
        return synthetic.many_spheres_scene(int(name.split(":")[1]), width, height)
    return _modules[name].build(width, height)
