"""Name -> scene table (reference Scenes/scenes.py:6-9) plus ``build_scene``
for rendering an example at another resolution."""
from . import example_1, example_2, example_3, example_4

_modules = {"example_1": example_1, "example_2": example_2,
            "example_3": example_3, "example_4": example_4}

scenes = {name: mod.scene for name, mod in _modules.items()}


def build_scene(name, width=None, height=None):
    """Fresh copy of an example scene whose screen is ``width`` x ``height``."""
    return _modules[name].build(width, height)
