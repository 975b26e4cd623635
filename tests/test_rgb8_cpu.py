"""The output step of the reference's main() (plt.imsave): quantisation and PNG codec on the
host, and the oracle against the 8-bit goldens the reference produced through its own
output step (tests/golden/make_golden.py, generator "rgb8")."""
import os

import numpy as np
import pytest

from oracle import numpy_oracle
from raytracing_amd.imageio import decode_png, encode_png, quantise
from raytracing_amd.Scenes.scenes import build_scene


def test_png_round_trip():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, size=(37, 53, 3), dtype=np.uint8)
    back = decode_png(encode_png(img))
    assert back.shape == (37, 53, 4)
    assert (back[..., :3] == img).all() and (back[..., 3] == 255).all()


def test_png_rejects_wrong_input():
    with pytest.raises(ValueError):
        encode_png(np.zeros((4, 4, 3)))
    with pytest.raises(ValueError):
        decode_png(b"not a png")


def test_quantise_truncates_like_imsave():
    v = np.array([[-0.5, 0.0, 0.999 / 255], [1.0 / 255, 0.5, 254.999 / 255], [1.0, 1.5, 0.25]])
    assert quantise(v).tolist() == [[0, 0, 0], [1, 127, 254], [255, 255, 63]]


@pytest.mark.parametrize("name", ["example_1", "example_2"])
def test_oracle_matches_reference_rgb8(golden_dir, name):
    gold = np.load(os.path.join(golden_dir, "rgb8.npz"))[name]
    h, w, _ = gold.shape
    img, _ = numpy_oracle.render(build_scene(name, w, h), seed=0)
    assert (quantise(img.reshape(h, w, 3)) == gold).all()
