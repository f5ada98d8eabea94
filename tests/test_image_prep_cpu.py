"""CPU: the numpy restatement of Pillow's 8-bit BILINEAR resample (and the integer coefficient tables the device kernel uses)
against Pillow itself - the library the reference's torchvision resize calls for PIL images (data/transforms.py:14-25)."""
import numpy as np
import pytest
from PIL import Image

from labelanything_amd.image_prep import pil_bilinear_coeffs, resize_shape
from oracle import preprocess_oracle as PO

SIZES = [(150, 200, 168, 224), (224, 100, 224, 100), (480, 640, 768, 1024), (37, 53, 224, 321), (500, 333, 224, 149),
         (1200, 900, 1024, 768), (427, 640, 683, 1024), (3, 5, 224, 224), (64, 64, 64, 64), (2000, 1500, 224, 168)]


@pytest.mark.parametrize("h,w,nh,nw", SIZES)
def test_numpy_resample_equals_pillow(h, w, nh, nw):
    img = np.random.default_rng(h * 7 + w).integers(0, 256, (h, w, 3), dtype=np.uint8)
    ref = np.asarray(Image.fromarray(img).resize((nw, nh), Image.BILINEAR))
    assert np.array_equal(PO.resize_numpy(img, nh, nw), ref)


def test_coefficient_tables_are_normalised_and_in_range():
    for i, o in [(640, 1024), (1024, 640), (480, 480), (5, 224), (4000, 1024)]:
        b, k = pil_bilinear_coeffs(i, o)
        assert b.shape == (o, 2) and k.shape[0] == o
        assert (b[:, 0] >= 0).all() and (b[:, 0] + b[:, 1] <= i).all() and (b[:, 1] >= 1).all()
        assert np.abs(k.sum(axis=1) - (1 << 22)).max() <= k.shape[1]          # rounding of each tap only
        assert (k[np.arange(k.shape[1])[None, :] >= b[:, 1:2]] == 0).all()      # taps beyond the count are zero


def test_resize_shapes_follow_the_reference_rules():
    assert resize_shape(480, 640, 1024, True, False) == (768, 1024)             # CustomResize: longest side -> 1024
    assert resize_shape(640, 427, 1024, True, False) == (1024, 683)
    assert resize_shape(480, 640, 224, False, True) == (224, 224)               # Resize((S, S))
    assert resize_shape(480, 640, 1024, False, False) == (1024, 1365)           # Resize(S): short side -> S
    assert resize_shape(640, 480, 1024, False, False) == (1365, 1024)
