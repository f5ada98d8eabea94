"""GPU: device image preprocessing (la_resample_u8, la_u8_to_chw_norm) against Pillow / the reference transform chain -
bit-exact for the uint8 resample, bit-exact for the normalised fp32 tensor."""
import numpy as np
import pytest
import torch
from PIL import Image

from labelanything_amd.image_prep import DevicePreprocessor
from oracle import preprocess_oracle as PO

pytestmark = pytest.mark.gpu
DEFAULT = ([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])


@pytest.mark.parametrize("h,w,nh,nw", [(150, 200, 168, 224), (480, 640, 768, 1024), (37, 53, 224, 321), (500, 333, 224, 149),
                                         (1200, 900, 1024, 768), (427, 640, 683, 1024), (64, 64, 64, 64), (2000, 1500, 224, 168)])
def test_device_resample_is_bit_exact_with_pillow(h, w, nh, nw):
    img = np.random.default_rng(h + 3 * w).integers(0, 256, (h, w, 3), dtype=np.uint8)
    ref = np.asarray(Image.fromarray(img).resize((nw, nh), Image.BILINEAR))
    dp = DevicePreprocessor(1024, True, *DEFAULT, square=False)
    got = dp.resize_u8(torch.from_numpy(img).cuda(), nh, nw).cpu().numpy()
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("mode", ["custom", "square", "short_side"])
@pytest.mark.parametrize("hw", [(480, 640), (640, 427), (333, 500), (1024, 1024)])
def test_device_preprocess_matches_reference_chain_bit_for_bit(mode, hw):
    """CustomResize -> ToTensor -> CustomNormalize (pad)  |  Resize((S,S)) -> ToTensor -> Normalize  |  Resize(S) -> ... :
    the three preprocessing variants of the generate_embeddings CLI (preprocess.py:109-121,240-246)."""
    side = 512
    custom, square = mode == "custom", mode == "square"
    img = np.random.default_rng(hw[0]).integers(0, 256, (*hw, 3), dtype=np.uint8)
    ref = PO.reference_preprocess(img, side, custom, *DEFAULT, square)
    got = DevicePreprocessor(side, custom, *DEFAULT, square=square)(torch.from_numpy(img)).cpu()
    assert got.shape == ref.shape and got.dtype == torch.float32
    assert torch.equal(got, ref)
    if custom:
        assert got.shape == (3, side, side) and float(got[:, -1, -1].abs().max()) == 0.0 or hw[0] == hw[1]


def test_bad_inputs_raise():
    dp = DevicePreprocessor(224, True, *DEFAULT, square=False)
    with pytest.raises(ValueError):
        dp(torch.zeros(10, 10, 3))                       # not uint8
    with pytest.raises(ValueError):
        dp(torch.zeros(10, 10, 4, dtype=torch.uint8))    # not RGB
