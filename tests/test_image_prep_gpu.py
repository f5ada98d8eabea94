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


@pytest.mark.parametrize("custom", [True, False])
@pytest.mark.parametrize("hw", [(480, 640), (640, 427), (333, 500), (1024, 1024), (37, 53)])
def test_prompt_masks_match_reference_apply_masks(custom, hw):
    """OR of instance masks -> nearest resize -> pad -> nearest resize to 256 (data/transforms.py:203-224) + flag_masks."""
    from labelanything_amd.prompts import prompt_masks_from_instances
    rng = np.random.default_rng(hw[0] + custom)
    n = 5
    inst = np.zeros((n, *hw), dtype=np.uint8)
    for i in range(n - 1):                                    # rectangles + sparse noise; the last instance stays empty
        y0, x0 = rng.integers(0, hw[0] - 8), rng.integers(0, hw[1] - 8)
        inst[i, y0:y0 + rng.integers(4, hw[0] - y0), x0:x0 + rng.integers(4, hw[1] - x0)] = 1
        inst[i] |= (rng.random(hw) < 0.01).astype(np.uint8)
    slots = [[0], [1, 2, 3], [], [4], [0, 4]]
    got, flags = prompt_masks_from_instances(torch.from_numpy(inst).cuda(), slots, side=512, mask_side=256, custom_preprocess=custom)
    for c, s in enumerate(slots):
        ref = PO.reference_apply_masks([inst[i] for i in s], side=512, mask_side=256, custom_preprocess=custom).reshape(256, 256)
        assert torch.equal(got[c].cpu(), ref.float()), f"slot {c}"
        assert int(flags[c]) == int(ref.sum() > 0)
    assert got.dtype == torch.float32 and flags.tolist() == [1, 1, 0, 0, 1]


def test_coords_boxes_and_flags_merge():
    from labelanything_amd.prompts import apply_boxes, apply_coords, flags_merge
    pts = torch.tensor([[[10.0, 20.0], [639.0, 479.0]]])
    out = apply_coords(pts, (480, 640), side=1024, custom_preprocess=True)
    assert torch.allclose(out, torch.tensor([[[16.0, 32.0], [1022.4, 766.4]]]))
    assert torch.allclose(apply_boxes(torch.tensor([[0.0, 0.0, 640.0, 480.0]]), (480, 640), 1024, False), torch.tensor([[0.0, 0.0, 1024.0, 1024.0]]))
    fm = torch.tensor([[0, 1, 0], [0, 0, 0]], dtype=torch.uint8)
    fp = torch.tensor([[[0, 0], [0, 0], [1, 0]], [[0, 0], [0, 0], [0, 0]]], dtype=torch.uint8)
    assert flags_merge(fm, fp).tolist() == [[True, True, True], [True, False, False]]      # background column forced to 1
    with pytest.raises(ValueError):
        flags_merge()


def test_annotations_to_tensor_rasterises_mask_prompts_on_the_device():
    """collate.annotations_to_tensor(prompt_type="mask"): per image a dict {category: uint8 [k, H, W] instance masks}; the union of a
    category's instances goes through the same device rasteriser as prompt_masks_from_instances (bit-exact with the reference's
    PromptsProcessor.apply_masks, tested above) and lands in the (N, C, 256, 256) tensor with its flag."""
    from labelanything_amd.collate import annotations_to_tensor
    from labelanything_amd.prompts import prompt_masks_from_instances
    g = torch.Generator().manual_seed(11)
    sizes = [(120, 200), (333, 150)]
    anns = []
    for (h, w) in sizes:
        anns.append({7: (torch.rand(2, h, w, generator=g) > 0.7).to(torch.uint8).numpy(), 3: (torch.rand(1, h, w, generator=g) > 0.5).to(torch.uint8).numpy()})
    anns[1][3] = np.zeros((0, *sizes[1]), dtype=np.uint8)    # the second image has no instance of category 3
    t, f = annotations_to_tensor(anns, sizes, "mask", device=torch.device("cuda"))
    assert t.shape == (2, 2, 256, 256) and t.dtype == torch.float32 and f.shape == (2, 2) and f.dtype == torch.uint8
    for i, a in enumerate(anns):
        for j, c in enumerate(a):                            # class slots follow the dict order
            if a[c].shape[0] == 0:
                assert int(f[i, j]) == 0 and float(t[i, j].abs().max()) == 0.0
                continue
            inst = torch.from_numpy(a[c]).cuda()
            ref, rf = prompt_masks_from_instances(inst, [list(range(inst.shape[0]))], 1024, 256, True)
            assert int(f[i, j]) == int(rf[0]) and torch.equal(t[i, j], ref[0])
    with pytest.raises(RuntimeError, match="device"):
        annotations_to_tensor(anns, sizes, "mask")
