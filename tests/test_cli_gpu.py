"""GPU: the generate_embeddings CLI surface (preprocess.py:53-246) - file format, naming and values."""
import json
import os

import numpy as np
import pytest
import torch
from click.testing import CliRunner
from PIL import Image
from safetensors.torch import load_file, save_file

from labelanything_amd.config import ENCODER_SPECS, LamConfig
from labelanything_amd.weights import init_state_dict
from oracle import lam_oracle as O
from tests.cases import geometry_for
from tests.helpers import rel_err

pytestmark = pytest.mark.gpu


def _images(d, sizes):
    rng = np.random.default_rng(0)
    os.makedirs(d, exist_ok=True)
    for i, (h, w) in enumerate(sizes):
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(os.path.join(d, f"{i:012d}.png"))


def test_generate_embeddings_sam_style(tmp_path):
    from label_anything.cli import main
    from label_anything.preprocess import load_image, IMAGENET_DEFAULT
    cfg = LamConfig(encoder="sam_tiny", image_size=224)
    sd = init_state_dict(cfg, 5)
    enc = {k[len("image_encoder."):]: v for k, v in sd.items() if k.startswith("image_encoder.")}
    ckpt = str(tmp_path / "enc.safetensors")
    save_file(enc, ckpt)
    imgs, out, last = str(tmp_path / "imgs"), str(tmp_path / "emb"), str(tmp_path / "last")
    _images(imgs, [(150, 200), (224, 100)])
    r = CliRunner().invoke(main, ["generate_embeddings", "--encoder", "sam_tiny", "--checkpoint", ckpt, "--directory", imgs,
                                  "--outfolder", out, "--last_block_dir", last, "--custom_preprocess", "--batch_size", "2"])
    assert r.exit_code == 0, r.output + str(r.exception)
    assert sorted(os.listdir(out)) == ["000000000000.safetensors", "000000000001.safetensors"]
    geo = geometry_for(cfg)
    for i in range(2):
        x = load_image(os.path.join(imgs, f"{i:012d}.png"), 224, True, *IMAGENET_DEFAULT, square=False).unsqueeze(0)
        with torch.no_grad():
            ref, ref_last = O.sam_encoder(sd, geo, x, return_last_block=True)
        got = load_file(os.path.join(out, f"{i:012d}.safetensors"))["embedding"]
        got_last = load_file(os.path.join(last, f"{i:012d}.safetensors"))["embedding"]
        assert got.shape == (96, 14, 14) and got.dtype == torch.float32 and got_last.shape == (128, 14, 14)
        assert rel_err(got, ref[0]) < 3e-3 and rel_err(got_last, ref_last[0]) < 3e-3


def test_generate_embeddings_huggingface_dir(tmp_path):
    from label_anything.cli import main
    from label_anything.preprocess import load_image, IMAGENET_DEFAULT
    cfg = LamConfig(encoder="hf_tiny", image_size=240, image_embed_dim=128)
    sd = init_state_dict(cfg, 6)
    model_dir = tmp_path / "vit"
    os.makedirs(model_dir)
    spec = ENCODER_SPECS["hf_tiny"]
    json.dump({"hidden_size": spec.dim, "num_hidden_layers": spec.depth, "num_attention_heads": spec.heads,
               "intermediate_size": spec.mlp, "patch_size": 16, "image_size": 224}, open(model_dir / "config.json", "w"))
    save_file({"vit." + k[len("image_encoder."):]: v for k, v in sd.items() if k.startswith("image_encoder.")},
              str(model_dir / "model.safetensors"))
    imgs, out = str(tmp_path / "imgs"), str(tmp_path / "emb")
    _images(imgs, [(300, 200)])
    r = CliRunner().invoke(main, ["generate_embeddings", "--huggingface", "--model_name", str(model_dir), "--directory", imgs,
                                  "--outfolder", out, "--image_resolution", "240"])
    assert r.exit_code == 0, r.output + str(r.exception)
    x = load_image(os.path.join(imgs, "000000000000.png"), 240, False, *IMAGENET_DEFAULT, square=True).unsqueeze(0)
    with torch.no_grad():
        ref = O.hf_vit_encoder(sd, geometry_for(cfg), x)
    got = load_file(os.path.join(out, "000000000000.safetensors"))["embedding"]
    assert got.shape == (128, 15, 15)
    assert rel_err(got, ref[0]) < 3e-3
