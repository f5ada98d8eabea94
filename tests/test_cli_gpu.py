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


def test_embedding_cache_round_trip_and_prototype_serving(tmp_path):
    """generate_embeddings -> files -> load_episode_embeddings -> Lam(embeddings) equals Lam(images) on the same pictures
    (data/coco.py:251-275 wire format), and set_class_embeddings + predict serves the query from cached prototypes
    (experiment/utils.py:210-249, lam.py:362-381)."""
    from label_anything.cli import main
    from label_anything.preprocess import load_image, IMAGENET_DEFAULT
    from labelanything_amd.cache import embedding_path, load_embedding, load_episode_embeddings, set_class_embeddings
    from labelanything_amd.episodes import make_episode
    from labelanything_amd.models import Lam
    # no LAM neck (image_embed_dim == embed_dim): generate_class_embeddings / predict take cached embeddings as they are
    from labelanything_amd.config import EncoderSpec, register_encoder
    register_encoder("sam_tiny64", EncoderSpec("sam", dim=128, depth=2, heads=2, mlp=512, img_size=224, global_idx=(1,), window=8,
                                               out_chans=64))
    cfg = LamConfig(encoder="sam_tiny64", image_size=224, image_embed_dim=64, embed_dim=64, spatial_convs=3, custom_preprocess=True)
    sd = init_state_dict(cfg, 9)
    ckpt = str(tmp_path / "enc.safetensors")
    save_file({k[len("image_encoder."):]: v for k, v in sd.items() if k.startswith("image_encoder.")}, ckpt)
    imgs, out = str(tmp_path / "imgs"), str(tmp_path / "emb")
    sizes = [(150, 200), (224, 100), (180, 180)]
    _images(imgs, sizes)
    r = CliRunner().invoke(main, ["generate_embeddings", "--encoder", "sam_tiny64", "--checkpoint", ckpt, "--directory", imgs,
                                  "--outfolder", out, "--custom_preprocess", "--batch_size", "3"])
    assert r.exit_code == 0, r.output + str(r.exception)
    assert os.path.basename(embedding_path(out, 2)) == "000000000002.safetensors"
    emb0, gt0 = load_embedding(out, 0)
    assert emb0.shape == (64, 14, 14) and emb0.dtype == torch.float32 and gt0 is None
    with pytest.raises(FileNotFoundError):
        load_embedding(out, 7)

    # 1-way 2-shot episode: query = image 0, supports = images 1, 2
    emb = load_episode_embeddings(out, [[0, 1, 2]])
    assert emb.shape == (1, 3, 64, 14, 14)
    batch = make_episode(batch=1, n_ways=1, k_shots=2, image_size=224, seed=3, prompts=("mask", "point"))
    batch["dims"] = torch.tensor([[list(s) for s in sizes]])
    pics = torch.stack([load_image(os.path.join(imgs, f"{i:012d}.png"), 224, True, *IMAGENET_DEFAULT, square=False) for i in range(3)])
    lam = Lam(cfg, seed=9).cuda()
    from_images = lam({**batch, "images": pics.unsqueeze(0)})["logits"]
    b2 = {k: v for k, v in batch.items() if k != "images"}
    from_cache = lam({**b2, "embeddings": emb})["logits"]
    torch.cuda.synchronize()
    assert from_cache.shape == from_images.shape == (1, 2, 224, 200)      # max over ALL original sizes, supports included (lam.py:401)
    assert rel_err(from_cache, from_images) < 1e-6                      # same fp32 embeddings -> same decoder inputs

    # prototype cache: supports encoded once (no batch axis, like the reference passes them), then query-only predict
    examples = {k: v[0, 1:] if k in ("dims",) else v[0] for k, v in b2.items() if isinstance(v, torch.Tensor)}
    examples["embeddings"] = emb[0, 1:]
    set_class_embeddings(lam, examples)
    assert lam.class_embeddings["class_embeddings"].shape == (1, 2, 64)
    pred = lam.predict({"embeddings": emb[:, :1], "dims": batch["dims"][:, 0]})
    torch.cuda.synchronize()
    assert rel_err(pred, from_images[..., :150, :200]) < 1e-5
    # a model WITH a LAM neck must refuse pre-neck embeddings on this path instead of mis-shaping them
    necked = Lam(LamConfig(encoder="sam_tiny64", image_size=224, image_embed_dim=64, embed_dim=32, spatial_convs=3), seed=1).cuda()
    necked.class_embeddings = {"class_embeddings": torch.zeros(1, 2, 32).cuda()}
    with pytest.raises(ValueError, match="without the neck"):
        necked.predict({"embeddings": emb[:, :1], "dims": batch["dims"][:, 0]})
