"""Offline image-embedding extraction: the ``generate_embeddings`` surface of the reference
(/root/reference/label_anything/preprocess.py:53-246, cli.py:54-175) on the MI355X encoder kernels.

Output format is the reference's: one ``<image stem>.safetensors`` per image holding ``{"embedding": (C, g, g) fp32}``
(consumer: data/coco.py:251-275); with ``--last_block_dir`` the pre-neck block state goes to a second directory.
Image decoding / resizing stays on the host with PIL, as in the reference (torchvision's PIL path: antialiased bilinear).
"""
from __future__ import annotations

import json
import logging
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
from PIL import Image
from safetensors.torch import load_file, save_file

from labelanything_amd.config import ENCODER_SPECS, EncoderSpec, LamConfig, register_encoder
from labelanything_amd.image_prep import DevicePreprocessor
from labelanything_amd.models import Lam, _hf5_to_hf4

IMAGENET_DEFAULT = ([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])      # data/utils.py "default"
IMAGENET_STANDARD = ([0.5, 0.5, 0.5], [0.5, 0.5, 0.5])                 # data/utils.py "standard"


def get_mean_std(name: str):
    return {"default": IMAGENET_DEFAULT, "standard": IMAGENET_STANDARD}[name]


def preprocess_shape(h: int, w: int, side: int) -> Tuple[int, int]:
    """data/utils.py:441-449."""
    s = side * 1.0 / max(h, w)
    return int(h * s + 0.5), int(w * s + 0.5)


def load_image(path: str, side: int, custom_preprocess: bool, mean, std, square: bool) -> torch.Tensor:
    """CustomResize -> ToTensor -> CustomNormalize (pad to side x side)   [custom_preprocess]
    Resize((side, side)) -> ToTensor -> Normalize                        [square, HF branch preprocess.py:240-246]
    Resize(side) (short side) -> ToTensor -> Normalize                    [SAM branch without custom_preprocess, :119]"""
    img = Image.open(path).convert("RGB")
    w, h = img.size
    if custom_preprocess:
        nh, nw = preprocess_shape(h, w, side)
    elif square:
        nh, nw = side, side
    else:
        if h <= w:
            nh, nw = side, int(side * w / h)
        else:
            nh, nw = int(side * h / w), side
    img = img.resize((nw, nh), Image.BILINEAR)
    x = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float() / 255.0
    x = (x - torch.tensor(mean).view(3, 1, 1)) / torch.tensor(std).view(3, 1, 1)
    if custom_preprocess:
        x = torch.nn.functional.pad(x, (0, side - nw, 0, side - nh))
    return x


def list_images(directory: str) -> List[str]:
    return sorted(f for f in os.listdir(directory) if os.path.isfile(os.path.join(directory, f)))


def _batches(files: Sequence[str], n: int):
    for i in range(0, len(files), n):
        yield files[i:i + n]


@torch.no_grad()
def _run(lam: Lam, directory: str, outfolder: str, last_block_dir: Optional[str], batch_size: int, side: int,
         custom_preprocess: bool, mean, std, square: bool) -> int:
    os.makedirs(outfolder, exist_ok=True)
    if last_block_dir is not None:
        os.makedirs(last_block_dir, exist_ok=True)
    files = list_images(directory)
    n_done = 0
    prep = DevicePreprocessor(side, custom_preprocess, mean, std, square, device=lam.engine().dev)
    for step, names in enumerate(_batches(files, batch_size)):
        # decode on the host, resize / normalise / pad on the device (bit-identical to load_image, which stays as the
        # CPU statement of the reference's transform chain)
        imgs = [prep(torch.from_numpy(np.asarray(Image.open(os.path.join(directory, f)).convert("RGB"), dtype=np.uint8).copy()))
                for f in names]
        if len({tuple(i.shape) for i in imgs}) != 1:
            raise ValueError("images of one batch must share a size (use --custom_preprocess or --batch_size 1)")
        x = torch.stack(imgs)
        if last_block_dir is not None:
            out = lam.image_encoder(x, return_last_block_state=True)
            hidden, block = out["last_hidden_state"].cpu(), out["last_block_state"].cpu()
        else:
            hidden, block = lam.image_encoder(x).cpu(), None
        for i, f in enumerate(names):
            stem = os.path.splitext(f)[0]
            save_file({"embedding": hidden[i].contiguous()}, os.path.join(outfolder, f"{stem}.safetensors"))
            if block is not None:
                save_file({"embedding": block[i].contiguous()}, os.path.join(last_block_dir, f"{stem}.safetensors"))
        n_done += len(names)
        if step % 10 == 0:
            logging.info("Step %d/%d", step, (len(files) + batch_size - 1) // batch_size)
    return n_done


def _load_checkpoint(path: str):
    if path.endswith(".safetensors"):
        return load_file(path)
    return torch.load(path, map_location="cpu")


def preprocess_images_to_embeddings(encoder_name, checkpoint, use_sam_checkpoint, directory, batch_size=1, num_workers=0,
                                    outfolder="data/processed/embeddings", last_block_dir=None, device="cuda", compile=False,
                                    custom_preprocess=True, compute_dtype=torch.float16) -> int:
    """SAM-style encoders (preprocess.py:78-139).  ``num_workers`` / ``compile`` are accepted for CLI compatibility."""
    if encoder_name not in ENCODER_SPECS or ENCODER_SPECS[encoder_name].kind != "sam":
        raise KeyError(f"{encoder_name!r} is not a SAM-style encoder; use --huggingface for plain ViTs")
    from label_anything.models import model_registry
    spec = ENCODER_SPECS[encoder_name]
    # preprocess.py:105-107: the encoder-only entry of the registry, built from the checkpoint
    from labelanything_amd.models import build_encoder
    build = model_registry.get(encoder_name) or (lambda **kw: build_encoder(encoder_name, **kw))      # (registered test geometries)
    model = build(checkpoint=checkpoint, use_sam_checkpoint=use_sam_checkpoint, compute_dtype=compute_dtype)
    lam = model.lam
    lam = lam.to(device)
    mean, std = IMAGENET_DEFAULT
    return _run(lam, directory, outfolder, last_block_dir, batch_size, spec.img_size, custom_preprocess, mean, std, square=False)


def preprocess_images_to_embeddings_huggingface(model_name, directory, batch_size=1, num_workers=0,
                                                outfolder="data/processed/embeddings", device="cuda", compile=False,
                                                image_resolution=480, custom_preprocess=True, mean_std="default",
                                                compute_dtype=torch.float16) -> int:
    """Plain HF ViT encoders (preprocess.py:209-246).  ``model_name`` is a LOCAL HuggingFace model directory
    (config.json + model.safetensors | pytorch_model.bin); there is no network in this build."""
    with open(os.path.join(model_name, "config.json")) as fh:
        hc = json.load(fh)
    spec = EncoderSpec("hf", dim=hc["hidden_size"], depth=hc["num_hidden_layers"], heads=hc["num_attention_heads"],
                       mlp=hc["intermediate_size"], patch=hc.get("patch_size", 16), img_size=hc.get("image_size", 224))
    name = "hf:" + os.path.abspath(model_name)
    register_encoder(name, spec)
    lam = Lam(LamConfig(encoder=name, image_size=image_resolution, image_embed_dim=spec.dim, vit_patch_size=spec.patch),
              compute_dtype=compute_dtype)
    wpath = os.path.join(model_name, "model.safetensors")
    sd = load_file(wpath) if os.path.exists(wpath) else torch.load(os.path.join(model_name, "pytorch_model.bin"), map_location="cpu")
    sd = {(k[len("vit."):] if k.startswith("vit.") else k): v for k, v in sd.items() if not k.startswith("decoder.")}
    enc = _hf5_to_hf4({"image_encoder." + k: v for k, v in sd.items()})
    enc = {k: v for k, v in enc.items() if "mask_token" not in k}
    missing, unexpected = torch.nn.Module.load_state_dict(lam, enc, strict=False)
    missing = [k for k in missing if k.startswith("image_encoder.")]
    if missing or unexpected:
        raise RuntimeError(f"weights do not match the ViT config: missing {missing[:5]}, unexpected {unexpected[:5]}")
    lam = lam.to(device)
    mean, std = get_mean_std(mean_std)
    return _run(lam, directory, outfolder, None, batch_size, image_resolution, custom_preprocess, mean, std, square=True)
