"""Embedding cache wire format and the test-time prototype cache.

Read side of what ``generate_embeddings`` writes (label_anything/preprocess.py here; reference ``preprocess.py:53-75``):
one ``<image id, zero padded to 12 digits>.safetensors`` per image holding the fp32 tensor ``"embedding"`` of shape
(C, g, g) - exactly what the reference dataset reads in ``data/coco.py:251-275`` (``_load_safe``) - plus
``set_class_embeddings`` (``experiment/utils.py:210-249``): encode the support set once, keep the prototypes on the model,
then serve query-only batches through ``Lam.predict``.
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Sequence

import torch
from safetensors.torch import load_file

EMBEDDING_KEY = "embedding"


def embedding_path(emb_dir: str, image_id) -> str:
    """``f"{emb_dir}/{str(id).zfill(12)}.safetensors"`` (data/coco.py:262-264)."""
    return os.path.join(emb_dir, f"{str(image_id).zfill(12)}.safetensors")


def load_embedding(emb_dir: str, image_id, gt_name: Optional[str] = None):
    """-> (embedding (C,g,g) fp32, ground truth or None); the optional ground truth is stored as ``"<dataset>_gt"``."""
    f = load_file(embedding_path(emb_dir, image_id))
    if EMBEDDING_KEY not in f:
        raise KeyError(f"{embedding_path(emb_dir, image_id)} has no '{EMBEDDING_KEY}' tensor (keys: {sorted(f)})")
    gt = f[f"{gt_name}_gt"] if gt_name is not None else None
    return f[EMBEDDING_KEY], gt


def load_episode_embeddings(emb_dir: str, image_ids: Sequence[Sequence]) -> torch.Tensor:
    """image_ids[b] = [query id, support ids ...] -> ``embeddings`` (B, M+1, C, g, g) of the batch dict (lam.py:65-89)."""
    rows = []
    for ids in image_ids:
        rows.append(torch.stack([load_embedding(emb_dir, i)[0] for i in ids]))
    if len({tuple(r.shape) for r in rows}) != 1:
        raise ValueError("every episode of a batch needs the same number of images and the same embedding shape")
    return torch.stack(rows)


def set_class_embeddings(model, examples: Dict[str, torch.Tensor], device=None):
    """Encode one support set (tensors WITHOUT the batch axis, as the reference passes them) and keep the result on the model.

    Mirrors experiment/utils.py:210-249: ``examples`` gets a leading batch axis, ``generate_class_embeddings`` runs under
    no_grad and the dict lands in ``model.class_embeddings`` (on the wrapped ``.model`` for a ``LabelAnything``), which
    ``predict`` then uses for every query batch.  The reference's chunk-size retry loop exists to survive OOM on its
    (P, hw, D) intermediates; those are never materialised here, so one call suffices.
    """
    lam = getattr(model, "model", model)
    ex = {k: (v.unsqueeze(0) if isinstance(v, torch.Tensor) else v) for k, v in examples.items()}
    if device is not None:
        ex = {k: (v.to(device) if isinstance(v, torch.Tensor) else v) for k, v in ex.items()}
    with torch.no_grad():
        lam.class_embeddings = lam.generate_class_embeddings(ex, chunk_size=1)
    return model
