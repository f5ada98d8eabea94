#!/usr/bin/env python
"""Golden vectors for the metric row: runs the REFERENCE's ``to_global_multiclass``
(/root/reference/label_anything/data/utils.py:567-590) on seeded label maps and stores inputs + outputs under
tests/golden/metrics_remap.safetensors (+ .json for the python-side arguments).  Build-container tooling only.

    PYTHONDONTWRITEBYTECODE=1 python tools/make_golden_metrics.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")

import types
import importlib.abc
import importlib.machinery

import numpy as np
import torch
from transformers import ViTModel, AutoModel, AutoBackbone, get_scheduler  # noqa: F401  (real imports before the stubs)
import accelerate, huggingface_hub, safetensors.torch  # noqa: F401,E401
from safetensors.torch import save_file

_STUB_ROOTS = {"ruamel", "torchvision", "pycocotools", "torchmetrics", "wandb", "easydict", "cv2", "timm", "dropblock",
               "lovely_tensors", "captum", "optuna", "wget", "nicegui", "streamlit", "colorlog", "skimage", "kornia",
               "streamlit_drawable_canvas", "sklearn"}


class _Stub(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return type(k, (object,), {"__init__": lambda s, *a, **kw: None, "__call__": lambda s, *a, **kw: None,
                                   "__getattr__": lambda s, k: (lambda *a, **kw: None)})


class _Loader(importlib.abc.Loader):
    def create_module(self, spec):
        m = _Stub(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


class _Finder(importlib.abc.MetaPathFinder):
    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in _STUB_ROOTS:
            return importlib.machinery.ModuleSpec(name, _Loader(), is_package=True)


def import_reference():
    """The reference package must win over this repo's same-named shim: its root goes FIRST on sys.path."""
    sys.meta_path.insert(0, _Finder())
    sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != ROOT]
    sys.path.insert(0, "/root/reference")
    sys.path.append(ROOT)


def main():
    import_reference()
    from label_anything.data.utils import to_global_multiclass as ref_fn
    from oracle import metrics_oracle as MO
    g = torch.Generator().manual_seed(77)
    cats = {int(c): {"name": f"c{c}"} for c in (1, 2, 3, 5, 7, 9, 11, 16, 18, 21, 27, 33)}       # 12 categories, sparse ids
    classes = [
        [[5], [5, 9], [9]],              # episode 0: local 1 -> compact 4, local 2 -> compact 6
        [[1, 2, 3]],                     # identity-looking chain (1->1, 2->2, 3->3)
        [[2, 3], [3, 5]],                # local 1 -> 2, then local 2 -> 3 (rewrites the pixels just set to 2!), local 3 -> 4
        [[33], [27, 33], [21]],          # local 1 -> 10, local 2 -> 11, local 3 -> 12
    ]
    b, h, w = len(classes), 32, 40
    preds = torch.randint(0, 4, (b, h, w), generator=g)
    gt = torch.randint(0, 4, (b, h, w), generator=g)
    gt[torch.rand(b, h, w, generator=g) < 0.1] = -100
    out = {}
    for compact in (True, False):
        rp, rg = ref_fn(classes, cats, preds, gt, compact=compact)
        op, og = MO.to_global_multiclass(classes, cats, preds.numpy(), gt.numpy(), compact=compact)
        assert np.array_equal(op, rp.numpy()) and np.array_equal(og, rg.numpy()), "oracle != reference"
        for i in range(b):
            lut = MO.label_lut(classes[i], cats, 8, compact=compact)
            assert np.array_equal(lut[np.clip(preds[i].numpy(), 0, 7)], rp[i].numpy())
        tag = "compact" if compact else "raw"
        out[f"preds_{tag}"] = rp.to(torch.int32).contiguous()
        out[f"gt_{tag}"] = rg.to(torch.int32).contiguous()
    out["preds"] = preds.to(torch.int32)
    out["gt"] = gt.to(torch.int32)
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    save_file(out, os.path.join(ROOT, "tests", "golden", "metrics_remap.safetensors"))
    json.dump({"categories": list(cats.keys()), "classes": classes}, open(os.path.join(ROOT, "tests", "golden", "metrics_remap.json"), "w"))
    print("oracle to_global_multiclass == reference (compact and raw); fixture written")


if __name__ == "__main__":
    main()
