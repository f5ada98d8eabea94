"""Host-side mirror of the reference's model API on top of LamEngine.

  reference                                            here
  label_anything.models.lam.Lam                        Lam
  label_anything.models.build_lam.LabelAnything        LabelAnything   (+ from_pretrained / save_pretrained)
  label_anything.models.build_lam.build_lam*           build_lam, build_lam_no_vit, build_lam_vit_b, ...

Same constructor kwargs, same ``state_dict()`` key layout (``model.image_encoder.*``, ``model.neck.*``,
``model.prompt_encoder.*``, ``model.mask_decoder.*``), same batch dictionary in, same result dictionary out.
Parameters are plain tensors held in a module tree; all arithmetic runs in libla_hip.so.  ``forward`` / ``predict`` are the
inference path; training (forward + backward + AdamW on the same kernels) goes through ``labelanything_amd.train.LamTrainer``.
"""
from __future__ import annotations

import inspect
import json
import os
from typing import Any, Dict, Optional

import torch
from torch import nn

from .config import LamConfig, ENCODER_SPECS, config_from_kwargs
from .engine import LamEngine, PRECISE_DEFAULT, resolve_precise
from .weights import model_shapes, init_state_dict
from . import _lib as L

try:  # the reference mixes in huggingface_hub.PyTorchModelHubMixin (models/hfhub.py:27-47)
    from huggingface_hub import PyTorchModelHubMixin
except Exception:  # pragma: no cover
    class PyTorchModelHubMixin:  # type: ignore
        pass

_BUFFERS = ("positional_encoding_gaussian_matrix",)


class _Tree(nn.Module):
    """Bare container: child modules and parameters are attached by dotted name."""


def _attach(root: nn.Module, name: str, tensor: torch.Tensor) -> None:
    parts = name.split(".")
    mod = root
    for pth in parts[:-1]:
        if pth not in mod._modules:
            mod.add_module(pth, _Tree())
        mod = mod._modules[pth]
    if parts[-1] in _BUFFERS:
        mod.register_buffer(parts[-1], tensor)
    else:
        mod.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=False))


class _EncoderHandle(_Tree):
    """``lam.image_encoder(images)`` like the reference's encoder modules: (Bn,3,S,S) -> (Bn,C,g,g)."""

    def forward(self, x, return_last_block_state: bool = False):
        owner = self.__dict__["_owner_ref"]()
        return owner.encode_images_nchw(x, return_last_block_state=return_last_block_state)


def _on_model_device(fn):
    """Run a Lam entry point with the model's GPU as the current device: every kernel is enqueued on the CURRENT device's
    stream (labelanything_amd._lib), so a model living on cuda:N must not launch from a host thread whose current device is
    another GPU (single-process multi-GPU hosts)."""
    import functools

    @functools.wraps(fn)
    def wrapper(self, *a, **kw):
        dev = self._device()
        if dev.type != "cuda":
            return fn(self, *a, **kw)
        with torch.cuda.device(dev):
            return fn(self, *a, **kw)
    return wrapper


class Lam(nn.Module):
    mask_threshold: float = 0.0
    image_format: str = "RGB"

    def __init__(self, cfg: LamConfig, seed: Optional[int] = None, compute_dtype: torch.dtype = torch.float16,
                 decoder_dtype: Optional[torch.dtype] = torch.float32, precise=PRECISE_DEFAULT):
        super().__init__()
        self.cfg = cfg
        self.image_size = cfg.image_size
        self.custom_preprocess = cfg.custom_preprocess
        self.compute_dtype = compute_dtype
        self.decoder_dtype = decoder_dtype
        self.precise = resolve_precise(cfg, precise, compute_dtype)      # encoder GEMM groups in split precision ('auto': engine.PRECISE_*)
        self.class_embeddings = None
        sd = init_state_dict(cfg, 0 if seed is None else seed)
        for k, v in sd.items():
            _attach(self, k, v)
        if "image_encoder" not in self._modules:
            self.image_encoder = None
        else:
            import weakref
            self.image_encoder.__class__ = _EncoderHandle
            self.image_encoder.__dict__["_owner_ref"] = weakref.ref(self)
        if "neck" not in self._modules:
            self.neck = None
        self._engine: Optional[LamEngine] = None
        self._engine_key = None
        self.weights_version = 0         # bumped by every out-of-band weight change (load_state_dict, _apply, invalidate): consumers that
                                         # keep their own packed copies (train_encoder's private engine, W^T caches) compare it
        self._plist = None
        self._graphs: Dict[Any, Any] = {}
        self.use_graphs = False          # replay the device-side launch sequence from a HIP graph (per input plan)
        self.norm_fold = True            # A/B switch: False keeps the LayerNorm kernels where the engine would fold them into the GEMMs (LamEngine.norm_fold)
        self.attn_fp8 = False            # opt-in: fp8 (e4m3) QK^T in the HF encoder's attention (BASELINE configs[4]); outside the 1e-3 tolerance
        self.selected_rows: Optional[torch.Tensor] = None   # fix the RandomMatrixEncoder rows (parity / reproducibility)

    # -- engine management --------------------------------------------------------------------------
    def _device(self) -> torch.device:
        return self.prompt_encoder.no_mask_embed.weight.device

    def engine(self, validate: bool = True) -> LamEngine:
        """Packed-weight engine for the current parameters.  validate=True re-checks the parameter versions (so in-place
        weight updates / load_state_dict through a wrapper are picked up); hot inner calls pass validate=False."""
        if self._engine is not None and not validate:
            return self._engine
        dev = self._device()
        if dev.type != "cuda":
            raise RuntimeError("labelanything_amd runs on an MI355X only: move the model to 'cuda' (no CPU fallback)")
        if self._plist is None:
            self._plist = list(self.parameters()) + list(self.buffers())
        key = (dev, self.compute_dtype, self.decoder_dtype, self.precise, sum(p._version for p in self._plist))
        if self._engine is None or self._engine_key != key:
            self._graphs = {}
            self._engine = LamEngine(self.cfg, self.state_dict(), dev, self.compute_dtype, self.decoder_dtype, self.precise)
            self._engine_key = key
        if self._engine.attn_fp8 != bool(self.attn_fp8):
            self._engine.attn_fp8 = bool(self.attn_fp8)
            self._graphs = {}                # a captured launch sequence holds the kernel choice it was recorded with
        fold = self._engine.norm_fold_packed and bool(self.norm_fold)
        if self._engine.norm_fold != fold:
            self._engine.norm_fold = fold
            self._graphs = {}
        return self._engine

    def invalidate(self) -> None:
        """Drop the packed-weight engine (and its captured graphs): call after ANY out-of-band parameter update - an optimizer
        that writes through raw pointers (FlatAdamW / la_adamw_step) does not bump the tensors' version counters, so the cache key
        of ``engine()`` cannot see it.  The next forward re-packs from the live parameters."""
        self._engine = None
        self._engine_key = None
        self._graphs = {}
        self.weights_version += 1

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        sd = dict(state_dict)
        # tolerate accelerate / DDP wrappers like the reference's loader (utils/utils.py:119-142)
        if sd and all(k.startswith("model.") for k in sd):
            sd = {k[len("model."):]: v for k, v in sd.items()}
        if sd and all(k.startswith("module.") for k in sd):
            sd = {k[len("module."):]: v for k, v in sd.items()}
        sd = _hf5_to_hf4(sd)
        out = super().load_state_dict(sd, strict=strict)
        self._engine = None
        self.weights_version += 1
        return out

    def _apply(self, fn, *a, **kw):
        self._engine = None
        self._plist = None
        self.weights_version = getattr(self, "weights_version", 0) + 1
        return super()._apply(fn, *a, **kw)

    # -- reference API --------------------------------------------------------------------------------
    def init_pretrained_weights(self, weights) -> None:
        """Seed the encoder and the SAM-shaped decoder parts from a Segment-Anything state dict (models/lam.py:241-319)."""
        init_pretrained_weights(self, weights)

    @_on_model_device
    def get_dense_pe(self) -> torch.Tensor:
        eng = self.engine()
        g = self.cfg.grid
        out = torch.empty(1, self.cfg.embed_dim, g, g, device=eng.dev)
        L.nhwc_to_nchw(eng.dense_pe(g), 1, self.cfg.embed_dim, g * g, out)
        return out

    # ---------------------------------------------------------------------------------------------------------
    # forward = host preparation (flags / sizes decisions, H2D of the small tensors) + a device-only launch sequence.
    # The launch sequence allocates nothing new after its first run and never touches the host, so it can be replayed
    # from a HIP graph (``use_graphs``): the decoder alone is ~170 short launches and is launch-bound otherwise.
    # ---------------------------------------------------------------------------------------------------------
    @staticmethod
    def _any_nonzero(t: torch.Tensor) -> bool:
        """Host decision 'does this prompt type carry any flag' without waking torch's CPU thread pool for a few bytes."""
        if t.is_cuda:
            return bool((t != 0).any())          # device-resident flags: one host sync, as in the reference
        return bool(t.numpy().any())

    def _prepare(self, batched_input: Dict[str, Any], with_prompts: bool = True, with_post: bool = True, eng=None):
        """Host side (lam.py:138-170, 214-239, 401-404): returns (device inputs, hashable plan).  ``eng``: the engine whose
        host helpers to use (the trainer passes its own so that a training step never re-packs the inference weights)."""
        eng = eng if eng is not None else self.engine()
        inp: Dict[str, torch.Tensor] = {}
        if "embeddings" in batched_input:
            emb = batched_input["embeddings"]
            if isinstance(emb, dict):
                raise NotImplementedError("feature pyramids are an off-path ablation of the reference")
            inp["embeddings"] = eng.h2d(emb, torch.float32)
        elif "images" in batched_input:
            inp["images"] = eng.h2d(batched_input["images"], torch.float32)
        else:
            raise ValueError("Either 'images' or 'embeddings' must be provided.")
        kinds = []
        if with_prompts:
            # a prompt type is dropped entirely when all its flags are zero (host decision, lam.py:214-239)
            if "prompt_points" in batched_input and self._any_nonzero(batched_input["flag_points"]):
                inp["prompt_points"] = eng.h2d(batched_input["prompt_points"], torch.float32)
                inp["flag_points"] = eng.h2d(batched_input["flag_points"], torch.int32)
                kinds.append("point")
            if "prompt_bboxes" in batched_input and self._any_nonzero(batched_input["flag_bboxes"]):
                inp["prompt_bboxes"] = eng.h2d(batched_input["prompt_bboxes"], torch.float32)
                inp["flag_bboxes"] = eng.h2d(batched_input["flag_bboxes"], torch.int32)
                kinds.append("box")
            if "prompt_masks" in batched_input and self._any_nonzero(batched_input["flag_masks"]):
                inp["prompt_masks"] = eng.h2d(batched_input["prompt_masks"], torch.float32)
                inp["flag_masks"] = eng.h2d(batched_input["flag_masks"], torch.int32)
                kinds.append("mask")
            if not kinds:
                raise ValueError("No prompts provided")
            inp["flag_examples"] = eng.h2d(batched_input["flag_examples"], torch.uint8)
            if self.cfg.bank_size:
                c = inp["flag_examples"].shape[2]
                rows = self.selected_rows if self.selected_rows is not None else eng.sample_rows(c)
                inp["selected_rows"] = eng.h2d(rows.to(torch.long))
        hmax = wmax = 0
        if with_post:
            sizes, hmax, wmax = eng.post_sizes(batched_input["dims"])
            inp["sizes"] = eng.h2d(sizes)
            if "flag_gts" in batched_input:
                inp["flag_gts"] = eng.h2d(batched_input["flag_gts"], torch.uint8)
        plan = (tuple(sorted((k, tuple(v.shape), str(v.dtype)) for k, v in inp.items())), tuple(kinds), hmax, wmax)
        return inp, plan

    def _embeddings_dev(self, inp: Dict[str, torch.Tensor], apply_neck_to_embeddings: bool):
        """-> (emb32 [B*N*hw, D] NHWC fp32, B, N, g).  Device only."""
        eng = self.engine(validate=False)
        cfg = self.cfg
        if "embeddings" in inp:
            emb = inp["embeddings"]
            b, n, c, h, w = emb.shape
            g = h
            x = emb.reshape(b * n, c, h * w).contiguous()
            need_neck = cfg.lam_neck and apply_neck_to_embeddings
            if not need_neck and c != cfg.embed_dim:
                # lam.py:193-213: prepare_embeddings (generate_class_embeddings / predict) passes cached embeddings
                # through WITHOUT the neck - they must already be embed_dim wide
                raise ValueError(f"embeddings have {c} channels but this path feeds them to the {cfg.embed_dim}-wide decoder "
                                 "without the neck (reference lam.py:193-213); pass post-neck embeddings or use forward()")
            e32 = eng.f32("in.emb32", (b * n * h * w, c))
            e16 = eng.buf("in.emb16", (b * n * h * w, c)) if need_neck else None
            L.nchw_to_nhwc(x, b * n, c, h * w, out32=e32, out16=e16, dt=eng.dti)
            if need_neck:
                e32 = eng.lam_neck(e32, e16, b * n, g)
            return e32, b, n, g
        im = inp["images"]
        b, n = im.shape[:2]
        e32, e16, c, g = eng.encode_images(im.flatten(0, 1))
        if cfg.lam_neck:
            e32 = eng.lam_neck(e32, e16, b * n, g)
        return e32, b, n, g

    def _embeddings_nhwc(self, batched_input: Dict[str, Any], apply_neck_to_embeddings: bool):
        inp, _ = self._prepare(batched_input, with_prompts=False, with_post=False)
        return self._embeddings_dev(inp, apply_neck_to_embeddings)

    def prepare_prompts(self, batched_input):
        """Reference-shaped helper (lam.py:214-239): a prompt type is dropped when all its flags are zero."""
        points = boxes = masks = None
        if "prompt_points" in batched_input and bool((batched_input["flag_points"] != 0).any()):
            points = (batched_input["prompt_points"], batched_input["flag_points"])
        if "prompt_bboxes" in batched_input and bool((batched_input["flag_bboxes"] != 0).any()):
            boxes = (batched_input["prompt_bboxes"], batched_input["flag_bboxes"])
        if "prompt_masks" in batched_input and bool((batched_input["flag_masks"] != 0).any()):
            masks = (batched_input["prompt_masks"], batched_input["flag_masks"])
        return points, boxes, masks, batched_input["flag_examples"]

    @staticmethod
    def _prompts_of(inp):
        points = (inp["prompt_points"], inp["flag_points"]) if "prompt_points" in inp else None
        boxes = (inp["prompt_bboxes"], inp["flag_bboxes"]) if "prompt_bboxes" in inp else None
        masks = (inp["prompt_masks"], inp["flag_masks"]) if "prompt_masks" in inp else None
        return points, boxes, masks

    def _run(self, inp: Dict[str, torch.Tensor], plan, want_argmax: bool, want_post: bool = True):
        """Device-only launch sequence of Lam.forward (lam.py:115-136 + postprocess)."""
        eng = self.engine(validate=False)
        d = self.cfg.embed_dim
        e32, b, n, g = self._embeddings_dev(inp, apply_neck_to_embeddings=True)
        hw = g * g
        ev = e32.view(b, n, hw, d)
        query = ev[:, 0].contiguous().view(b * hw, d)
        support = ev[:, 1:].contiguous().view(b * (n - 1) * hw, d)
        points, boxes, masks = self._prompts_of(inp)
        pe_result = eng.prompt_encoder(support, b, n - 1, g, points, boxes, masks, inp["flag_examples"], inp.get("selected_rows"))
        seg = eng.mask_decoder(query, b, g, pe_result["class_embeddings"])
        out = {"low_res_logits": seg, "class_embeddings": pe_result["class_embeddings"],
               "class_examples_embeddings": pe_result["class_examples_embeddings"]}
        if want_post:
            logits, am = eng.postprocess_dev(seg, inp["sizes"], plan[2], plan[3], inp.get("flag_gts"), want_argmax)
            out["logits"] = logits
            if want_argmax:
                out["argmax"] = am
        return out

    def _run_graphed(self, inp, plan, want_argmax: bool):
        key = (plan, want_argmax)
        entry = self._graphs.get(key)
        if entry is None:
            static = {k: v.clone() for k, v in inp.items()}
            self._run(static, plan, want_argmax)                      # warm-up: sizes every arena buffer, fills caches
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                out = self._run(static, plan, want_argmax)
            entry = (graph, static, out)
            self._graphs[key] = entry
        graph, static, out = entry
        for k, v in inp.items():
            if v.data_ptr() != static[k].data_ptr():
                static[k].copy_(v, non_blocking=True)
        graph.replay()
        # the graph's outputs live in the arena and are overwritten by the next replay: hand out copies (logits of 32 episodes: 0.13 ms)
        return {k: v.clone() for k, v in out.items()}

    @_on_model_device
    @torch.no_grad()
    def _forward(self, batched_input):
        """(low-res logits, prompt-encoder result) like the reference's Lam._forward (lam.py:115-136)."""
        inp, plan = self._prepare(batched_input, with_post=False)
        out = self._run(inp, plan, False, want_post=False)
        return out["low_res_logits"], out

    @_on_model_device
    @torch.no_grad()
    def forward(self, batched_input: Dict[str, Any], want_argmax: bool = False) -> Dict[str, torch.Tensor]:
        inp, plan = self._prepare(batched_input)
        out = self._run_graphed(inp, plan, want_argmax) if self.use_graphs else self._run(inp, plan, want_argmax)
        res = {"logits": out["logits"], "class_examples_embeddings": out["class_examples_embeddings"]}
        if want_argmax:
            res["argmax"] = out["argmax"]
        return res

    def forward_argmax(self, batched_input: Dict[str, Any]):
        """forward + the caller's ``logits.argmax(dim=1)`` (experiment/run.py:697) fused into the last kernel."""
        return self.forward(batched_input, want_argmax=True)

    @_on_model_device
    def postprocess_masks(self, masks: torch.Tensor, original_sizes: torch.Tensor) -> torch.Tensor:
        return self.engine().postprocess(masks.to(self._device(), torch.float32).contiguous(), original_sizes)

    @_on_model_device
    @torch.no_grad()
    def generate_class_embeddings(self, example_dict, chunk_size=None):
        """Supports only: every image of the dict is a support (lam.py:349-360).  chunk_size is accepted and ignored
        (the kernels never materialise the tensors the reference chunks for)."""
        inp, _ = self._prepare(example_dict, with_post=False)
        eng = self.engine(validate=False)
        e32, b, n, g = self._embeddings_dev(inp, apply_neck_to_embeddings=False)
        points, boxes, masks = self._prompts_of(inp)
        res = eng.prompt_encoder(e32, b, n, g, points, boxes, masks, inp["flag_examples"], inp.get("selected_rows"))
        pcount, hw, d = res["class_examples_src"].shape
        src = torch.empty(pcount, d, g, g, device=eng.dev)
        L.nhwc_to_nchw(res["class_examples_src"].contiguous(), pcount, d, hw, src)
        res["class_examples_src"] = src
        return res

    @_on_model_device
    @torch.no_grad()
    def predict(self, batched_input, class_embeddings=None):
        """Query-only encode + decode against cached prototypes (lam.py:362-381)."""
        if class_embeddings is None and self.class_embeddings is None:
            return self.forward(batched_input)
        if class_embeddings is None:
            class_embeddings = self.class_embeddings
        eng = self.engine()
        d = self.cfg.embed_dim
        e32, b, n, g = self._embeddings_nhwc(batched_input, apply_neck_to_embeddings=False)
        query = e32.view(b, n, g * g, d)[:, 0].contiguous().view(b * g * g, d)
        seg = eng.mask_decoder(query, b, g, class_embeddings["class_embeddings"])
        return eng.postprocess(seg, batched_input["dims"].unsqueeze(1))

    @_on_model_device
    @torch.no_grad()
    def encode_images_nchw(self, images: torch.Tensor, return_last_block_state: bool = False):
        """``model.image_encoder(images)`` of the reference: (Bn,3,S,S) -> (Bn,C,g,g) fp32
        (image_encoder.py:110-131, build_encoder.py:83-100)."""
        eng = self.engine()
        x = images.to(eng.dev, torch.float32)
        spec = self.cfg.encoder_spec
        if return_last_block_state and spec.kind == "sam":
            (e32, _, c), last = eng.sam_encoder(x, want_last_block=True)
            g = x.shape[-1] // spec.patch
            out = torch.empty(x.shape[0], c, g, g, device=eng.dev)
            L.nhwc_to_nchw(e32, x.shape[0], c, g * g, out)
            lb = torch.empty(x.shape[0], spec.dim, g, g, device=eng.dev)
            L.nhwc_to_nchw(last, x.shape[0], spec.dim, g * g, lb)
            return {"last_hidden_state": out, "last_block_state": lb}
        e32, _, c, g = eng.encode_images(x)
        out = torch.empty(x.shape[0], c, g, g, device=eng.dev)
        L.nhwc_to_nchw(e32, x.shape[0], c, g * g, out)
        return out

    def get_learnable_params(self, training_params: dict) -> list:
        """lam.py:321-347: the parameter groups the optimizer receives (LamTrainer / FlatAdamW consume them)."""
        def not_enc(kv):
            return "image_encoder" not in kv[0]
        freeze = training_params.get("freeze_backbone", False)
        if freeze and "backbone_lr" in training_params:
            raise ValueError("Cannot freeze the backbone and set a learning rate for it at the same time.")
        named = list(self.named_parameters())
        if freeze:
            return [p for _, p in filter(not_enc, named)]
        if "backbone_lr" in training_params:
            return [{"params": [p for k, p in named if "image_encoder" in k], "lr": training_params["backbone_lr"]},
                    {"params": [p for _, p in filter(not_enc, named)]}]
        return [p for _, p in named]


def _hf5_to_hf4(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Accept transformers-5.x ViT key names for the HF encoder and map them to the 4.x layout."""
    out = {}
    for k, v in sd.items():
        if k.startswith("image_encoder.") and (".layers." in k and "encoder.layer." not in k):
            k = (k.replace("image_encoder.encoder.layers.", "image_encoder.encoder.layer.")
                  .replace("image_encoder.layers.", "image_encoder.encoder.layer.")
                  .replace("attention.q_proj", "attention.attention.query")
                  .replace("attention.k_proj", "attention.attention.key")
                  .replace("attention.v_proj", "attention.attention.value")
                  .replace("attention.o_proj", "attention.output.dense")
                  .replace("mlp.fc1", "intermediate.dense")
                  .replace("mlp.fc2", "output.dense"))
        if "pooler" in k:
            continue
        out[k] = v
    return out


# ----------------------------------------------------------------------------------------------------
def has_config(func):
    """Constructor decorator of the reference's hub wrapper (models/hfhub.py:50-67): the arguments the object was built with -
    explicit ones over declared defaults, a ``config=dict`` argument merged in first - are kept as ``self.config`` (what
    ``save_pretrained`` writes to config.json and ``from_pretrained`` feeds back)."""
    sig = inspect.signature(func)

    def wrapper(self, *args, **kwargs):
        merged = dict(kwargs.pop("config", None) or {})
        merged.update(kwargs)
        bound = sig.bind(self, *args, **merged)
        bound.apply_defaults()
        self.config = {k: v for k, v in list(bound.arguments.items())[1:]}
        func(*bound.args, **bound.kwargs)
    wrapper.__wrapped__ = func
    return wrapper


SAM_EMBED_DIM = 256     # width of a Segment-Anything checkpoint's prompt encoder / mask decoder (models/common.py)

# Segment-Anything checkpoint -> LabelAnything modules (Lam.init_pretrained_weights, models/lam.py:241-319): source prefix in the
# SAM state dict -> destination prefixes here.  SAM's mask-decoder transformer seeds BOTH two-way transformers.
_SAM_INIT_MAP = (
    ("prompt_encoder.pe_layer.", ("prompt_encoder.pe_layer.",)),
    ("prompt_encoder.point_embeddings.", ("prompt_encoder.point_embeddings.",)),
    ("prompt_encoder.not_a_point_embed.", ("prompt_encoder.not_a_point_embed.",)),
    ("prompt_encoder.mask_downscaling.", ("prompt_encoder.mask_downscaling.",)),
    ("prompt_encoder.no_mask_embed.", ("prompt_encoder.no_mask_embed.",)),
    ("mask_decoder.transformer.", ("prompt_encoder.transformer.", "mask_decoder.transformer.")),
    ("mask_decoder.output_upscaling.", ("mask_decoder.output_upscaling.",)),
)


def init_pretrained_weights(lam: Lam, weights: Dict[str, torch.Tensor]) -> None:
    """Initialise a Lam from a Segment-Anything checkpoint (``use_sam_checkpoint=True``; models/lam.py:241-319): the image
    encoder always (every ``image_encoder.*`` tensor, strictly), and - only when the model has SAM's width (D = 256) - the
    positional-encoding matrix, point / not-a-point / no-mask embeddings, mask_downscaling, both two-way transformers and the
    mask decoder's output_upscaling.  Everything else (class attention blocks, class_mlp, spatial convs, LAM neck, class
    encoder) keeps its fresh initialisation.  Like the reference, every destination module is loaded strictly: a missing or
    mis-shaped tensor raises."""
    own = lam.state_dict()
    picked: Dict[str, torch.Tensor] = {}

    def take(src_prefix: str, dst_prefix: str) -> None:
        dst_keys = [k for k in own if k.startswith(dst_prefix)]
        src = {k[len(src_prefix):]: v for k, v in weights.items() if k.startswith(src_prefix)}
        missing = [k for k in dst_keys if k[len(dst_prefix):] not in src]
        extra = [k for k in src if dst_prefix + k not in own]
        if missing or extra:
            raise RuntimeError(f"SAM checkpoint does not match {dst_prefix}*: missing {missing[:4]}, unexpected {extra[:4]}")
        for k in dst_keys:
            v = src[k[len(dst_prefix):]]
            if tuple(v.shape) != tuple(own[k].shape):
                raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(v.shape)} vs model {tuple(own[k].shape)}")
            picked[k] = v

    if lam.image_encoder is not None:
        take("image_encoder.", "image_encoder.")
    if lam.cfg.embed_dim == SAM_EMBED_DIM:
        for src, dsts in _SAM_INIT_MAP:
            for dst in dsts:
                take(src, dst)
    nn.Module.load_state_dict(lam, picked, strict=False)
    lam._engine = None
    lam.weights_version += 1


def build_lam(encoder: Optional[str] = "vit_b", seed: Optional[int] = None, compute_dtype=torch.float16,
              checkpoint: Optional[str] = None, use_sam_checkpoint: bool = False, ignore_encoder_checkpoint: bool = False,
              **kwargs) -> Lam:
    """``build_lam`` of the reference (build_lam.py:180-236).  checkpoint + use_sam_checkpoint: Segment-Anything weights seed
    the encoder and the SAM-shaped decoder parts (init_pretrained_weights); otherwise the checkpoint is a full LabelAnything
    state dict, loaded strictly - with ignore_encoder_checkpoint only ``image_encoder.*`` keys may be absent
    (utils/utils.py:111-139)."""
    cfg = config_from_kwargs(encoder=encoder, **kwargs)
    lam = Lam(cfg, seed=seed, compute_dtype=compute_dtype)
    lam.eval()
    if checkpoint is not None:
        sd = _load_any(checkpoint)
        if use_sam_checkpoint:
            init_pretrained_weights(lam, sd)
        elif ignore_encoder_checkpoint:
            res = lam.load_state_dict(sd, strict=False)
            missing = [k for k in res.missing_keys if "image_encoder" not in k]
            if missing:
                raise RuntimeError(f"Missing keys: {missing}")
            if res.unexpected_keys:
                raise RuntimeError(f"Unexpected keys: {list(res.unexpected_keys)}")
        else:
            lam.load_state_dict(sd)
    return lam


def build_lam_no_vit(**kw) -> Lam:
    return build_lam(encoder=None, use_vit=False, **kw)


def build_lam_vit_b(**kw) -> Lam:
    return build_lam(encoder="vit_b", **kw)


def build_lam_vit_l(**kw) -> Lam:
    return build_lam(encoder="vit_l", **kw)


def build_lam_vit_mae_b(**kw) -> Lam:
    return build_lam(encoder="vit_b_mae", **kw)


def build_lam_vit_h(**kw) -> Lam:
    """build_lam.py:46-50."""
    return build_lam(encoder="vit_h", **kw)


def build_lam_vit_b_imagenet_i21k(**kw) -> Lam:
    """build_lam.py:74-78 (google/vit-base-patch16-224-in21k geometry; weights come from the checkpoint - no hub access here)."""
    return build_lam(encoder="vit_b_imagenet_i21k", **kw)


def build_lam_dino_b8(**kw) -> Lam:
    """build_lam.py:90-94 (facebook/dino-vitb8: 8 x 8 patches)."""
    kw.setdefault("vit_patch_size", 8)
    return build_lam(encoder="vit_dino_b8", **kw)


class ImageEncoder(nn.Module):
    """Encoder-only model, what ``model_registry[encoder_name](...)`` returns in the reference (``ENCODERS``,
    models/build_encoder.py:143-151; consumer: preprocess.py:105-107): ``enc(images)`` -> (Bn, C, g, g) fp32,
    ``enc(images, return_last_block_state=True)`` -> {"last_hidden_state", "last_block_state"} (image_encoder.py:110-131).
    ``state_dict`` / ``load_state_dict`` use the encoder's own key names (no ``image_encoder.`` prefix), like the reference's modules.
    project_last_hidden=False returns the last block state instead of the SAM neck output (image_encoder.py:123-124)."""

    def __init__(self, encoder: str, image_size: Optional[int] = None, project_last_hidden: bool = True, seed: Optional[int] = None,
                 compute_dtype=torch.float16, vit_patch_size: Optional[int] = None):
        super().__init__()
        spec = ENCODER_SPECS[encoder]
        side = image_size or (spec.img_size if spec.kind == "sam" else 480)
        kw = dict(encoder=encoder, image_size=side, vit_patch_size=vit_patch_size or spec.patch)
        if spec.kind == "hf":
            kw.update(image_embed_dim=spec.dim)
        self.lam = Lam(LamConfig(**kw), seed=seed, compute_dtype=compute_dtype)
        self.project_last_hidden = bool(project_last_hidden) or spec.kind == "hf"
        self.kind = spec.kind

    def forward(self, x, return_last_block_state: bool = False):
        if self.kind == "sam" and not self.project_last_hidden:
            return self.lam.encode_images_nchw(x, return_last_block_state=True)["last_block_state"]
        return self.lam.encode_images_nchw(x, return_last_block_state=return_last_block_state)

    def state_dict(self, *a, **kw):
        pre = "image_encoder."
        return {k[len(pre):]: v for k, v in self.lam.state_dict(*a, **kw).items() if k.startswith(pre)}

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        sd = _hf5_to_hf4({"image_encoder." + k: v for k, v in state_dict.items()})
        res = nn.Module.load_state_dict(self.lam, sd, strict=False)
        missing = [k[len("image_encoder."):] for k in res.missing_keys if k.startswith("image_encoder.")]
        unexpected = [k[len("image_encoder."):] for k in res.unexpected_keys]
        self.lam._engine = None
        self.lam.weights_version += 1
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for ImageEncoder: missing {missing[:5]}, unexpected {unexpected[:5]}")
        return res


def _build_sam_encoder(name: str, checkpoint=None, use_sam_checkpoint=False, project_last_hidden=True, **kw) -> ImageEncoder:
    """``_build_vit`` (models/build_encoder.py:43-80): a SAM checkpoint prefixes the encoder's tensors with ``image_encoder.``."""
    enc = ImageEncoder(name, project_last_hidden=project_last_hidden, **kw)
    if checkpoint is not None:
        weights = _load_any(checkpoint)
        if use_sam_checkpoint:
            weights = {k[len("image_encoder."):]: v for k, v in weights.items() if k.startswith("image_encoder.")}
        enc.load_state_dict(weights)
    return enc


def build_vit_h(**kw) -> ImageEncoder:
    return _build_sam_encoder("vit_h", **kw)


def build_vit_l(**kw) -> ImageEncoder:
    return _build_sam_encoder("vit_l", **kw)


def build_vit_b(**kw) -> ImageEncoder:
    return _build_sam_encoder("vit_b", **kw)


def _build_hf_encoder(name: str, project_last_hidden=False, pretrained: Optional[str] = None, **kw) -> ImageEncoder:
    """``ViTModelWrapper.from_pretrained(<hub id>)`` (models/build_encoder.py:103-118).  There is no hub access in this build:
    ``pretrained`` names a LOCAL HuggingFace model directory (model.safetensors | pytorch_model.bin); without it the encoder keeps its
    seeded random initialisation."""
    enc = ImageEncoder(name, **kw)
    if pretrained is not None:
        wpath = os.path.join(pretrained, "model.safetensors")
        sd = _load_any(wpath if os.path.exists(wpath) else os.path.join(pretrained, "pytorch_model.bin"))
        sd = {(k[len("vit."):] if k.startswith("vit.") else k): v for k, v in sd.items()
              if not k.startswith("decoder.") and "mask_token" not in k}
        enc.load_state_dict(sd)
    return enc


def build_vit_b_mae(**kw) -> ImageEncoder:
    return _build_hf_encoder("vit_b_mae", **kw)


def build_vit_b_imagenet_i21k(**kw) -> ImageEncoder:
    return _build_hf_encoder("vit_b_imagenet_i21k", **kw)


def build_vit_dino_b8(**kw) -> ImageEncoder:
    return _build_hf_encoder("vit_dino_b8", **kw)


def build_encoder(name: str, **kw) -> ImageEncoder:
    """models/build_encoder.py:138-141; geometries added with ``config.register_encoder`` build through the same two paths."""
    if name in ENCODERS:
        return ENCODERS[name](**kw)
    if name not in ENCODER_SPECS:
        raise KeyError(f"unknown encoder {name!r}; available: {sorted(ENCODER_SPECS)}")
    return (_build_sam_encoder if ENCODER_SPECS[name].kind == "sam" else _build_hf_encoder)(name, **kw)


# models/build_encoder.py:143-151, on-path entries (resnet50 / swin_b feed the out-of-scope pyramid variants)
ENCODERS: Dict[str, Any] = {
    "vit_h": build_vit_h,
    "vit_l": build_vit_l,
    "vit_b": build_vit_b,
    "vit_b_mae": build_vit_b_mae,
    "vit_dino_b8": build_vit_dino_b8,
}


def _load_any(path: str) -> Dict[str, torch.Tensor]:
    """.pth/.pt/.bin via torch.load, .safetensors via safetensors (utils/utils.py:91-108)."""
    if path.endswith((".pth", ".pt", ".bin")):
        return torch.load(path, map_location="cpu")
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    raise ValueError("File extension not supported")


class LabelAnything(nn.Module, PyTorchModelHubMixin):
    """``label_anything.models.LabelAnything`` (build_lam.py:467-508): ``.model`` is a Lam, ``.config`` a dict of the
    constructor arguments; ``from_pretrained(dir)`` reads config.json + model.safetensors with ``model.``-prefixed keys."""

    @has_config
    def __init__(self, encoder="vit_b", checkpoint=None, use_sam_checkpoint=False, use_vit_sam_neck=True, use_vit=True,
                 image_embed_dim=256, embed_dim=256, image_size=1024, vit_patch_size=16, class_attention=False,
                 example_attention=False, example_class_attention=True, class_embedding_dim=None, spatial_convs=None,
                 encoder_attention_downsample_rate: int = 2, decoder_attention_downsample_rate: int = 2,
                 classification_layer_downsample_rate: int = 8, use_support_features_in_prompt_encoder: bool = True,
                 fusion_transformer="TwoWayTransformer", few_type="Prototype", class_fusion="sum",
                 transformer_keys_are_images=True, transformer_feature_size=None, class_encoder=None,
                 segment_example_logits=False, dropout: float = 0.0, binary=False, custom_preprocess=True):
        super().__init__()
        cfg = dict(self.config)
        enc = cfg.pop("encoder")
        if enc is not None and enc not in ENCODER_SPECS:
            raise KeyError(f"unknown encoder {enc!r}; available: {sorted(ENCODER_SPECS)}")
        self.model = build_lam(encoder=enc if cfg.get("use_vit", True) else None, **cfg)

    def init_pretrained_weights(self, weights) -> None:
        init_pretrained_weights(self.model, weights)

    def forward(self, *args, **kwargs):
        return self.model(*args, **kwargs)

    # PyTorchModelHubMixin in recent huggingface_hub versions handles config.json + model.safetensors itself;
    # these two helpers give the same on-disk format without it (and are what the tests exercise offline).
    def save_local(self, directory: str) -> None:
        from safetensors.torch import save_file
        os.makedirs(directory, exist_ok=True)
        with open(os.path.join(directory, "config.json"), "w") as fh:
            json.dump(self.config, fh, indent=2)
        save_file({k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()}, os.path.join(directory, "model.safetensors"))

    @classmethod
    def from_local(cls, directory: str, **overrides) -> "LabelAnything":
        from safetensors.torch import load_file
        with open(os.path.join(directory, "config.json")) as fh:
            config = json.load(fh)
        config.update(overrides)
        obj = cls(**config)
        sd = load_file(os.path.join(directory, "model.safetensors"))
        obj.model.load_state_dict({k[len("model."):] if k.startswith("model.") else k: v for k, v in sd.items()})
        return obj
