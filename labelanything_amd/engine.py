"""LamEngine: composes the libla_hip.so kernels into the LabelAnything inference path.

Data layout in HBM (all device-resident, reused across calls through a shape-keyed arena):
  * activations are row-major [pixels | tokens, channels] (NHWC); the fp32 residual stream is kept in fp32,
    every GEMM operand is a 16-bit copy written by the producing kernel's epilogue;
  * weights are repacked once per (device, dtype): nn.Linear layout [N, K] in 16 bit, conv weights flattened
    to the im2col column order, ConvTranspose2d(k2,s2) as [(ky,kx,cout), cin], q/k/v projections concatenated;
  * the SAM V operand is written pre-transposed ([b*heads, 64, Tpad]) by the qkv GEMM epilogue.

Each method cites the reference code it replaces (/root/reference/label_anything/...).
PyTorch is used for device memory, views/copies and the host-side prompt bookkeeping only.
"""
from __future__ import annotations


import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

from . import _lib as L
from .config import LamConfig, EncoderSpec

Tensor = torch.Tensor


def _ceil(a: int, b: int) -> int:
    return (a + b - 1) // b * b


class Arena:
    """Shape-keyed cache of device buffers (no allocation in steady state; zero-filled buffers keep their padding)."""

    def __init__(self, device: torch.device):
        self.device = device
        self.bufs: Dict[tuple, Tensor] = {}

    def get(self, name: str, shape, dtype, zero: bool = False, init=None) -> Tensor:
        """init(t): called ONCE, when the buffer is created (constant contents that later launches never overwrite)."""
        key = (name, tuple(shape), dtype)
        t = self.bufs.get(key)
        if t is None:
            try:
                t = (torch.zeros if zero else torch.empty)(tuple(shape), device=self.device, dtype=dtype)
            except RuntimeError as e:  # keep the substring the reference's OOM handler looks for (run.py:339-340)
                raise RuntimeError(f"HIP out of memory allocating {name}{tuple(shape)}: {e}") from e
            if init is not None:
                init(t)
            self.bufs[key] = t
        return t

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.bufs.values())


# Encoder GEMM groups that run in split precision (tools/error_budget.py): the 16-bit rounding of the WEIGHTS is
# the same perturbation for every token, so its effect survives attention and pooling instead of averaging out like the
# activation roundings, and the necks / patch embedding have no residual stream to dilute their error.
#   "qkv", "proj", "lin1", "lin2": weights as two 16-bit planes [W_hi | W_lo] (la_gemm a_kmod), activations 16-bit;
#   "v": only the V rows of the qkv weight carry the second plane (the qkv GEMM becomes a one-plane q/k launch and a two-plane V
#        launch).  tools/error_budget.py: rounding Wq / Wk perturbs the attention SCORES incoherently and washes out (cfg1 low-res
#        logits 9.5e-4 with only Wq, Wk split = the same as with no plane at all), the V projection is linear all the way to the
#        block output (7.0e-4 with only Wv split, 6.7e-4 with all three);
#   "patch", "neck": fp32-class products (fp16 plane pairs on both operands where the shape allows, exact-fp32 MFMA otherwise).
# Which groups are worth their cost was measured on the golden cases (profiles/r03_parity_groups.log) AND over other weight / episode
# seeds (tests/test_parity_seeds_gpu.py, profiles/r03_parity_seeds.log: the max-norm error moves by +-15 % with the seed; worst stage,
# tolerance 1e-3).  The 768+-wide encoders of the BASELINE configs keep patch / V / proj / neck - as second planes in round 2
# (6.6-8.0e-4 over the seeds tried; without proj 7.0-9.4e-4), as token-mean corrections since round 3 (5.6-8.0e-4); lin2's plane would
# buy another ~1e-4 for 8 % of the step.  Encoders narrower than 512 keep the full qkv, proj and lin2 planes (cheap there, and their
# share of the error is larger: sam_tiny 1.02e-3 -> 7.4e-4).
#   "vmean", "projmean": TOKEN-MEAN correction instead of a second plane (round 3).  With one plane token i errs by a_i . W_lo^T; what
#        survives attention and pooling is the part every token of an image shares, mean_i(a_i) . W_lo^T - a per-image vector.  It is
#        formed in fp32 from the token means of the GEMM operands (left behind by the LayerNorm / attention kernels that write them,
#        mean_fix) and ONE few-row exact-fp32 la_gemm per block against [Wo Wv_lo | Wo_lo], and carried as a pending per-image correction
#        R of the residual stream that every LayerNorm adds on the fly (la_layernorm_g).  V: softmax rows sum to one, so the vector
#        c_v = mean(x) Wv_lo^T that belongs on every V row of the image comes out of attention unchanged and goes through proj as
#        c_v Wo^T; proj adds mean(o) Wo_lo^T.  Emulated (tools/error_budget.py --mean) and measured: the same error as
#        the second planes for 3.5 ms instead of 8.7 ms of the cfg2 step.
PRECISE_FULL = ("patch", "qkv", "proj", "lin2", "neck")
PRECISE_WIDE = ("patch", "vmean", "projmean", "neck")
PRECISE_WIDE_PLANES = ("patch", "v", "proj", "neck")          # round 2's default: second weight planes for V and proj
PRECISE_DEFAULT = "auto"
PRECISE_GROUPS = ("patch", "qkv", "v", "proj", "lin1", "lin2", "neck", "vmean", "projmean")


_BICUBIC: Dict[tuple, Tensor] = {}


def bicubic_matrix_t(gin: int, gout: int, device) -> Tensor:
    """B^T [gin^2, gout^2] of the linear map ``F.interpolate(x, (gout, gout), mode='bicubic', align_corners=False)`` on a gin x gin
    grid (transformers ViTEmbeddings.interpolate_pos_encoding under build_encoder.py:83-100), built once per geometry by pushing
    the identity through torch's own CPU kernel - the resample of the position table then is one small exact-fp32 product (and its
    backward the transposed one) instead of torch's device kernels (30 ms per step in the trainable-encoder step, most of it the
    atomics of upsample_bicubic2d_backward)."""
    key = (gin, gout, str(device))
    m = _BICUBIC.get(key)
    if m is None:
        eye = torch.eye(gin * gin).view(gin * gin, 1, gin, gin)
        m = F.interpolate(eye, size=(gout, gout), mode="bicubic", align_corners=False).reshape(gin * gin, gout * gout).contiguous().to(device)
        _BICUBIC[key] = m
    return m


def resolve_precise(cfg: LamConfig, precise, dtype: torch.dtype = torch.float16) -> tuple:
    """'auto' -> the measured default for this encoder width and operand type; None / () -> no split precision; else the given
    groups.  bf16 operands (8 mantissa bits) always take the full set: its activations alone cost more than fp16's weights."""
    if isinstance(precise, str):
        if precise != "auto":
            raise ValueError("precise must be 'auto', None or a sequence of group names")
        spec = cfg.encoder_spec
        wide = spec is not None and spec.dim >= 512 and dtype == torch.float16
        return PRECISE_WIDE if wide else PRECISE_FULL
    return tuple(precise or ())


class LamEngine:
    def __init__(self, cfg: LamConfig, weights: Dict[str, Tensor], device: torch.device, dtype: torch.dtype = torch.float16,
                 decoder_dtype: Optional[torch.dtype] = torch.float32, precise=PRECISE_DEFAULT, fuse_twoway: bool = True,
                 window_scatter: bool = True, scope: str = "all"):
        """dtype: MFMA operand type of the image encoder and necks (>99% of the FLOPs).  decoder_dtype: operand type of the
        prompt encoder / mask decoder GEMMs - fp32 by default (exact-fp32 MFMA; ~1% of the FLOPs but the stage where
        16-bit operand rounding would dominate the logit error), or None to follow ``dtype``.  precise: encoder GEMM groups
        that run in split precision (see PRECISE_DEFAULT); () = every encoder GEMM with plain 16-bit operands.
        fuse_twoway / window_scatter: explicit constructor switches for A/B measurements (tools/) of the fused two-way kernels and of
        the window-order scatter epilogue; the product never reads the environment.  scope="encoder": only the image encoder's weights
        are packed (the trainer-private engine of train_encoder.HfEncoderGraph, re-packed after every optimizer step by
        ``repack_encoder``)."""
        if scope not in ("all", "encoder"):
            raise ValueError("scope must be 'all' or 'encoder'")
        self.scope = scope
        if dtype not in (torch.float16, torch.bfloat16):
            raise ValueError("compute dtype must be torch.float16 or torch.bfloat16")
        self.precise = frozenset(resolve_precise(cfg, precise, dtype))
        # opt-in (Lam.attn_fp8): QK^T of the plain (HF) attention on the fp8 MFMA - BASELINE configs[4].  3 mantissa bits on q and k cost
        # 1e-2-class logit error (profiles/r03_attn_fp8.log): never the parity configuration
        self.attn_fp8 = False
        self.patch_split = False                 # set by _patch_weight
        # decoder_dtype "f16x2": the IMAGE-side GEMM operands of the prompt encoder / mask decoder (the (P, hw, D) stream) are pairs
        # of fp16 planes [hi | lo] and their weights [W_hi | W_hi | W_lo] (LA_F16X2, la_hip.h): 3 fast-MFMA products instead of the
        # exact-fp32 MFMA, same accuracy class; the few-row token side stays exact fp32.
        self.isplit = isinstance(decoder_dtype, str) and decoder_dtype == "f16x2"
        if isinstance(decoder_dtype, str):
            if not self.isplit:
                raise ValueError("decoder_dtype must be a torch dtype, None or 'f16x2'")
            decoder_dtype = torch.float32
            # the narrowest image-side GEMM (output_upscaling.3, K = D / 4) needs 2 K % 64 == 0 for the plane-pair operand:
            # narrower decoders (test geometries) run the exact-fp32 MFMA throughout
            self.isplit = cfg.embed_dim % 128 == 0
        if not self.precise <= set(PRECISE_GROUPS):
            raise ValueError(f"unknown precise groups {sorted(self.precise - set(PRECISE_GROUPS))}; known: {PRECISE_GROUPS}")
        self.kmod: Dict[str, int] = {}      # packed-weight key -> a_kmod of its GEMM (split-precision planes)
        self.mean_kx: Dict[str, int] = {}   # block prefix -> columns of its '.mean.w32' that multiply mean(x) (the rest multiply mean(o))
        # the image side of the two-way transformers runs in the fused kernels (one read / one read + write of the stream per
        # attention, csrc/twoway.hip) for the published decoder geometry; fuse_twoway=False keeps the GEMM + attention + norm chain
        self.fuse_twoway = bool(fuse_twoway) and cfg.embed_dim in (256, 512) and cfg.dec_heads == 8 and decoder_dtype == torch.float32
        self.window_scatter = bool(window_scatter)
        # attention without V^T copies / window buffers (la_attn_fwd_rows) wherever its forms cover the block: plain attention, the 64 x 64
        # rel-pos grid, 16-slot windows; False keeps the V^T epilogue + window scatter path (A/B, and what the other geometries still use)
        self.attn_rows = True
        self.win_fused_cs = True           # folded SAM stack: the window blocks' token means of the attention output from the attention epilogue's column sums
                                           # (on the matrix pipe since round 6; as the global blocks always did) instead of a la_colmean16 pass (A/B: False)
        self.conv_implicit = True          # the necks' 3 x 3 convolution as an implicit GEMM on zero-bordered plane-pair maps (A/B: False = im2col + GEMM)
        self.conv_split = True             # the mask decoder's 32-channel spatial convolutions on la_conv3x3_split (A/B: False = la_conv3x3_f32)
        # LayerNorm folded into its neighbour GEMMs (round 6; LaGemmEpilogue.nstat_out / nstat_in): the residual GEMMs (patch embedding, proj,
        # lin2) also write the 16-bit copy of the stream and the partial sums of every row, q | k | v and lin1 run on gamma-folded weights
        # and normalise the product in their epilogue - no LayerNorm pass over the stream (24 launches, 10.4 ms of a 131 ms cfg2 step).
        # Wide fp16 encoders with one-plane q | k | v and lin1 weights (the measured default of the BASELINE encoders); emulated on the
        # fixtures before it was built (tools/error_budget.py --fold, profiles/r06_normfold_emulation.log).  False keeps the LayerNorm kernels.
        spec = cfg.encoder_spec
        self.norm_fold = (scope == "all" and spec is not None and dtype == torch.float16 and spec.dim >= 512 and spec.dim % 256 == 0
                          and spec.mlp % 256 == 0 and (3 * spec.heads * 64 * ((spec.head_dim + 63) // 64)) % 256 == 0
                          and not (self.precise & {"qkv", "v", "lin1"}))
        self.norm_fold_packed = self.norm_fold      # (the folded weights exist; Lam.norm_fold = False switches back to the LayerNorm kernels)
        self._last16 = None
        self.ddt = dtype if decoder_dtype is None else decoder_dtype
        self.ddti = L._DT[self.ddt]
        L.lib()  # fail loudly if the HIP extension is missing
        self.cfg = cfg
        self.dev = device
        self.dt = dtype
        self.dti = L._DT[dtype]
        self.arena = Arena(device)
        self.w32: Dict[str, Tensor] = {k: v.detach().to(device=device, dtype=torch.float32).contiguous() for k, v in weights.items()}
        self.p: Dict[str, Tensor] = {}
        self._pe_cache: Dict[int, Tensor] = {}
        self._pe_tables: Dict[tuple, Tensor] = {}
        self._hfpos_cache: Dict[int, Tensor] = {}
        self._pack()

    # ------------------------------------------------------------------------------------------------
    # weight packing (one-time layout transforms)
    # ------------------------------------------------------------------------------------------------
    def _h(self, t: Tensor) -> Tensor:
        return t.to(self.dt).contiguous()

    def _hd(self, t: Tensor) -> Tensor:
        return t.to(self.ddt).contiguous()

    @staticmethod
    def _split3(t: Tensor) -> Tensor:
        """[N, K] fp32 -> [N, 3K] fp16 = [W_hi | W_hi | W_lo] (pairs with A = [A_hi | A_lo], a_kmod = 2K)."""
        hi = t.to(torch.float16)
        lo = (t - hi.float()).to(torch.float16)
        return torch.cat([hi, hi, lo], dim=1).contiguous()

    def ibuf(self, name: str, rows: int, e: int) -> Tensor:
        """Image-side GEMM operand buffer: [rows, 2e] fp16 plane pairs in split mode, else a decoder-dtype [rows, e] buffer."""
        if self.isplit:
            return self.arena.get(name, (rows, 2 * e), torch.float16, False)
        return self.dbuf(name, (rows, e))

    @property
    def idti(self) -> int:
        return L.LA_F16X2 if self.isplit else self.ddti

    def igemm(self, a: Tensor, wkey: str, **kw) -> None:
        """GEMM with an image-side A operand against decoder weight ``wkey`` (split-plane aware)."""
        if self.isplit:
            w = self.p[wkey + "s"]
            L.gemm(a, w, a_kmod=2 * (w.shape[1] // 3), **kw)
        else:
            L.gemm(a, self.p[wkey], **kw)

    def _hw(self, key: str, t: Tensor, group: str) -> None:
        """Pack an encoder GEMM weight [N, K]: one 16-bit plane, or [W_hi | W_lo] when its group is precise."""
        t = t.contiguous()
        if group in self.precise and t.shape[1] % 64 == 0:
            hi = t.to(self.dt)
            lo = (t - hi.float()).to(self.dt)
            self.p[key] = torch.cat([hi, lo], dim=1).contiguous()
            self.kmod[key] = t.shape[1]
        else:
            self.p[key] = t.to(self.dt)
            self.kmod.pop(key, None)

    def _pack_fold(self, key: str, wt: Tensor, bias: Tensor, gamma: Tensor, beta: Tensor) -> None:
        """The consumer side of a folded LayerNorm (LaGemmEpilogue.nstat_in): weight rn16(W diag(gamma)) as ``key + "n"``, its row sums
        (what multiplies the row mean) as ``.cn`` and b + W beta as ``.bn`` beside the plain packing of the same layer."""
        wf = (wt * gamma[None, :]).to(self.dt).contiguous()
        self.p[key + "n"] = wf
        self.p[key[:-2] + ".cn"] = wf.double().sum(1).float().contiguous()
        self.p[key[:-2] + ".bn"] = (bias.double() + wt.double() @ beta.double()).float().contiguous()

    def _pack_mean(self, pre: str, wv: Tensor, wo: Tensor, gamma: Optional[Tensor] = None) -> None:
        if gamma is not None and self.norm_fold and self.mean_planes:
            # the folded form multiplies the token means of the normalised rows BEFORE gamma: the lost plane is that of Wv diag(gamma)
            cols = []
            if "vmean" in self.precise:
                wvg = wv * gamma[None, :]
                cols.append((wo.double() @ (wvg - wvg.to(self.dt).float()).double()).float())
            if "projmean" in self.precise:
                cols.append((wo - wo.to(self.dt).float()).float())
            self.p[pre + ".mean.w32n"] = torch.cat(cols, dim=1).contiguous()
        self._pack_mean_plain(pre, wv, wo)

    def _pack_mean_plain(self, pre: str, wv: Tensor, wo: Tensor) -> None:
        """fp32 operand of the token-mean corrections of one block: rvec += [mean(x) | mean(o)] . [Wo Wv_lo | Wo_lo]^T, where *_lo is what
        the 16-bit plane of a weight lost (c_v = mean(x) Wv_lo^T belongs on every V row of the image, softmax rows sum to one, so it
        leaves attention unchanged and goes through proj as c_v Wo^T: one [E, E] product formed here, once)."""
        cols = []
        if "vmean" in self.precise:
            v_lo = (wv - wv.to(self.dt).float()).double()                                       # [ea, E]
            cols.append((wo.double() @ v_lo).float())                                            # [E, E]
        if "projmean" in self.precise:
            cols.append((wo - wo.to(self.dt).float()).float())                                   # [E, ea]
        if cols:
            self.p[pre + ".mean.w32"] = torch.cat(cols, dim=1).contiguous()
            self.mean_kx[pre] = cols[0].shape[1] if "vmean" in self.precise else 0

    @property
    def mean_planes(self) -> bool:
        return "vmean" in self.precise or "projmean" in self.precise

    def mean_parts(self, bn: int, rpg: int, e: int, ea: int, o_chunks: int):
        """Scratch of one block's fused column sums: (LayerNorm partials or None, attention partials or None) - see mean_fix."""
        xp = self.f32("mean.xpart", (bn * L.ln_cs_chunks(rpg) * e,)) if "vmean" in self.precise else None
        op = self.f32("mean.opart", (bn * max(o_chunks, _ceil(rpg, 128) // 128) * ea,)) if "projmean" in self.precise else None
        return xp, op

    def mean_fix(self, pre: str, xpart, opart, rvec: Tensor, bn: int, rpg: int, o_chunks: int, e: int, ea: int, ao=None, ao_win=(0, 0)) -> None:
        """rvec[img] += mean(x) (Wo Wv_lo)^T + mean(o) Wo_lo^T for one attention block (see PRECISE_WIDE).  xpart: column sums of the
        16-bit qkv operand per 128-row chunk, left by the LayerNorm that wrote it (la_layernorm_g); opart: column sums of the attention
        output per 128-query block (o_chunks per image), left by la_attn_fwd_cs - or, o_chunks == 0, scratch for a separate pass over
        ao (la_colmean16; ao_win = (ws, g) when ao is window-partitioned).  Chunks are folded in a fixed order: an image's vectors do
        not depend on the rest of the batch."""
        wm = self.p[pre + ".mean.w32"]
        bar = self.f32("mean.bar", (bn, wm.shape[1]))
        kx = self.mean_kx[pre]
        if xpart is not None:
            L.colsum_fold(xpart, bn, L.ln_cs_chunks(rpg), e, 1.0 / rpg, bar[:, :kx])
        if opart is not None and o_chunks > 0:
            L.colsum_fold(opart, bn, o_chunks, ea, 1.0 / rpg, bar[:, kx:])
        elif opart is not None:
            obar = self.f32("mean.obar", (bn, ea))
            L.colmean16(ao, bn, rpg, obar, opart, ao_win[0], ao_win[1], ao_win[1])
            bar[:, kx:].copy_(obar)
        L.gemm(bar, wm, res=rvec, out32=rvec)

    def _hw_qkv(self, key: str, t: Tensor, bias: Tensor, ea: int) -> None:
        """Pack a fused qkv weight [3 ea, K] + bias.  Group "qkv": planes for all rows; group "v": a one-plane [2 ea, K] q/k weight and
        a two-plane [ea, 2 K] V weight (two launches, qkv_gemm); else one plane."""
        self.p[key[:-2] + ".b"] = bias
        # q | k | v of a token whose LayerNorm output was zero-padded (pad-after-norm windows, image_encoder.py:160-172): the bias, as the
        # GEMM's 16-bit epilogue would round it - the row la_attn_fwd_rows reads for window tokens beyond the image
        self.p[key[:-2] + ".pad16"] = bias.to(self.dt).contiguous()
        if "qkv" not in self.precise and "v" in self.precise and t.shape[1] % 64 == 0 and (2 * ea) % 8 == 0:
            self.p[key + ".qk"] = t[: 2 * ea].to(self.dt).contiguous()
            self._hw(key + ".v", t[2 * ea:], "v")
            self.p.pop(key, None)
            return
        self._hw(key, t, "qkv")

    def qkv_gemm(self, x: Tensor, key: str, qkv: Tensor, vt: Tensor, ea: int, rowmap=None, **vtkw) -> None:
        """qkv[:, :2 ea] = q, k rows of x W^T + b; V goes transposed to vt (la_gemm vt epilogue) - one launch, or two when only the V
        rows of the weight carry a second plane.  rowmap = (map, p): output row map of both (image-order tokens -> window order)."""
        b = self.p[key[:-2] + ".b"]
        mkw = {} if rowmap is None else {"map": rowmap[0], "p": rowmap[1]}
        if vt is None:          # no V^T copy (la_attn_fwd_rows reads the v columns): every column through the plain row-major epilogue
            if (key + ".qk") in self.p:
                L.gemm(x, self.p[key + ".qk"], bias=b[: 2 * ea], out16=qkv[:, : 2 * ea], **mkw)
                L.gemm(x, self.p[key + ".v"], bias=b[2 * ea:], out16=qkv[:, 2 * ea:], a_kmod=self.kmod.get(key + ".v", 0), **mkw)
            else:
                self.gemm_w(x, key, bias=b, out16=qkv, **mkw)
            return
        if (key + ".qk") in self.p:
            L.gemm(x, self.p[key + ".qk"], bias=b[: 2 * ea], out16=qkv[:, : 2 * ea], **mkw)
            L.gemm(x, self.p[key + ".v"], bias=b[2 * ea:], out16=qkv[:, 2 * ea:], vt=vt, vt_col0=0, a_kmod=self.kmod.get(key + ".v", 0),
                   **vtkw, **mkw)
        else:
            self.gemm_w(x, key, bias=b, out16=qkv, vt=vt, vt_col0=2 * ea, **vtkw, **mkw)

    def _patch_weight(self, pw: Tensor) -> Tensor:
        """Patch-embed weight [dim, 3 p p]: plain 16-bit; or, in the split-precision group "patch", fp16 plane triples
        [W_hi | W_hi | W_lo] against [A_hi | A_lo] patches (three fp16 MFMA products, ~21 mantissa bits; 2.8x the rate of
        the exact-fp32 MFMA it replaces) - fp32 for bf16 operands, whose planes would carry 16 bits only."""
        if "patch" not in self.precise:
            return self._h(pw)
        self.patch_split = self.dt == torch.float16 and pw.shape[1] % 64 == 0
        return self._split3(pw) if self.patch_split else pw.contiguous()

    def patches(self, name: str, images: Tensor, rows: int, patch: int):
        """im2col of the patch embedding in the operand form _patch_weight chose; returns (A, extra la_gemm keywords)."""
        k = 3 * patch * patch
        if "patch" not in self.precise:
            a = self.buf(name, (rows, k))
            L.im2col_patch(images, patch, a)
            return a, {}
        if self.patch_split:
            a = self.buf(name, (rows, 2 * k), torch.float16)
            L.im2col_patch(images, patch, a, split=True)
            return a, {"a_kmod": 2 * k}
        a = self.buf(name, (rows, k), torch.float32)
        L.im2col_patch(images, patch, a)
        return a, {}

    def gemm_w(self, a: Tensor, key: str, **kw) -> None:
        """la_gemm against the packed encoder weight ``key`` (split-precision aware)."""
        L.gemm(a, self.p[key], a_kmod=self.kmod.get(key, 0), **kw)

    @property
    def head_pad(self) -> int:
        """Encoder head width as the attention kernels see it: the next multiple of 64 (64 or 128).  Heads of another
        width (SAM ViT-H: 80) get zero weight rows / bias entries / table columns for the extra q, k, v dims and zero
        proj columns, which changes neither q.k nor the projected output."""
        hd = self.cfg.encoder_spec.head_dim
        hdp = 64 * ((hd + 63) // 64)
        if hdp > 128:
            raise NotImplementedError(f"encoder head_dim {hd} > 128 is not built")
        return hdp

    @staticmethod
    def _pad_heads_out(t: Tensor, groups: int, hd: int, hdp: int) -> Tensor:
        """[groups*hd, ...] -> [groups*hdp, ...]: zero rows appended to every head block of an output dimension."""
        if hd == hdp:
            return t.contiguous()
        v = t.reshape(groups, hd, *t.shape[1:])
        pad = torch.zeros(groups, hdp - hd, *t.shape[1:], dtype=t.dtype, device=t.device)
        return torch.cat([v, pad], dim=1).reshape(groups * hdp, *t.shape[1:]).contiguous()

    @staticmethod
    def _pad_heads_in(t: Tensor, groups: int, hd: int, hdp: int) -> Tensor:
        """[N, groups*hd] -> [N, groups*hdp]: zero columns appended to every head block of the input dimension."""
        if hd == hdp:
            return t.contiguous()
        v = t.reshape(t.shape[0], groups, hd)
        return F.pad(v, (0, hdp - hd)).reshape(t.shape[0], groups * hdp).contiguous()

    def _pack_attn(self, pre: str, fuse: str) -> None:
        """decoder Attention (common.py:57-148).  fuse: 'qkv' (same input), 'qk' (q,k share input), 'none'."""
        w, p = self.w32, self.p
        if fuse == "qkv":
            p[pre + ".qkv.w"] = self._hd(torch.cat([w[pre + ".q_proj.weight"], w[pre + ".k_proj.weight"], w[pre + ".v_proj.weight"]]))
            p[pre + ".qkv.b"] = torch.cat([w[pre + ".q_proj.bias"], w[pre + ".k_proj.bias"], w[pre + ".v_proj.bias"]]).contiguous()
        if fuse == "qk":
            p[pre + ".qk.w"] = self._hd(torch.cat([w[pre + ".q_proj.weight"], w[pre + ".k_proj.weight"]]))
            p[pre + ".qk.b"] = torch.cat([w[pre + ".q_proj.bias"], w[pre + ".k_proj.bias"]]).contiguous()
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            p[f"{pre}.{n}.w"] = self._hd(w[f"{pre}.{n}.weight"])
            if self.isplit:
                p[f"{pre}.{n}.ws"] = self._split3(w[f"{pre}.{n}.weight"])
            if self.fuse_twoway and (".cross_attn_" in pre or pre.endswith("final_attn_token_to_image")):
                hi = w[f"{pre}.{n}.weight"].to(torch.float16).contiguous()       # fp16 plane pair of the fused kernels (csrc/twoway.hip)
                p[f"{pre}.{n}.planes"] = (hi, (w[f"{pre}.{n}.weight"] - hi.float()).to(torch.float16).contiguous())

    def _pack_mlp(self, pre: str) -> None:
        self.p[pre + ".lin1.w"] = self._hd(self.w32[pre + ".lin1.weight"])
        self.p[pre + ".lin2.w"] = self._hd(self.w32[pre + ".lin2.weight"])

    def _pack_two_way(self, pre: str) -> None:
        for l in range(2):
            lp = f"{pre}.layers.{l}"
            self._pack_attn(lp + ".self_attn", "qkv" if l == 0 else "qk")
            self._pack_attn(lp + ".cross_attn_token_to_image", "none")
            self._pack_attn(lp + ".cross_attn_image_to_token", "none")
            self._pack_mlp(lp + ".mlp")
        self._pack_attn(pre + ".final_attn_token_to_image", "none")

    def _pack_conv_neck(self, pre: str) -> None:
        w = self.w32
        cast = (lambda t: t.contiguous()) if "neck" in self.precise else self._h
        self.p[pre + ".0.w"] = cast(w[pre + ".0.weight"].flatten(1))
        w2 = w[pre + ".2.weight"].permute(0, 2, 3, 1).flatten(1)                            # [Cout, (ky,kx,cin)]
        self.p[pre + ".2.w"] = cast(w2)
        # fp16 operands: the 3x3 conv of the precise neck runs on plane pairs (LN -> [hi | lo] -> im2col -> three fp16 products, ~21
        # mantissa bits) instead of the exact-fp32 implicit GEMM: 2.5x faster in the same accuracy class
        if "neck" in self.precise and self.dt == torch.float16 and w2.shape[0] % 32 == 0 and w2.shape[1] % 32 == 0:
            self.p[pre + ".2.ws"] = self._split3(w2)
        # ... and so does the 1 x 1 conv when its input arrives as plane pairs (conv_neck xs: the SAM stack's last pass over the stream
        # writes them, la_add_rowvec_split): 393216 x 256 x 768 in 0.7 ms instead of 1.6 ms on the exact-fp32 MFMA
        w0 = w[pre + ".0.weight"].flatten(1)
        if "neck" in self.precise and self.dt == torch.float16 and w0.shape[1] % 64 == 0 and w0.shape[0] % 32 == 0:
            self.p[pre + ".0.ws"] = self._split3(w0)

    def _pack(self) -> None:
        cfg, w, p = self.cfg, self.w32, self.p
        spec = cfg.encoder_spec
        if spec is not None and spec.kind == "sam":
            pre = "image_encoder"
            pw = w[pre + ".patch_embed.proj.weight"].flatten(1)
            p[pre + ".patch.w"] = self._patch_weight(pw)
            p[pre + ".pos"] = w[pre + ".pos_embed"].reshape(-1, spec.dim).contiguous()
            g = spec.img_size // spec.patch
            hd, hdp = spec.head_dim, self.head_pad
            for i in range(spec.depth):
                bp = f"{pre}.blocks.{i}"
                self._hw_qkv(bp + ".qkv.w", self._pad_heads_out(w[bp + ".attn.qkv.weight"], 3 * spec.heads, hd, hdp),
                             self._pad_heads_out(w[bp + ".attn.qkv.bias"], 3 * spec.heads, hd, hdp), spec.heads * hdp)
                self._hw(bp + ".proj.w", self._pad_heads_in(w[bp + ".attn.proj.weight"], spec.heads, hd, hdp), "proj")
                self._pack_mean(bp, self._pad_heads_out(w[bp + ".attn.qkv.weight"], 3 * spec.heads, hd, hdp)[2 * spec.heads * hdp:],
                                self._pad_heads_in(w[bp + ".attn.proj.weight"], spec.heads, hd, hdp), gamma=w[bp + ".norm1.weight"])
                if self.norm_fold:
                    self._pack_fold(bp + ".qkv.w", self._pad_heads_out(w[bp + ".attn.qkv.weight"], 3 * spec.heads, hd, hdp),
                                    self._pad_heads_out(w[bp + ".attn.qkv.bias"], 3 * spec.heads, hd, hdp), w[bp + ".norm1.weight"], w[bp + ".norm1.bias"])
                    self._pack_fold(bp + ".lin1.w", w[bp + ".mlp.lin1.weight"], w[bp + ".mlp.lin1.bias"], w[bp + ".norm2.weight"], w[bp + ".norm2.bias"])
                self._hw(bp + ".lin1.w", w[bp + ".mlp.lin1.weight"], "lin1")
                self._hw(bp + ".lin2.w", w[bp + ".mlp.lin2.weight"], "lin2")
                size = g if i in spec.global_idx else spec.window
                for ax in ("h", "w"):
                    tab = w[f"{bp}.attn.rel_pos_{ax}"]
                    if tab.shape[0] != 2 * size - 1:   # get_rel_pos linear resampling (image_encoder.py:321-330), constant per model
                        tab = F.interpolate(tab.t().unsqueeze(0), size=2 * size - 1, mode="linear")[0].t()
                    p[f"{bp}.tab{ax}"] = self._h(F.pad(tab, (0, hdp - hd)))
            self._pack_conv_neck(pre + ".neck")
        elif spec is not None and spec.kind == "hf":
            pre = "image_encoder"
            pw = w[pre + ".embeddings.patch_embeddings.projection.weight"].flatten(1)
            p[pre + ".patch.w"] = self._patch_weight(pw)
            hd, hdp = spec.head_dim, self.head_pad
            for i in range(spec.depth):
                lp = f"{pre}.encoder.layer.{i}"
                qkv_w = torch.cat([w[lp + ".attention.attention.query.weight"], w[lp + ".attention.attention.key.weight"],
                                   w[lp + ".attention.attention.value.weight"]])
                qkv_b = torch.cat([w[lp + ".attention.attention.query.bias"], w[lp + ".attention.attention.key.bias"],
                                   w[lp + ".attention.attention.value.bias"]])
                self._hw_qkv(lp + ".qkv.w", self._pad_heads_out(qkv_w, 3 * spec.heads, hd, hdp),
                             self._pad_heads_out(qkv_b, 3 * spec.heads, hd, hdp), spec.heads * hdp)
                self._hw(lp + ".o.w", self._pad_heads_in(w[lp + ".attention.output.dense.weight"], spec.heads, hd, hdp), "proj")
                self._pack_mean(lp, self._pad_heads_out(w[lp + ".attention.attention.value.weight"], spec.heads, hd, hdp),
                                self._pad_heads_in(w[lp + ".attention.output.dense.weight"], spec.heads, hd, hdp), gamma=w[lp + ".layernorm_before.weight"])
                if self.norm_fold:
                    self._pack_fold(lp + ".qkv.w", self._pad_heads_out(qkv_w, 3 * spec.heads, hd, hdp), self._pad_heads_out(qkv_b, 3 * spec.heads, hd, hdp),
                                    w[lp + ".layernorm_before.weight"], w[lp + ".layernorm_before.bias"])
                    self._pack_fold(lp + ".fc1.w", w[lp + ".intermediate.dense.weight"], w[lp + ".intermediate.dense.bias"],
                                    w[lp + ".layernorm_after.weight"], w[lp + ".layernorm_after.bias"])
                self._hw(lp + ".fc1.w", w[lp + ".intermediate.dense.weight"], "lin1")
                self._hw(lp + ".fc2.w", w[lp + ".output.dense.weight"], "lin2")
        if self.scope == "encoder":
            return
        if cfg.lam_neck:
            self._pack_conv_neck("neck")
        pe = "prompt_encoder"
        p[pe + ".type_emb"] = torch.cat([w[f"{pe}.point_embeddings.{i}.weight"] for i in range(4)]).contiguous()
        p[pe + ".mask_w"] = [w[pe + ".mask_downscaling.0.weight"].contiguous(), w[pe + ".mask_downscaling.0.bias"],
                             w[pe + ".mask_downscaling.1.weight"], w[pe + ".mask_downscaling.1.bias"],
                             w[pe + ".mask_downscaling.3.weight"].contiguous(), w[pe + ".mask_downscaling.3.bias"],
                             w[pe + ".mask_downscaling.4.weight"], w[pe + ".mask_downscaling.4.bias"],
                             w[pe + ".mask_downscaling.6.weight"].flatten(1).contiguous(), w[pe + ".mask_downscaling.6.bias"],
                             w[pe + ".not_a_mask_embed.weight"].flatten().contiguous(), w[pe + ".no_mask_embed.weight"].flatten().contiguous()]
        self._pack_two_way(pe + ".transformer")
        for blk in ("sparse_embedding_attention", "class_attention", "class_example_attention", "example_attention"):
            if f"{pe}.{blk}.norm.weight" in w:
                self._pack_attn(f"{pe}.{blk}.attn", "qkv")
                self._pack_mlp(f"{pe}.{blk}.mlp")
        md = "mask_decoder"
        self._pack_two_way(md + ".transformer")
        for i in range(3):
            p[f"{md}.class_mlp.{i}.w"] = self._hd(w[f"{md}.class_mlp.layers.{i}.weight"])
        # ConvTranspose2d weight (Cin, Cout, 2, 2) -> GEMM weight [(ky, kx, cout), cin]
        p[md + ".up0.w"] = self._hd(w[md + ".output_upscaling.0.weight"].permute(2, 3, 1, 0).flatten(0, 2))
        p[md + ".up3.w"] = self._hd(w[md + ".output_upscaling.3.weight"].permute(2, 3, 1, 0).flatten(0, 2))
        if self.isplit:
            p[md + ".up0.ws"] = self._split3(w[md + ".output_upscaling.0.weight"].permute(2, 3, 1, 0).flatten(0, 2))
            p[md + ".up3.ws"] = self._split3(w[md + ".output_upscaling.3.weight"].permute(2, 3, 1, 0).flatten(0, 2))
        if cfg.spatial_convs:
            for i in range(cfg.spatial_convs):
                p[f"{md}.sc{i}.w"] = self._hd(w[f"{md}.spatial_convs.{3 * i}.weight"].permute(0, 2, 3, 1).flatten(1))

    def repack_encoder(self, weights: Dict[str, Tensor]) -> None:
        """Refresh the packed image-encoder weights from ``weights`` (live parameters after an optimizer step); the arena, the cached
        positional tables that do not depend on parameters and everything outside the encoder stay.  Only for scope="encoder" engines."""
        if self.scope != "encoder":
            raise RuntimeError("repack_encoder is for scope='encoder' engines; full engines are rebuilt by Lam.engine()")
        for k, v in weights.items():
            if k.startswith("image_encoder."):
                self.w32[k] = v.detach().to(device=self.dev, dtype=torch.float32).contiguous()
        self._hfpos_cache.clear()          # resampled position embeddings are functions of a parameter
        self._pack()

    # ------------------------------------------------------------------------------------------------
    # small helpers
    # ------------------------------------------------------------------------------------------------
    def h2d(self, t: Tensor, dtype=None) -> Tensor:
        """Small host tensor -> device without stalling the launch queue (pinned staging + async copy)."""
        if t.is_cuda:
            return t.to(self.dev, dtype) if dtype is not None else t.to(self.dev)
        if dtype is not None:
            t = t.to(dtype)
        return t.contiguous().pin_memory().to(self.dev, non_blocking=True)

    def buf(self, name, shape, dtype=None, zero=False, init=None) -> Tensor:
        return self.arena.get(name, shape, self.dt if dtype is None else dtype, zero, init)

    def dbuf(self, name, shape) -> Tensor:
        return self.arena.get(name, shape, self.ddt, False)

    def dln(self, x, name, eps, **kw):
        L.layernorm(x, self.w32[name + ".weight"], self.w32[name + ".bias"], eps, dt=self.ddti, **kw)

    def f32(self, name, shape, zero=False) -> Tensor:
        return self.arena.get(name, shape, torch.float32, zero)

    def dense_pe(self, g: int) -> Tensor:
        """[g*g, D] fp32, cached per grid (input independent; prompt_encoder.py:213-224)."""
        t = self._pe_cache.get(g)
        if t is None:
            t = torch.empty(g * g, self.cfg.embed_dim, device=self.dev, dtype=torch.float32)
            L.dense_pe(self.w32["prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"], g, self.cfg.embed_dim, t)
            self._pe_cache[g] = t
        return t

    def ln(self, x, name, eps, rvec=None, rpg=0, **kw):
        """LayerNorm of x (+ the pending per-image corrections rvec[row // rpg] of the mean planes, never written back)."""
        if rvec is not None:       # (colsum_part=...: also leave the column sums of the stored rows, mean_fix)
            L.layernorm_g(x, rvec, rpg, self.w32[name + ".weight"], self.w32[name + ".bias"], eps, dt=self.dti, **kw)
        else:
            L.layernorm(x, self.w32[name + ".weight"], self.w32[name + ".bias"], eps, dt=self.dti, **kw)

    # ------------------------------------------------------------------------------------------------
    # conv neck: 1x1 conv -> LN2d -> 3x3 conv -> LN2d on NHWC rows (image_encoder.py:92-108, build_lam.py:150-171)
    # ------------------------------------------------------------------------------------------------
    def conv_neck(self, pre: str, x16: Optional[Tensor], bn: int, g: int, tag: str, x32: Optional[Tensor] = None,
                  xs: Optional[Tensor] = None) -> Tensor:
        """xs: x32 as fp16 plane pairs [rows, 2 Cin] (LA_F16X2), when the caller's last pass over the stream wrote them."""
        cout = self.p[pre + ".0.w"].shape[0]
        rows = bn * g * g
        a = self.f32(tag + ".n0", (rows, cout))
        if "neck" in self.precise:      # fp32-level products on the fp32 stream itself (no single 16-bit copy of the input at all)
            if xs is not None and (pre + ".0.ws") in self.p:
                L.gemm(xs, self.p[pre + ".0.ws"], out32=a, a_kmod=xs.shape[1])
            else:
                L.gemm(x32, self.p[pre + ".0.w"], out32=a)
            a1 = None
            if (pre + ".2.ws") not in self.p:
                a1 = self.f32(tag + ".n1f", (rows, cout))
                L.layernorm(a, self.w32[pre + ".1.weight"], self.w32[pre + ".1.bias"], 1e-6, out32=a1, dt=L.LA_F32)
            if (pre + ".2.ws") in self.p and self.conv_implicit and cout % 64 == 0 and cout % 256 == 0 and rows > 512:
                # IMPLICIT 3 x 3 convolution (round 6): the LayerNorm writes its plane pairs into the interior of zero-bordered
                # [bn, g + 2, g + 2] maps, the GEMM's k-tiles read the nine taps as wave-uniform shifts of its source base
                # (LA_MAP_CONV3X3) - no im2col buffer (18 x the map written and read again: 1.1 ms + most of the 1.2 ms GEMM's traffic) - and
                # the last LayerNorm gathers the interior rows.  Border rows of the product are computed and never read.
                gp = g + 2
                mp, guard = bn * gp * gp, gp + 1
                a1p = self.arena.get(tag + ".n1p", (mp + 2 * guard, 2 * cout), torch.float16, True)      # zero once: borders and guard rows stay zero
                L.layernorm(a, self.w32[pre + ".1.weight"], self.w32[pre + ".1.bias"], 1e-6, out16=a1p[guard:], dt=L.LA_F16X2, window=-1, H=g, W=g)
                a2p = self.f32(tag + ".n2p", (mp, cout))
                L.gemm(a1p[guard:guard + mp], self.p[pre + ".2.ws"], out32=a2p, amap=L.MAP_CONV3X3, p=(gp, cout, 2 * cout, 0, 0))
                out = self.f32(tag + ".out", (rows, cout))
                L.layernorm(a2p, self.w32[pre + ".3.weight"], self.w32[pre + ".3.bias"], 1e-6, out32=out, dt=L.LA_F32, window=-2, H=g, W=g)
                return out
            if (pre + ".2.ws") in self.p:
                a1s = self.buf(tag + ".n1s", (rows, 2 * cout), torch.float16)
                L.layernorm(a, self.w32[pre + ".1.weight"], self.w32[pre + ".1.bias"], 1e-6, out16=a1s, dt=L.LA_F16X2)
                col = self.buf(tag + ".cols", (rows, 18 * cout), torch.float16)
                L.im2col_3x3(a1s, bn, g, g, cout, col, split=True)
                L.gemm(col, self.p[pre + ".2.ws"], out32=a, a_kmod=18 * cout)
            elif cout % 32 == 0:
                L.conv3x3_f32(a1, bn, g, g, cout, self.p[pre + ".2.w"], None, cout, a)
            else:
                col = self.f32(tag + ".colf", (rows, 9 * cout))
                L.im2col_3x3(a1, bn, g, g, cout, col)
                L.gemm(col, self.p[pre + ".2.w"], out32=a)
            out = self.f32(tag + ".out", (rows, cout))
            L.layernorm(a, self.w32[pre + ".3.weight"], self.w32[pre + ".3.bias"], 1e-6, out32=out, dt=L.LA_F32)
            return out
        L.gemm(x16, self.p[pre + ".0.w"], out32=a)
        a16 = self.buf(tag + ".n1", (rows, cout))
        self.ln(a, pre + ".1", 1e-6, out16=a16)
        col = self.buf(tag + ".col", (rows, 9 * cout))
        L.im2col_3x3(a16, bn, g, g, cout, col)
        L.gemm(col, self.p[pre + ".2.w"], out32=a)
        out = self.f32(tag + ".out", (rows, cout))
        self.ln(a, pre + ".3", 1e-6, out32=out)
        return out

    # ------------------------------------------------------------------------------------------------
    # SAM ViTDet encoder (image_encoder.py:110-131,179-197,200-255)
    # ------------------------------------------------------------------------------------------------
    def _check_encoder_input(self, images: Tensor) -> None:
        spec: EncoderSpec = self.cfg.encoder_spec
        _ = self.head_pad          # raises for head widths beyond 128
        if images.shape[-1] % spec.patch or self.cfg.vit_patch_size != spec.patch:
            raise ValueError(f"image side {images.shape[-1]} / vit_patch_size {self.cfg.vit_patch_size} do not match the "
                             f"encoder's {spec.patch}x{spec.patch} patches")

    def sam_encoder(self, images: Tensor, want_last_block: bool = False):
        spec: EncoderSpec = self.cfg.encoder_spec
        self._check_encoder_input(images)
        pre = "image_encoder"
        bn, _, s, _ = images.shape
        if s != spec.img_size:
            raise ValueError(f"SAM encoder expects {spec.img_size}x{spec.img_size} inputs, got {s}")
        e, heads, g, ws = spec.dim, spec.heads, s // spec.patch, spec.window
        hw = g * g
        rows = bn * hw
        scale = spec.head_dim ** -0.5
        hdp = self.head_pad
        ea = heads * hdp                # width of the q / k / v / attention-output blocks (== e unless the heads are padded)
        w, p = self.w32, self.p
        images = images.contiguous()
        # (the producer epilogue adds the per-image correction to groups of whole row tiles or of >= 128 rows: smaller grids keep the kernels)
        if self.norm_fold and self.attn_rows and ws <= 16 and (g == 64 or not spec.global_idx) and (hw % 256 == 0 or hw >= 128):
            return self._sam_encoder_fold(images, want_last_block)
        a, akw = self.patches("enc.patchA", images, rows, spec.patch)
        res = self.f32("enc.res", (rows, e))
        L.gemm(a, p[pre + ".patch.w"], bias=w[pre + ".patch_embed.proj.bias"], res=p[pre + ".pos"], res_mod=hw, out32=res, **akw)
        nwy = (g + ws - 1) // ws
        x16 = self.buf("enc.x16", (rows, e))
        last16 = None
        rvec = None
        if self.mean_planes:           # pending per-image corrections of the single-plane V / proj weights (PRECISE_WIDE)
            rvec = self.f32("enc.rvec", (bn, e), zero=True)
            rvec.zero_()
        rkw = dict(rvec=rvec, rpg=hw) if rvec is not None else {}
        for i in range(spec.depth):
            bp = f"{pre}.blocks.{i}"
            is_global = i in spec.global_idx
            xpart = opart = None
            if rvec is not None:
                o_chunks = _ceil(hw, 128) // 128 if is_global else nwy * nwy * (_ceil(ws * ws, 128) // 128)
                xpart, opart = self.mean_parts(bn, hw, e, ea, o_chunks)
            ckw = dict(colsum_part=xpart) if xpart is not None else {}
            if is_global:
                nb, t, gg, arows = bn, hw, g, rows
                xin = x16
                self.ln(res, bp + ".norm1", 1e-6, out16=xin, **rkw, **ckw)
            else:
                nb, t, gg = bn * nwy * nwy, ws * ws, ws
                arows = nb * t
            win16 = (not is_global) and gg <= 16      # windows: V^T / K in 16-wide padded slot order (LA_ATTN_RELPOS_WIN16)
            tpad = _ceil(16 * gg, 64) if win16 else _ceil(t, 64)
            tag = "g" if is_global else "w"
            # No V^T copy, no window buffers (la_attn_fwd_rows): the q | k | v GEMM of EVERY block walks the image-order tokens with its plain
            # row-major epilogue; attention stages V tiles row-major (LDS transpose reads) and, for 14 x 14 windows, addresses the image's
            # tokens directly (tokens beyond the image are the bias row); its output is in image order, so proj is a plain GEMM as well.
            # Measured (profiles/r05_notes.md 2): the V^T scatter cost 110 / 275 us of a 1.53 / 1.65 ms launch, the kernels are equal or faster.
            rows_path = self.attn_rows and ((is_global and gg == 64) or win16)
            if rows_path:
                xin = x16
                if not is_global:
                    self.ln(res, bp + ".norm1", 1e-6, out16=xin, **rkw, **ckw)
                qkv = self.buf("enc.qkv.r", (rows, 3 * ea))
                self.qkv_gemm(xin, bp + ".qkv.w", qkv, None, ea)
                ao = self.buf("enc.ao.r", (rows, ea))
                fused_o = opart is not None and (is_global or self.win_fused_cs)
                if is_global:
                    L.attn_fwd_rows(qkv, ao, nb, heads, t, tpad, gg, ea, scale, L.ATTN_RELPOS, tabh=p[bp + ".tabh"], tabw=p[bp + ".tabw"],
                                    cspart=opart if fused_o else None)
                else:
                    L.attn_fwd_rows(qkv, ao, nb, heads, t, tpad, gg, ea, scale, L.ATTN_RELPOS_WIN16, tabh=p[bp + ".tabh"], tabw=p[bp + ".tabw"],
                                    img_hw=(g, g), padrow=p[bp + ".qkv.pad16"], cspart=opart if fused_o else None)
                if rvec is not None:
                    self.mean_fix(bp, xpart, opart, rvec, bn, hw, o_chunks if fused_o else 0, e, ea, ao=ao)
                self.gemm_w(ao, bp + ".proj.w", bias=w[bp + ".attn.proj.bias"], res=res, out32=res)
                self._sam_mlp(bp, i, res, x16, rows, spec, w, rvec, rkw)
                last16 = self._last16 if self._last16 is not None else last16
                continue
            # Window blocks whose qkv weight is one plane (at most the V rows carry a second one): the GEMMs walk the REAL tokens in
            # image order and their epilogue scatters q / k rows and V^T slots into window order (LA_MAP_WINDOW_PART) - the padded
            # tokens (16 % of the rows at 64 x 64 / 14) are never multiplied.  Their q / k / v are the bias (pad-after-norm), constant
            # per block: every window block owns its buffers, filled once when they are created.
            scatter = win16 and "qkv" not in self.precise and arows > rows and self.window_scatter
            if is_global:
                pass
            elif scatter:
                xin = x16
                self.ln(res, bp + ".norm1", 1e-6, out16=xin, **rkw, **ckw)
            else:
                xin = self.buf("enc.xwin", (arows, e), zero=True)        # padded tokens stay zero
                self.ln(res, bp + ".norm1", 1e-6, out16=xin, window=ws, H=g, W=g, **rkw, **ckw)
            if scatter:
                qb = p[bp + ".qkv.b"]

                def fill_qk(tq, qb=qb):
                    tq[:, : 2 * ea] = qb[: 2 * ea].to(tq.dtype)

                def fill_v(tv, qb=qb):
                    tv.zero_()
                    tv.view(nb, heads, hdp, tpad)[..., : 16 * gg].view(nb, heads, hdp, gg, 16)[..., :gg] = \
                        qb[2 * ea:].view(1, heads, hdp, 1, 1).to(tv.dtype)

                qkv = self.arena.get(f"enc.qkv.w{i}", (arows, 3 * ea), self.dt, False, fill_qk)
                vt = self.arena.get(f"enc.vt.w{i}", (nb * heads, hdp, tpad), self.dt, False, fill_v)
                self.qkv_gemm(xin, bp + ".qkv.w", qkv, vt, ea, rowmap=(L.MAP_WINDOW_PART, (ws, nwy, nwy, g, g)), vt_T=t, vt_Tpad=tpad,
                              vt_hd=hdp, vt_heads=heads, vt_ws=gg)
            else:
                qkv = self.buf("enc.qkv." + tag, (arows, 3 * ea))
                vt = self.buf("enc.vt." + tag, (nb * heads, hdp, tpad), zero=True)
                self.qkv_gemm(xin, bp + ".qkv.w", qkv, vt, ea, vt_T=t, vt_Tpad=tpad, vt_hd=hdp, vt_heads=heads, vt_ws=gg if win16 else 0)
            ao = self.buf("enc.ao." + tag, (arows, ea))
            # column sums of the attention output from the attention kernel itself where that is cheaper than a pass over the output:
            # 260 VALU operations per wave at the end of >= 15 key tiles (global blocks: +2 %), not of a window's 4 (+30 %, measured)
            fused_o = opart is not None and not win16 and gg > 16
            okw = dict(cspart=opart) if fused_o else None

            def attn(relh, relw, mode, **tk):
                if okw is not None:
                    L.attn_fwd_cs(qkv, vt, ao, relh, relw, nb, heads, t, tpad, gg, ea, scale, mode, **okw, **tk)
                else:
                    L.attn_fwd(qkv, vt, ao, relh, relw, nb, heads, t, tpad, gg, ea, scale, mode, **tk)

            if win16:
                attn(None, None, L.ATTN_RELPOS_WIN16, tabh=p[bp + ".tabh"], tabw=p[bp + ".tabw"])
            elif gg <= 16 or gg == 64:      # rel-pos terms are computed inside the attention kernel
                attn(None, None, L.ATTN_RELPOS, tabh=p[bp + ".tabh"], tabw=p[bp + ".tabw"])
            else:
                relh = self.f32("enc.relh." + tag, (nb * heads, t, gg))
                relw = self.f32("enc.relw." + tag, (nb * heads, t, gg))
                L.relpos_terms(qkv, nb, heads, gg, ea, p[bp + ".tabh"], p[bp + ".tabw"], relh, relw)
                attn(relh, relw, L.ATTN_RELPOS)
            if rvec is not None:
                self.mean_fix(bp, xpart, opart, rvec, bn, hw, o_chunks if fused_o else 0, e, ea, ao=ao,
                              ao_win=(0, 0) if is_global else (ws, g))
            if is_global:
                self.gemm_w(ao, bp + ".proj.w", bias=w[bp + ".attn.proj.bias"], res=res, out32=res)
            else:       # window_unpartition as a row gather on the A operand: again only the real tokens are computed
                self.gemm_w(ao, bp + ".proj.w", bias=w[bp + ".attn.proj.bias"], res=res, out32=res, M=rows,
                            amap=L.MAP_WINDOW_PART, p=(ws, nwy, nwy, g, g))
            self._sam_mlp(bp, i, res, x16, rows, spec, w, rvec, rkw)
            last16 = self._last16 if self._last16 is not None else last16
        xs = None
        split_neck = self.cfg.use_vit_sam_neck and (pre + ".neck.0.ws") in p
        if split_neck:                 # the stream's last pass also leaves it as fp16 plane pairs: the neck's 1 x 1 conv operand
            # (range: the pair [hi | lo] holds |x| <= 131008 - la_add_rowvec_split saturates beyond, it never emits inf / NaN; SAM
            # checkpoints keep the un-normalised stream three orders of magnitude below that)
            xs = self.buf("enc.res_split", (rows, 2 * e), torch.float16)
            L.add_rowvec_split(res, rvec, hw, xs)
        elif rvec is not None:         # the stream leaves the block stack: fold the pending corrections in
            L.add_rowvec(res, rvec, hw)
        if rvec is not None and not (self.cfg.use_vit_sam_neck and "neck" in self.precise):
            last16 = self.buf("enc.last16", (rows, e))
            L.add_cast(res, out16=last16, dt=self.dti)
        if not self.cfg.use_vit_sam_neck:
            return (res, last16, e) if not want_last_block else ((res, last16, e), res)
        out = self.conv_neck(pre + ".neck", last16, bn, g, "enc.neck", x32=res, xs=xs)
        if want_last_block:
            return (out, None, spec.out_chans), res
        return out, None, spec.out_chans

    # ---- LayerNorm folded into its neighbour GEMMs (norm_fold) -------------------------------------------------------------------------
    def _fold_bufs(self, tag: str, rows: int, e: int):
        """(partial row sums of the producer GEMMs, (mean, rstd) rows of the consumer GEMMs - padded to whole 256-row tiles)."""
        return self.f32(tag + ".npart", (rows, e // 64, 2)), self.f32(tag + ".nmr", (_ceil(rows, 256), 2), zero=True)

    def _fold_mean(self, pre: str, xpart, opart, bn: int, rpg: int, o_chunks: int, e: int, ea: int, ao) -> Optional[Tensor]:
        """The block's token-mean correction as a FRESH per-image vector (the proj GEMM's epilogue adds it to the stream: nothing stays
        pending): dR[img] = [mean(z) | mean(o)] . [Wo (Wv gamma)_lo | Wo_lo]^T, z = the normalised rows before gamma (la_norm_finalize)."""
        if not self.mean_planes:
            return None
        wm = self.p[pre + ".mean.w32n"]
        bar = self.f32("mean.bar", (bn, wm.shape[1]))
        kx = self.mean_kx[pre]
        if xpart is not None:
            L.colsum_fold(xpart, bn, L.norm_cs_chunks(rpg), e, 1.0 / rpg, bar[:, :kx])
        if opart is not None and o_chunks > 0:
            L.colsum_fold(opart, bn, o_chunks, ea, 1.0 / rpg, bar[:, kx:])
        elif opart is not None:
            obar = self.f32("mean.obar", (bn, ea))
            L.colmean16(ao, bn, rpg, obar, opart, 0, 0, 0)
            bar[:, kx:].copy_(obar)
        dr = self.f32("mean.dr", (bn, e))
        L.gemm(bar, wm, out32=dr)
        return dr

    def _planes_to_f32(self, xs: Tensor, name: str) -> Tensor:
        """hi + lo of a plane-pair stream as an fp32 matrix (the callers that leave the folded path: final LayerNorm, un-split necks)."""
        e = xs.shape[1] // 2
        out = self.f32(name, (xs.shape[0], e))
        out.copy_(xs[:, :e])
        out.add_(xs[:, e:])
        return out

    def _sam_encoder_fold(self, images: Tensor, want_last_block: bool = False):
        """sam_encoder without LayerNorm passes (image_encoder.py:110-131,179-197): every residual GEMM is the PRODUCER of the next
        LayerNorm's input, q | k | v and lin1 are its CONSUMERS (gamma-folded weights, normalisation in the epilogue); the token-mean
        correction of a block joins the stream in the proj epilogue.  The residual stream itself is a pair of fp16 planes [hi | lo]
        (same bytes as fp32, ~22 mantissa bits): the producers read-modify-write it in place and the hi plane IS the consumers' operand -
        no separate 16-bit copy (a producer epilogue is bound by HBM round trips: the copy's 2 E bytes per row were + 20 % on proj,
        + 8 % on lin2) - and the neck's 1 x 1 convolution takes the pair as it stands."""
        spec: EncoderSpec = self.cfg.encoder_spec
        pre = "image_encoder"
        bn, _, s, _ = images.shape
        e, heads, g, ws = spec.dim, spec.heads, s // spec.patch, spec.window
        hw = g * g
        rows = bn * hw
        scale = spec.head_dim ** -0.5
        hdp = self.head_pad
        ea = heads * hdp
        w, p = self.w32, self.p
        a, akw = self.patches("enc.patchA", images, rows, spec.patch)
        xs = self.buf("enc.xs", (rows, 2 * e), torch.float16)          # the stream: [hi | lo]
        hi, lo = xs[:, :e], xs[:, e:]
        part, mr = self._fold_bufs("enc", rows, e)
        have_mr = False
        if hw % 256 == 0:      # (plane pairs straight from the patch embedding's epilogue: no fp32 matrix is written at all)
            L.gemm(a, p[pre + ".patch.w"], bias=w[pre + ".patch_embed.proj.bias"], res=p[pre + ".pos"], res_mod=hw, out16=hi, aux16=lo,
                   nstat_out=part, **akw)
        else:       # (a position table that is not whole row tiles: planes and statistics from a pass each, once)
            res32 = self.f32("enc.res", (rows, e))
            L.gemm(a, p[pre + ".patch.w"], bias=w[pre + ".patch_embed.proj.bias"], res=p[pre + ".pos"], res_mod=hw, out32=res32, **akw)
            L.add_rowvec_split(res32, None, hw, xs)
            L.norm_stats(res32, 1e-6, self.buf("enc.x16", (rows, e)), mr)
            have_mr = True
        nwy = (g + ws - 1) // ws
        vm, pm = "vmean" in self.precise, "projmean" in self.precise
        for i in range(spec.depth):
            bp = f"{pre}.blocks.{i}"
            is_global = i in spec.global_idx
            o_chunks = _ceil(hw, 128) // 128 if is_global else nwy * nwy * (_ceil(ws * ws, 128) // 128)
            xpart = self.f32("mean.xpart", (bn * L.ln_cs_chunks(hw) * e,)) if vm else None
            opart = self.f32("mean.opart", (bn * max(o_chunks, _ceil(hw, 128) // 128) * ea,)) if pm else None
            if vm or not have_mr:
                L.norm_finalize(None if have_mr else part, rows, e, 1e-6, mr, x16=hi if vm else None, rpg=hw, cs_part=xpart)
            have_mr = False
            qkv = self.buf("enc.qkv.r", (rows, 3 * ea))
            L.gemm(hi, p[bp + ".qkv.wn"], bias=p[bp + ".qkv.bn"], out16=qkv, nstat_in=mr, ncol=p[bp + ".qkv.cn"])
            ao = self.buf("enc.ao.r", (rows, ea))
            fused_o = opart is not None and (is_global or self.win_fused_cs)
            if is_global:
                L.attn_fwd_rows(qkv, ao, bn, heads, hw, _ceil(hw, 64), g, ea, scale, L.ATTN_RELPOS, tabh=p[bp + ".tabh"], tabw=p[bp + ".tabw"],
                                cspart=opart if fused_o else None)
            else:
                L.attn_fwd_rows(qkv, ao, bn * nwy * nwy, heads, ws * ws, _ceil(16 * ws, 64), ws, ea, scale, L.ATTN_RELPOS_WIN16,
                                tabh=p[bp + ".tabh"], tabw=p[bp + ".tabw"], img_hw=(g, g), padrow=p[bp + ".qkv.pad16"],
                                cspart=opart if fused_o else None)
            dr = self._fold_mean(bp, xpart, opart, bn, hw, o_chunks if fused_o else 0, e, ea, ao)
            L.gemm(ao, p[bp + ".proj.w"], bias=w[bp + ".attn.proj.bias"], out16=hi, aux16=lo, nstat_out=part, rvec=dr,
                   rvec_rpg=hw if dr is not None else 0, a_kmod=self.kmod.get(bp + ".proj.w", 0))
            L.norm_finalize(part, rows, e, 1e-6, mr)
            hbuf = self.buf("enc.mlp", (rows, spec.mlp))
            L.gemm(hi, p[bp + ".lin1.wn"], bias=p[bp + ".lin1.bn"], out16=hbuf, act=L.ACT_GELU, nstat_in=mr, ncol=p[bp + ".lin1.cn"])
            L.gemm(hbuf, p[bp + ".lin2.w"], bias=w[bp + ".mlp.lin2.bias"], out16=hi, aux16=lo, nstat_out=part,
                   a_kmod=self.kmod.get(bp + ".lin2.w", 0))
        split_neck = self.cfg.use_vit_sam_neck and (pre + ".neck.0.ws") in p
        res = None
        if want_last_block or not split_neck:
            res = self._planes_to_f32(xs, "enc.res")
        if not self.cfg.use_vit_sam_neck:
            last16 = self.buf("enc.last16", (rows, e))
            last16.copy_(hi)
            return (res, last16, e) if not want_last_block else ((res, last16, e), res)
        out = self.conv_neck(pre + ".neck", hi, bn, g, "enc.neck", x32=res, xs=xs if split_neck else None)
        if want_last_block:
            return (out, None, spec.out_chans), res
        return out, None, spec.out_chans

    def _sam_mlp(self, bp: str, i: int, res: Tensor, x16: Tensor, rows: int, spec, w, rvec, rkw) -> None:
        """norm2 + lin1 (GELU) + lin2 (residual) of one SAM block; the last block may leave a 16-bit copy of the stream in ``self._last16``."""
        e = spec.dim
        self.ln(res, bp + ".norm2", 1e-6, out16=x16, **rkw)
        hbuf = self.buf("enc.mlp", (rows, spec.mlp))
        self.gemm_w(x16, bp + ".lin1.w", bias=w[bp + ".mlp.lin1.bias"], out16=hbuf, act=L.ACT_GELU)
        self._last16 = None
        if i == spec.depth - 1 and not (self.cfg.use_vit_sam_neck and "neck" in self.precise) and rvec is None:
            self._last16 = self.buf("enc.last16", (rows, e))
            self.gemm_w(hbuf, bp + ".lin2.w", bias=w[bp + ".mlp.lin2.bias"], res=res, out32=res, out16=self._last16)
        else:
            self.gemm_w(hbuf, bp + ".lin2.w", bias=w[bp + ".mlp.lin2.bias"], res=res, out32=res)

    # ------------------------------------------------------------------------------------------------
    # HuggingFace plain ViT encoder (transformers ViTModel maths; build_encoder.py:83-100)
    # ------------------------------------------------------------------------------------------------
    def _hf_pos(self, g: int) -> Tensor:
        t = self._hfpos_cache.get(g)
        if t is None:
            spec = self.cfg.encoder_spec
            pos = self.w32["image_encoder.embeddings.position_embeddings"]
            if g != spec.pos_grid:  # bicubic resample of the patch positions (constant per resolution)
                e = pos.shape[-1]
                bt = bicubic_matrix_t(spec.pos_grid, g, pos.device)                        # [gin^2, gout^2]
                grid = torch.zeros(g * g, e, device=pos.device)
                L.gemm_tn(bt, pos[0, 1:].contiguous(), grid)                               # grid = B . pos  (exact-fp32 MFMA)
                pos = torch.cat([pos[:, :1], grid.unsqueeze(0)], dim=1)
            t = pos[0].contiguous()
            cls_row = (self.w32["image_encoder.embeddings.cls_token"][0, 0] + t[0]).contiguous()
            self._hfpos_cache[g] = t
            self._hfpos_cache[-g] = cls_row
        return t

    def hf_encoder(self, images: Tensor):
        spec: EncoderSpec = self.cfg.encoder_spec
        self._check_encoder_input(images)
        pre = "image_encoder"
        bn, _, s, _ = images.shape
        e, heads, g = spec.dim, spec.heads, s // spec.patch
        hw = g * g
        t = hw + 1
        rows = bn * t
        w, p = self.w32, self.p
        pos = self._hf_pos(g)
        cls_row = self._hfpos_cache[-g]
        images = images.contiguous()
        a, akw = self.patches("hf.patchA", images, bn * hw, spec.patch)
        res = self.f32("hf.res", (rows, e))
        res.view(bn, t, e)[:, 0].copy_(cls_row)      # CLS row = cls_token + pos[0] (weights only; plain copy)
        L.gemm(a, p[pre + ".patch.w"], bias=w[pre + ".embeddings.patch_embeddings.projection.bias"], res=pos, res_mod=t,
               out32=res, map=L.MAP_GROUP, p=(hw, t, 1, 0, 0), **akw)
        tpad = _ceil(t, 64)
        x16 = self.buf("hf.x16", (rows, e))
        hdp = self.head_pad
        ea = heads * hdp
        qkv = self.buf("hf.qkv", (rows, 3 * ea))
        fp8 = self.attn_fp8 and hdp == 64          # (the fp8 QK^T kernel keeps its V^T operand)
        rows_path = self.attn_rows and not fp8     # no V^T copy: la_attn_fwd_rows
        if self.norm_fold and rows_path and t >= 128:       # (per-image groups of the producer epilogue: >= 128 rows)
            return self._hf_fold_blocks(res, bn, t, hw, e, heads, hdp, ea, spec, qkv)
        vt = None if rows_path else self.buf("hf.vt", (bn * heads, hdp, tpad), zero=True)
        ao = self.buf("hf.ao", (rows, ea))
        hbuf = self.buf("hf.mlp", (rows, spec.mlp))
        scale = spec.head_dim ** -0.5
        rvec = None
        if self.mean_planes:
            rvec = self.f32("hf.rvec", (bn, e), zero=True)
            rvec.zero_()
        rkw = dict(rvec=rvec, rpg=t) if rvec is not None else {}
        xpart = opart = None
        o_chunks = _ceil(t, 128) // 128
        if rvec is not None:
            xpart, opart = self.mean_parts(bn, t, e, ea, o_chunks)
        ckw = dict(colsum_part=xpart) if xpart is not None else {}
        for i in range(spec.depth):
            lp = f"{pre}.encoder.layer.{i}"
            self.ln(res, lp + ".layernorm_before", 1e-12, out16=x16, **rkw, **ckw)
            if rows_path:
                self.qkv_gemm(x16, lp + ".qkv.w", qkv, None, ea)
            else:
                self.qkv_gemm(x16, lp + ".qkv.w", qkv, vt, ea, vt_T=t, vt_Tpad=tpad, vt_hd=hdp, vt_heads=heads)
            fused_o = opart is not None and not fp8
            if rows_path:
                L.attn_fwd_rows(qkv, ao, bn, heads, t, tpad, 0, ea, scale, L.ATTN_PLAIN, cspart=opart if fused_o else None)
            elif self.attn_fp8 and hdp == 64:
                qk8 = self.arena.get("hf.qk8", (rows, 2 * ea), torch.uint8, False)
                L.qk_fp8(qkv, ea, qk8)
                L.attn_fwd_fp8(qk8, vt, ao, bn, heads, t, tpad, ea, scale)
            elif fused_o:
                L.attn_fwd_cs(qkv, vt, ao, None, None, bn, heads, t, tpad, 0, ea, scale, L.ATTN_PLAIN, opart)
            else:
                L.attn_fwd(qkv, vt, ao, None, None, bn, heads, t, tpad, 0, ea, scale, L.ATTN_PLAIN)
            if rvec is not None:
                self.mean_fix(lp, xpart, opart, rvec, bn, t, o_chunks if fused_o else 0, e, ea, ao=ao)
            self.gemm_w(ao, lp + ".o.w", bias=w[lp + ".attention.output.dense.bias"], res=res, out32=res)
            self.ln(res, lp + ".layernorm_after", 1e-12, out16=x16, **rkw)
            self.gemm_w(x16, lp + ".fc1.w", bias=w[lp + ".intermediate.dense.bias"], out16=hbuf, act=L.ACT_GELU)
            self.gemm_w(hbuf, lp + ".fc2.w", bias=w[lp + ".output.dense.bias"], res=res, out32=res)
        fin = self.f32("hf.final", (rows, e))
        fin16 = self.buf("hf.final16", (rows, e))
        self.ln(res, pre + ".layernorm", 1e-12, out32=fin, out16=fin16, **rkw)
        out32 = self.f32("hf.out32", (bn * hw, e))
        out16 = self.buf("hf.out16", (bn * hw, e))
        out32.view(bn, hw, e).copy_(fin.view(bn, t, e)[:, 1:])          # drop CLS (plain strided copy)
        out16.view(bn, hw, e).copy_(fin16.view(bn, t, e)[:, 1:])
        return out32, out16, e

    def _hf_fold_blocks(self, res: Tensor, bn: int, t: int, hw: int, e: int, heads: int, hdp: int, ea: int, spec, qkv: Tensor):
        """The HF block stack without LayerNorm passes (transformers ViTLayer: layernorm_before / layernorm_after folded into the q | k | v
        and fc1 GEMMs; eps 1e-12) on a plane-pair stream: see _sam_encoder_fold.  The stream enters through a row map (CLS gap), so its
        planes and first statistics come from one pass each; the final ``layernorm`` is the LayerNorm kernel on hi + lo."""
        pre = "image_encoder"
        rows = bn * t
        w, p = self.w32, self.p
        xs = self.buf("hf.xs", (rows, 2 * e), torch.float16)
        hi, lo = xs[:, :e], xs[:, e:]
        part, mr = self._fold_bufs("hf", rows, e)
        L.add_rowvec_split(res, None, t, xs)
        L.norm_stats(res, 1e-12, self.buf("hf.x16", (rows, e)), mr)
        have_mr = True
        ao = self.buf("hf.ao", (rows, ea))
        hbuf = self.buf("hf.mlp", (rows, spec.mlp))
        scale = spec.head_dim ** -0.5
        tpad = _ceil(t, 64)
        vm, pm = "vmean" in self.precise, "projmean" in self.precise
        o_chunks = _ceil(t, 128) // 128
        xpart = self.f32("mean.xpart", (bn * L.ln_cs_chunks(t) * e,)) if vm else None
        opart = self.f32("mean.opart", (bn * o_chunks * ea,)) if pm else None
        for i in range(spec.depth):
            lp = f"{pre}.encoder.layer.{i}"
            if vm or not have_mr:
                L.norm_finalize(None if have_mr else part, rows, e, 1e-12, mr, x16=hi if vm else None, rpg=t, cs_part=xpart)
            have_mr = False
            L.gemm(hi, p[lp + ".qkv.wn"], bias=p[lp + ".qkv.bn"], out16=qkv, nstat_in=mr, ncol=p[lp + ".qkv.cn"])
            L.attn_fwd_rows(qkv, ao, bn, heads, t, tpad, 0, ea, scale, L.ATTN_PLAIN, cspart=opart)
            dr = self._fold_mean(lp, xpart, opart, bn, t, o_chunks, e, ea, ao)
            L.gemm(ao, p[lp + ".o.w"], bias=w[lp + ".attention.output.dense.bias"], out16=hi, aux16=lo, nstat_out=part, rvec=dr,
                   rvec_rpg=t if dr is not None else 0, a_kmod=self.kmod.get(lp + ".o.w", 0))
            L.norm_finalize(part, rows, e, 1e-12, mr)
            L.gemm(hi, p[lp + ".fc1.wn"], bias=p[lp + ".fc1.bn"], out16=hbuf, act=L.ACT_GELU, nstat_in=mr, ncol=p[lp + ".fc1.cn"])
            L.gemm(hbuf, p[lp + ".fc2.w"], bias=w[lp + ".output.dense.bias"], out16=hi, aux16=lo, nstat_out=part,
                   a_kmod=self.kmod.get(lp + ".fc2.w", 0))
        res = self._planes_to_f32(xs, "hf.res")
        fin = self.f32("hf.final", (rows, e))
        fin16 = self.buf("hf.final16", (rows, e))
        self.ln(res, pre + ".layernorm", 1e-12, out32=fin, out16=fin16)
        out32 = self.f32("hf.out32", (bn * hw, e))
        out16 = self.buf("hf.out16", (bn * hw, e))
        out32.view(bn, hw, e).copy_(fin.view(bn, t, e)[:, 1:])          # drop CLS (plain strided copy)
        out16.view(bn, hw, e).copy_(fin16.view(bn, t, e)[:, 1:])
        return out32, out16, e

    def encode_images(self, images: Tensor):
        """(Bn,3,S,S) fp32 -> (emb32 [Bn*hw, C] NHWC fp32, emb16 or None, C, g)."""
        spec = self.cfg.encoder_spec
        if spec is None:
            raise ValueError("this model was built without an image encoder (use_vit=False)")

        if spec.kind == "sam":
            out32, out16, c = self.sam_encoder(images)
        else:
            out32, out16, c = self.hf_encoder(images)
        return out32, out16, c, images.shape[-1] // spec.patch

    def lam_neck(self, emb32: Tensor, emb16: Optional[Tensor], bn: int, g: int) -> Tensor:
        if "neck" in self.precise:
            return self.conv_neck("neck", None, bn, g, "lamneck", x32=emb32)
        if emb16 is None:
            emb16 = self.buf("neck.in16", tuple(emb32.shape))
            L.add_cast(emb32, out16=emb16, dt=self.dti)
        return self.conv_neck("neck", emb16, bn, g, "lamneck")

    # ------------------------------------------------------------------------------------------------
    # decoder attention building blocks
    # ------------------------------------------------------------------------------------------------
    def _attn_core(self, q32, k32, v32, groups, nq, nk, internal, tag, image_side: bool = False) -> Tensor:
        """image_side: the output feeds an image-side GEMM (rows = groups * hw): plane-pair operand in split mode."""
        heads = self.cfg.dec_heads
        if image_side:
            o16 = self.ibuf(tag + ".o16", groups * nq, internal)
            L.attn_small(q32, k32, v32, groups, nq, nk, heads, internal // heads, out16=o16, dt=self.idti)
            return o16
        o16 = self.dbuf(tag + ".o16", (groups * nq, internal))
        L.attn_small(q32, k32, v32, groups, nq, nk, heads, internal // heads, out16=o16, dt=self.ddti)
        return o16

    def attention_mlp_block(self, pre: str, x32: Tensor, groups: int, n: int, tag: str) -> Tensor:
        """AttentionMLPBlock (common.py:151-184): y = LN(attn(x)+x); out = LN(mlp(y)+y), one shared LayerNorm, GELU."""
        w, p = self.w32, self.p
        rows, d = x32.shape
        internal = p[pre + ".attn.q_proj.w"].shape[0]
        x16 = self.dbuf(tag + ".x16", (rows, d))
        L.add_cast(x32, out16=x16, dt=self.ddti)
        qkv = self.f32(tag + ".qkv", (rows, 3 * internal))
        L.gemm(x16, p[pre + ".attn.qkv.w"], bias=p[pre + ".attn.qkv.b"], out32=qkv)
        o16 = self._attn_core(qkv[:, :internal], qkv[:, internal:2 * internal], qkv[:, 2 * internal:], groups, n, n, internal, tag)
        y = self.f32(tag + ".y", (rows, d))
        L.gemm(o16, p[pre + ".attn.out_proj.w"], bias=w[pre + ".attn.out_proj.bias"], res=x32, out32=y)
        y16 = self.dbuf(tag + ".y16", (rows, d))
        self.dln(y, pre + ".norm", 1e-5, out32=y, out16=y16)
        hbuf = self.dbuf(tag + ".h", (rows, self.cfg.dec_mlp))
        L.gemm(y16, p[pre + ".mlp.lin1.w"], bias=w[pre + ".mlp.lin1.bias"], out16=hbuf, act=L.ACT_GELU)
        z = self.f32(tag + ".z", (rows, d))
        L.gemm(hbuf, p[pre + ".mlp.lin2.w"], bias=w[pre + ".mlp.lin2.bias"], res=y, out32=z)
        self.dln(z, pre + ".norm", 1e-5, out32=z)
        return z

    def fused_ok(self, nt: int) -> bool:
        return self.fuse_twoway and nt <= 32

    def pe_table(self, proj: str, pe32: Tensor) -> Tensor:
        """pe W^T + b of one image-side projection, [hw, internal] fp32: the positional encoding's share of (x + pe) W^T + b is a constant
        of (weights, grid), computed once per engine and grid (exact-fp32 MFMA) and added inside the fused two-way kernels."""
        key = (proj, pe32.shape[0])
        t = self._pe_tables.get(key)
        if t is None:
            t = torch.empty(pe32.shape[0], self.w32[proj + ".weight"].shape[0], device=self.dev)
            L.gemm(pe32, self.w32[proj + ".weight"], bias=self.w32[proj + ".bias"], out32=t)
            t = L.twoway_pe_layout(t)            # in the order the tile kernels read it (la_twoway_pe_layout)
            self._pe_tables[key] = t
        return t

    def _t2i(self, ca: str, q: Tensor, img32: Tensor, pe32: Tensor, img16, imgpe16, groups: int, nt: int, hw: int, tag: str) -> Tensor:
        """Attention output (before out_proj) of the tokens over the image side; fused: K / V are never materialised."""
        w, p = self.w32, self.p
        di = self.cfg.embed_dim // 2
        if self.fused_ok(nt):
            o = self.f32(tag + ".t2i.o", (groups * nt, di))
            part = self.f32(tag + ".t2i.part", (L.twoway_part_size(groups, hw, nt, self.cfg.embed_dim),))
            L.twoway_t2i(img32, p[ca + ".k_proj.planes"], p[ca + ".v_proj.planes"], self.pe_table(ca + ".k_proj", pe32), w[ca + ".v_proj.bias"], q,
                         groups, hw, nt, self.cfg.dec_heads, part, o)
            return o
        ri = groups * hw
        k = self.f32(tag + ".ik", (ri, di))
        v = self.f32(tag + ".iv", (ri, di))
        self.igemm(imgpe16, ca + ".k_proj.w", bias=w[ca + ".k_proj.bias"], out32=k)
        self.igemm(img16, ca + ".v_proj.w", bias=w[ca + ".v_proj.bias"], out32=v)
        return self._attn_core(q, k, v, groups, nt, hw, di, tag + ".t2i")

    def two_way(self, pre: str, tok32: Tensor, groups: int, nt: int, img32: Tensor, img16, imgpe16, hw: int,
                pe32: Tensor, tag: str, want_tokens: bool):
        """TwoWayTransformer (transformer.py:206-329).  tok32 [groups*nt, D] fp32 (also the token PE); image side
        [groups*hw, D] as fp32 stream, updated in place (+ its GEMM-operand copies x and x+pe when the unfused chain runs:
        ``fused_ok(nt)`` false).  Returns (tokens32, tokens16)."""
        w, p, cfg = self.w32, self.p, self.cfg
        d = cfg.embed_dim
        di = d // 2
        r = groups * nt
        ri = groups * hw
        fused = self.fused_ok(nt)
        tpe = tok32
        t32 = self.f32(tag + ".t32", (r, d))
        t16 = self.dbuf(tag + ".t16", (r, d))
        tq16 = self.dbuf(tag + ".tq16", (r, d))
        tnew = self.f32(tag + ".tnew", (r, d))
        for l in range(2):
            lp = f"{pre}.layers.{l}"
            sa = lp + ".self_attn"
            if l == 0:
                L.add_cast(tok32, out16=t16, dt=self.ddti)
                qkv = self.f32(tag + ".sa_qkv", (r, 3 * d))
                L.gemm(t16, p[sa + ".qkv.w"], bias=p[sa + ".qkv.b"], out32=qkv)
                o16 = self._attn_core(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], groups, nt, nt, d, tag + ".sa")
                L.gemm(o16, p[sa + ".out_proj.w"], bias=w[sa + ".out_proj.bias"], out32=tnew)      # replaces the tokens
            else:
                qk = self.f32(tag + ".sa_qk", (r, 2 * d))
                L.gemm(tq16, p[sa + ".qk.w"], bias=p[sa + ".qk.b"], out32=qk)
                v = self.f32(tag + ".sa_v", (r, d))
                L.gemm(t16, p[sa + ".v_proj.w"], bias=w[sa + ".v_proj.bias"], out32=v)
                o16 = self._attn_core(qk[:, :d], qk[:, d:], v, groups, nt, nt, d, tag + ".sa")
                L.gemm(o16, p[sa + ".out_proj.w"], bias=w[sa + ".out_proj.bias"], res=t32, out32=tnew)
            self.dln(tnew, lp + ".norm1", 1e-5, out32=t32, out16=t16, out16_pe=tq16, pe=tpe, pe_mod=0)
            # tokens -> image
            ca = lp + ".cross_attn_token_to_image"
            q = self.f32(tag + ".tq", (r, di))
            L.gemm(tq16, p[ca + ".q_proj.w"], bias=w[ca + ".q_proj.bias"], out32=q)
            o16 = self._t2i(ca, q, img32, pe32, img16, imgpe16, groups, nt, hw, tag)
            L.gemm(o16, p[ca + ".out_proj.w"], bias=w[ca + ".out_proj.bias"], res=t32, out32=tnew)
            self.dln(tnew, lp + ".norm2", 1e-5, out32=t32, out16=t16)
            # MLP (ReLU)
            hbuf = self.dbuf(tag + ".mlp", (r, cfg.dec_mlp))
            L.gemm(t16, p[lp + ".mlp.lin1.w"], bias=w[lp + ".mlp.lin1.bias"], out16=hbuf, act=L.ACT_RELU)
            L.gemm(hbuf, p[lp + ".mlp.lin2.w"], bias=w[lp + ".mlp.lin2.bias"], res=t32, out32=tnew)
            self.dln(tnew, lp + ".norm3", 1e-5, out32=t32, out16=t16, out16_pe=tq16, pe=tpe, pe_mod=0)
            # image -> tokens
            ca = lp + ".cross_attn_image_to_token"
            kt = self.f32(tag + ".tk", (r, di))
            vtok = self.f32(tag + ".tv", (r, di))
            L.gemm(tq16, p[ca + ".k_proj.w"], bias=w[ca + ".k_proj.bias"], out32=kt)
            L.gemm(t16, p[ca + ".v_proj.w"], bias=w[ca + ".v_proj.bias"], out32=vtok)
            if fused:       # q_proj + attention + out_proj + residual + norm4: one read and one write of the stream
                L.twoway_i2t(img32, p[ca + ".q_proj.planes"], self.pe_table(ca + ".q_proj", pe32), kt, vtok, p[ca + ".out_proj.planes"],
                             w[ca + ".out_proj.bias"], w[lp + ".norm4.weight"], w[lp + ".norm4.bias"], 1e-5, groups, hw, nt, cfg.dec_heads)
            else:
                qi = self.f32(tag + ".iq", (ri, di))
                self.igemm(imgpe16, ca + ".q_proj.w", bias=w[ca + ".q_proj.bias"], out32=qi)
                oi16 = self._attn_core(qi, kt, vtok, groups, hw, nt, di, tag + ".i2t", image_side=True)
                self.igemm(oi16, ca + ".out_proj.w", bias=w[ca + ".out_proj.bias"], res=img32, out32=img32)
                L.layernorm(img32, w[lp + ".norm4.weight"], w[lp + ".norm4.bias"], 1e-5, out32=img32, out16=img16, out16_pe=imgpe16, pe=pe32,
                            pe_mod=hw, dt=self.idti)
        if not want_tokens:
            return None, None
        ca = pre + ".final_attn_token_to_image"
        q = self.f32(tag + ".tq", (r, di))
        L.gemm(tq16, p[ca + ".q_proj.w"], bias=w[ca + ".q_proj.bias"], out32=q)
        o16 = self._t2i(ca, q, img32, pe32, img16, imgpe16, groups, nt, hw, tag)
        L.gemm(o16, p[ca + ".out_proj.w"], bias=w[ca + ".out_proj.bias"], res=t32, out32=tnew)
        self.dln(tnew, pre + ".norm_final_attn", 1e-5, out32=t32, out16=t16)
        return t32, t16

    # ------------------------------------------------------------------------------------------------
    # prompt encoder (prompt_encoder.py:564-827)
    # ------------------------------------------------------------------------------------------------
    def _sparse_tokens(self, b, m, c, points, boxes) -> Tuple[Tensor, int]:
        """Bookkeeping of the sparse prompt tokens -> (xy, kind, shift) device arrays.  Index work only, all on device
        (inputs are device tensors) so that the whole forward can be captured in a HIP graph."""
        dev = self.dev
        pcount = b * m * c
        parts_xy, parts_kind, parts_shift = [], [], []
        if points is not None:
            xy, lab = points
            xy = xy.reshape(pcount, -1, 2).to(dev, torch.float32)
            lab = lab.reshape(pcount, -1).to(dev)
            kind = ((lab != 0).to(torch.int32) + (lab > 0).to(torch.int32))      # 0 NULL, 1 negative, 2 positive
            shift = torch.ones_like(kind)
            if boxes is None:   # extra token at (0,0), label -1 == NEGATIVE in this code base, not shifted (prompt_encoder.py:91-95)
                xy = torch.cat([xy, torch.zeros(pcount, 1, 2, device=dev)], dim=1)
                kind = torch.cat([kind, torch.ones(pcount, 1, dtype=torch.int32, device=dev)], dim=1)
                shift = torch.cat([shift, torch.zeros(pcount, 1, dtype=torch.int32, device=dev)], dim=1)
            parts_xy.append(xy); parts_kind.append(kind); parts_shift.append(shift)
        if boxes is not None:
            bx, bf = boxes
            nb = bx.shape[3]
            corners = bx.reshape(pcount, nb * 2, 2).to(dev, torch.float32)
            kind = (torch.arange(2 * nb, device=dev, dtype=torch.int32) % 2 + 3).expand(pcount, -1)
            flags2 = bf.reshape(pcount, nb).to(dev).repeat(1, 2)            # tiled flags vs interleaved corners (:661-667)
            kind = (kind * (flags2 != 0).to(torch.int32)).contiguous()
            parts_xy.append(corners); parts_kind.append(kind); parts_shift.append(torch.ones_like(kind))
        if not parts_xy:
            xy = torch.zeros(pcount, 1, 2, device=dev)
            kind = torch.full((pcount, 1), 5, dtype=torch.int32, device=dev)
            shift = torch.zeros_like(kind)
        else:
            xy, kind, shift = torch.cat(parts_xy, 1), torch.cat(parts_kind, 1), torch.cat(parts_shift, 1)
        ns = kind.shape[1]
        return (xy.contiguous(), kind.contiguous(), shift.contiguous()), ns

    def sample_rows(self, c: int) -> Tensor:
        """RandomMatrixEncoder.sample_rows: a fresh permutation every forward, also in eval (prompt_encoder.py:245-248)."""
        return torch.cat([torch.zeros(1, dtype=torch.long), torch.randperm(self.cfg.bank_size - 1)[: c - 1] + 1])

    def prompt_encoder(self, support32: Tensor, b: int, m: int, g: int, points, boxes, masks, flag_examples: Tensor,
                       selected_rows: Optional[Tensor] = None) -> Dict[str, Tensor]:
        """support32: [B*M*hw, D] NHWC fp32.  Returns class_embeddings (B,C,D), class_examples_embeddings (B,M,C,D),
        class_examples_src (P, hw, D) NHWC."""
        cfg, w, p = self.cfg, self.w32, self.p
        pe_ = "prompt_encoder"
        d = cfg.embed_dim
        first = points[0] if points is not None else boxes[0] if boxes is not None else masks[0] if masks is not None else None
        if first is None:
            raise ValueError("No prompts provided")
        c = first.shape[2]
        pcount = b * m * c
        hw = g * g
        # sparse tokens
        (xy, kind, shift), ns = self._sparse_tokens(b, m, c, points, boxes)
        sp = self.f32("pe.sparse0", (pcount * ns, d))
        L.point_embed(xy, kind, shift, d, cfg.image_size, w[pe_ + ".pe_layer.positional_encoding_gaussian_matrix"],
                      p[pe_ + ".type_emb"], w[pe_ + ".not_a_point_embed.weight"], w[pe_ + ".no_sparse_embedding.weight"], sp)
        sp = self.attention_mlp_block(pe_ + ".sparse_embedding_attention", sp, b * m, c * ns, "pe.sea")
        class_enc = None
        if cfg.bank_size:
            if selected_rows is None:
                selected_rows = self.sample_rows(c)
            class_enc = w[pe_ + ".class_encoder.pos_embedding"][0, 0].index_select(0, self.h2d(selected_rows)).contiguous()
            ce_rows = class_enc.repeat_interleave(ns, dim=0).contiguous()       # rows ordered (c, n)
            sp2 = self.f32("pe.sparse_ce", (pcount * ns, d))
            L.add_cast(sp, ce_rows, c * ns, out32=sp2, dt=self.ddti)
            sp = sp2
        # dense stream
        pe32 = self.dense_pe(g)
        src32 = self.f32("pe.src32", (pcount * hw, d))
        fused = self.fused_ok(ns)           # fused two-way kernels read the fp32 stream itself: no GEMM-operand copies
        src16 = None if fused else self.ibuf("pe.src16", pcount * hw, d)
        srcpe16 = None if fused else self.ibuf("pe.srcpe16", pcount * hw, d)
        if masks is not None:
            mk, mf = masks
            mk = self.h2d(mk, torch.float32).reshape(pcount, mk.shape[-2], mk.shape[-1]).contiguous()
            if mk.shape[-1] != mk.shape[-2]:
                raise ValueError("prompt masks must be square")
            mf = self.h2d(mf.reshape(pcount), torch.int32).contiguous()
            L.mask_embed(mk, mf, pcount, c, mk.shape[-1], g, d, p[pe_ + ".mask_w"], support32, class_enc, pe32, src32, src16,
                         srcpe16, self.idti)
        else:
            L.mask_embed(None, None, pcount, c, 0, g, d, p[pe_ + ".mask_w"], support32, class_enc, pe32, src32, src16, srcpe16,
                         self.idti)
        self.two_way(pe_ + ".transformer", sp, pcount, ns, src32, src16, srcpe16, hw, pe32, "pe.tw", want_tokens=False)
        emb = self.f32("pe.emb", (pcount, d))
        L.colmean(src32, pcount, hw, d, emb, self.f32("pe.colmean.part", (pcount, L.COLMEAN_SPLIT, d)))
        if cfg.class_attention:
            emb = self.attention_mlp_block(pe_ + ".class_attention", emb, b * m, c, "pe.ca")
        if cfg.example_attention:
            e2 = emb.view(b, m, c, d).permute(0, 2, 1, 3).contiguous().view(b * c * m, d)
            e2 = self.attention_mlp_block(pe_ + ".example_attention", e2, b * c, m, "pe.ea")
            emb = e2.view(b, c, m, d).permute(0, 2, 1, 3).contiguous().view(pcount, d)
        if cfg.example_class_attention:
            emb = self.attention_mlp_block(pe_ + ".class_example_attention", emb, b, m * c, "pe.cea")
        fe = self.h2d(flag_examples.reshape(b, m, c), torch.uint8).contiguous()
        cls = torch.empty(b, c, d, device=self.dev, dtype=torch.float32)
        L.class_mean(emb, fe, b, m, c, d, cls)
        return {"flag_examples": flag_examples, "class_embeddings": cls,
                "class_examples_embeddings": emb.view(b, m, c, d).clone(), "class_examples_src": src32.view(pcount, hw, d)}

    # ------------------------------------------------------------------------------------------------
    # mask decoder (mask_decoder.py:316-363)
    # ------------------------------------------------------------------------------------------------
    def mask_decoder(self, query32: Tensor, b: int, g: int, class_emb: Tensor) -> Tensor:
        """query32 [B*hw, D] NHWC fp32, class_emb (B,C,D) fp32 -> low-res logits (B, C, 4g, 4g) fp32."""
        cfg, w, p = self.cfg, self.w32, self.p
        md = "mask_decoder"
        d = cfg.embed_dim
        hw = g * g
        c = class_emb.shape[1]
        pe32 = self.dense_pe(g)
        img32 = self.f32("md.img32", (b * hw, d))
        fused = self.fused_ok(c)
        img16 = imgpe16 = None
        if fused:
            L.add_cast(query32, out32=img32, dt=L.LA_F32)
        else:
            img16 = self.ibuf("md.img16", b * hw, d)
            imgpe16 = self.ibuf("md.imgpe16", b * hw, d)
            L.add_cast(query32, out32=img32, out16=img16, dt=self.idti)
            L.add_cast(query32, pe32, hw, out16=imgpe16, dt=self.idti)
        tok = self.h2d(class_emb, torch.float32).reshape(b * c, d).contiguous()
        t32, t16 = self.two_way(md + ".transformer", tok, b, c, img32, img16, imgpe16, hw, pe32, "md.tw", want_tokens=True)
        # class_mlp (3 x Linear, ReLU between) -> prototypes
        h1 = self.dbuf("md.cm1", (b * c, d))
        L.gemm(t16, p[md + ".class_mlp.0.w"], bias=w[md + ".class_mlp.layers.0.bias"], out16=h1, act=L.ACT_RELU)
        h2 = self.dbuf("md.cm2", (b * c, d))
        L.gemm(h1, p[md + ".class_mlp.1.w"], bias=w[md + ".class_mlp.layers.1.bias"], out16=h2, act=L.ACT_RELU)
        cf = d // 8
        protos = self.f32("md.protos", (b * c, cf))
        L.gemm(h2, p[md + ".class_mlp.2.w"], bias=w[md + ".class_mlp.layers.2.bias"], out32=protos)
        # output_upscaling: ConvT(k2,s2) -> LN2d -> GELU -> ConvT(k2,s2), both as pixel-shuffle GEMMs
        c1 = d // 4
        up1 = self.f32("md.up1", (b * 4 * hw, c1))
        if fused:       # the stream itself is the (fp32) operand: there is no separate copy to read
            L.gemm(img32, p[md + ".up0.w"], bias=w[md + ".output_upscaling.0.bias"], out32=up1, map=L.MAP_CONVT2X2, p=(g, g, c1, 0, 0))
        else:
            self.igemm(img16, md + ".up0.w", bias=w[md + ".output_upscaling.0.bias"], out32=up1, map=L.MAP_CONVT2X2, p=(g, g, c1, 0, 0))
        up1h = self.ibuf("md.up1h", b * 4 * hw, c1)
        L.layernorm(up1, w[md + ".output_upscaling.1.weight"], w[md + ".output_upscaling.1.bias"], 1e-6, gelu=True, out16=up1h, dt=self.idti)
        npix = 16 * hw
        feat32 = self.f32("md.feat32", (b * npix, cf))
        feat16 = self.f32("md.feat16f", (b * npix, cf)) if self.isplit else self.dbuf("md.feat16", (b * npix, cf))
        self.igemm(up1h, md + ".up3.w", bias=w[md + ".output_upscaling.3.bias"], out32=feat32, out16=None if self.isplit else feat16,
                   map=L.MAP_CONVT2X2, p=(2 * g, 2 * g, cf, 0, 0))
        if cfg.spatial_convs:
            implicit = self.ddt == torch.float32 and cf % 32 == 0
            col = None if implicit else self.dbuf("md.col", (b * npix, 9 * cf))
            if self.isplit:       # the up3 GEMM cannot emit a second fp32 copy in split mode: the first conv reads its fp32 output
                L.add_cast(feat32, out32=feat16, dt=L.LA_F32)
            for i in range(cfg.spatial_convs):
                if implicit:      # fp32 implicit GEMM: no im2col buffer (feat16 IS fp32 here)
                    nxt = self.f32("md.feat32b" if i % 2 == 0 else "md.feat32", (b * npix, cf))
                    # (32 -> 32 channels, the D = 256 decoder: three fp16 products on plane pairs - fp32-class accuracy at ~3x the rate of
                    # the exact-fp32 MFMA, which bounds la_conv3x3_f32; other widths keep that kernel)
                    conv = L.conv3x3_split if (self.conv_split and L.conv3x3_split_ok(cf, cf)) else L.conv3x3_f32
                    conv(feat16, b, 4 * g, 4 * g, cf, p[f"{md}.sc{i}.w"], w[f"{md}.spatial_convs.{3 * i}.bias"], cf, nxt)
                    feat32 = nxt
                else:
                    L.im2col_3x3(feat16, b, 4 * g, 4 * g, cf, col)
                    L.gemm(col, p[f"{md}.sc{i}.w"], bias=w[f"{md}.spatial_convs.{3 * i}.bias"], out32=feat32)
                if i < cfg.spatial_convs - 1:
                    self.dln(feat32, f"{md}.spatial_convs.{3 * i + 1}", 1e-6, gelu=True, out16=feat16)
        seg = torch.empty(b, c, 4 * g, 4 * g, device=self.dev, dtype=torch.float32)
        L.classify(feat32, protos, b, npix, c, cf, seg)
        return seg

    # ------------------------------------------------------------------------------------------------
    # post-processing (lam.py:383-453, 92-93)
    # ------------------------------------------------------------------------------------------------
    def post_sizes(self, dims: Tensor):
        """Host side of postprocess_masks: per-item (orig_h, orig_w, crop_h, crop_w) + the padded output frame."""
        cfg = self.cfg
        s = cfg.image_size
        # plain Python on the (tiny) dims list: torch CPU reductions wake the whole intra-op thread pool (measured
        # ~19 ms per call on a 128-thread host) which would dwarf the GPU time of a small episode batch
        dl = dims.detach().to("cpu").tolist()                      # [B][M+1][2]
        hmax = max(int(hw[0]) for item in dl for hw in item)
        wmax = max(int(hw[1]) for item in dl for hw in item)
        sizes = []
        for item in dl:
            oh, ow = int(item[0][0]), int(item[0][1])
            if cfg.custom_preprocess:
                sc = s * 1.0 / max(oh, ow)
                ph, pw = int(oh * sc + 0.5), int(ow * sc + 0.5)
            else:
                ph, pw = s, s
            sizes.append([oh, ow, ph, pw])
        return torch.tensor(sizes, dtype=torch.int32), hmax, wmax

    def postprocess_dev(self, seg: Tensor, sizes_d: Tensor, hmax: int, wmax: int, flag_gts_u8: Optional[Tensor], want_argmax: bool):
        """Device side (capturable): bilinear to S x S, crop + resample + pad + flag_gts (+ argmax)."""
        b, c, h, wd = seg.shape
        s = self.cfg.image_size
        big = self.f32("post.big", (b * c, s, s))
        L.bilinear(seg, b * c, h, wd, s, s, big)
        logits = torch.empty(b, c, hmax, wmax, device=self.dev, dtype=torch.float32)
        am = torch.empty(b, hmax, wmax, device=self.dev, dtype=torch.int64) if want_argmax else None
        L.post_final(big, b, c, s, sizes_d, flag_gts_u8, hmax, wmax, logits, am)
        return logits, am

    def postprocess(self, seg: Tensor, dims: Tensor, flag_gts: Optional[Tensor] = None, want_argmax: bool = False):
        sizes, hmax, wmax = self.post_sizes(dims)
        fg = self.h2d(flag_gts, torch.uint8).contiguous() if flag_gts is not None else None
        logits, am = self.postprocess_dev(seg.contiguous(), self.h2d(sizes), hmax, wmax, fg, want_argmax)
        return (logits, am) if want_argmax else logits
