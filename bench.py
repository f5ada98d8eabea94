#!/usr/bin/env python
"""bench.py - BASELINE metric: episodes/sec (query+support forward), SAM ViT-B 1024 px, 1-way 1-shot, on N MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One step = one ``Lam.forward`` over a batch of ``--episodes`` synthetic 1-way 1-shot episodes (BASELINE cfg2: encoder on
query + support image, prompt encoder, mask decoder, post-processed full-resolution logits).  Episodes are independent,
so for N > 1 every rank runs its own batch (weak scaling, no data-path collective); the timed region is bracketed by a
barrier + device sync and the MAX over ranks is reported.  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline      the dominant kernel (la_gemm: every Linear / conv of the path) timed live with HIP events on the launch
                stream in a separate instrumented step: algorithmic FLOPs of all its launches / their summed duration,
                against the gfx950 dense fp16/bf16 MFMA peak (2.5 PFLOP/s, MI355X_MICROARCH.md).  `traffic` = HBM-side
                bytes per launch from the committed rocprofv3 PMC passes (profiles/r*_traffic.json).
  cpu_baseline  the CPU oracle (oracle/lam_oracle.py, a checked restatement of the reference's torch path) timed on this
                box's host cores for ONE episode of the same workload (N=1, rank 0 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_MFMA_TFLOPS = 2500.0      # dense fp16/bf16, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"


# BASELINE.json configs as bench workloads.  cfg2 is the one the metric is quoted on (the default and the only one the driver
# runs); the others are extra data points (north_star: "throughput on synthetic 1024x1024 / 480x480 episodes").
WORKLOADS = {
    "cfg2": dict(desc="BASELINE cfg2: SAM ViT-B 1024px encoder + LabelAnything decoder, 1-way 1-shot episodes (2 images each)",
                 model=dict(encoder="vit_b", image_size=1024, image_embed_dim=256, embed_dim=256, spatial_convs=3, custom_preprocess=False),
                 episode=dict(n_ways=1, k_shots=1, image_size=1024), default_episodes=48),
    "cfg1": dict(desc="BASELINE cfg1 geometry on the GPU: ViT-MAE-B 480px encoder + decoder, 1-way 1-shot episodes (2 images each)",
                 model=dict(encoder="vit_b_mae", image_size=480, image_embed_dim=768, embed_dim=256, spatial_convs=3,
                            example_class_attention=False, custom_preprocess=False),
                 episode=dict(n_ways=1, k_shots=1, image_size=480), default_episodes=32),
    "cfg3": dict(desc="BASELINE cfg3 geometry, the forward pass alone (its training step: cfg3_train): ViT-MAE-B 480px, 5-way 5-shot episodes (26 images, 150 prompt pairs each)",
                 model=dict(encoder="vit_b_mae", image_size=480, image_embed_dim=768, embed_dim=256, spatial_convs=3,
                            class_encoder={"name": "RandomMatrixEncoder", "bank_size": 100, "embed_dim": 256}, custom_preprocess=False),
                 episode=dict(n_ways=5, k_shots=5, image_size=480), default_episodes=2),
    "cfg4": dict(desc="BASELINE cfg4 geometry: precomputed 256x64x64 embeddings (encoder bypassed), 2-way 5-shot episodes, decoder only",
                 model=dict(encoder=None, use_vit=False, image_size=1024, image_embed_dim=256, embed_dim=256, spatial_convs=3,
                            custom_preprocess=False),
                 episode=dict(n_ways=2, k_shots=5, image_size=1024, embeddings_channels=256, grid=64), default_episodes=8),
    "cfg5": dict(desc="BASELINE cfg5 geometry, forward: ViT-MAE-L 480px (parameters/trainval/coco/mael.yaml), 10-way 5-shot episodes "
                      "(51 images, 550 prompt pairs each); 16-bit MFMA attention by default, --attn-fp8 runs QK^T on the fp8 MFMA as BASELINE "
                      "configs[4] words it (3e-3 on the logits: outside the 1e-3 tolerance, hence opt-in)",
                 model=dict(encoder="vit_l_mae", image_size=480, image_embed_dim=1024, embed_dim=256, spatial_convs=3,
                            class_encoder={"name": "RandomMatrixEncoder", "bank_size": 100, "embed_dim": 256}, custom_preprocess=False),
                 episode=dict(n_ways=10, k_shots=5, image_size=480), default_episodes=1),
    # BASELINE cfg3 as it is trained (mae_noembs.yaml): one optimizer step per episode batch = frozen-encoder forward, decoder forward,
    # focal objective, backward, flat-gradient SUM all-reduce over the ranks (RCCL), AdamW.  value = episodes/s of full training steps.
    "cfg3_train": dict(desc="BASELINE cfg3 TRAINING step: ViT-MAE-B 480px (frozen), 5-way 5-shot episodes, focal objective, backward through "
                            "neck + prompt encoder + mask decoder (10.14 M parameters), gradient all-reduce, AdamW",
                       model=dict(encoder="vit_b_mae", image_size=480, image_embed_dim=768, embed_dim=256, spatial_convs=3,
                                  class_encoder={"name": "RandomMatrixEncoder", "bank_size": 100, "embed_dim": 256}, custom_preprocess=False),
                       episode=dict(n_ways=5, k_shots=5, image_size=480), default_episodes=2, train=True),     # the reference batches 2-16 episodes (mae_noembs.yaml:94)
    # the headline geometry (cfg2) as a TRAINING step: SAM ViT-B 1024 with nothing frozen when run with --train-encoder (lam_b,
    # models/lam.py:321-347: window + global attention with decomposed rel-pos, SAM neck) - round 4's SAM-stack backward
    "cfg2_train": dict(desc="cfg2 geometry as a TRAINING step: SAM ViT-B 1024px + LabelAnything decoder, 1-way 1-shot episodes, focal objective, "
                            "backward, gradient all-reduce, AdamW (--train-encoder: the SAM ViTDet stack trains too)",
                       model=dict(encoder="vit_b", image_size=1024, image_embed_dim=256, embed_dim=256, spatial_convs=3, custom_preprocess=False),
                       episode=dict(n_ways=1, k_shots=1, image_size=1024), default_episodes=2, train=True),
}


def build_model(dtype, decoder_dtype=torch.float32, workload="cfg2", precise=None):
    from labelanything_amd.config import LamConfig
    from labelanything_amd.models import Lam
    cfg = LamConfig(**WORKLOADS[workload]["model"])
    return Lam(cfg, seed=2, compute_dtype=dtype, decoder_dtype=decoder_dtype, precise="auto" if precise is None else precise), cfg


def make_inputs(episodes: int, seed: int, device, workload="cfg2"):
    from labelanything_amd.episodes import make_episode
    batch = make_episode(batch=episodes, seed=seed, prompts=("mask",), **WORKLOADS[workload]["episode"])
    dev_keys = ("images", "embeddings", "prompt_masks")
    return {k: (v.to(device) if k in dev_keys else v) for k, v in batch.items()}


class KernelTimer:
    """Wraps labelanything_amd._lib launch functions with HIP events on the current (launch) stream."""

    def __init__(self, by_shape: bool = False):
        self.records = []
        self.by_shape = by_shape

    def __enter__(self):
        from labelanything_amd import _lib as L
        self.L = L
        self.saved = {}
        names = ["twoway_t2i", "twoway_i2t", "gemm", "layernorm", "im2col_patch", "im2col_3x3", "relpos_terms", "attn_fwd", "mask_embed", "attn_small",
                 "colmean", "class_mean", "classify", "add_cast", "bilinear", "post_final", "point_embed", "nchw_to_nhwc",
                 "conv3x3_f32", "nhwc_to_nchw", "dense_pe", "gemm_tn", "layernorm_bwd", "act_fwd", "act_bwd", "attn_small_lse",
                 "attn_small_bwd", "bilinear_bwd", "bilinear_bwd_set", "bilinear_rows", "bilinear_rows_bwd_set", "classify_bwd", "row_broadcast", "focal_loss", "adamw_step", "colsum_acc",
                 "attn_fwd_lse", "attn_bwd", "head_transpose", "cast", "gelu_bwd16", "axpy", "transpose16", "colmean16", "layernorm_g", "add_rowvec", "add_rowvec_split", "qk_fp8", "attn_fwd_fp8", "attn_fwd_cs", "colsum_fold", "gelu_fwd16",
                 "attn_fwd_relpos_lse", "attn_bwd_relpos", "relpos_bwd", "layernorm_bwd_res", "transpose_many", "attn_fwd_rows", "gemm_tn16",
                 "norm_finalize", "norm_stats", "conv3x3_split"]
        for n in names:
            fn = getattr(L, n)
            self.saved[n] = fn

            def wrapped(*a, _fn=fn, _n=n, **kw):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                _fn(*a, **kw)
                e.record()
                flops = 0.0
                nbytes = 0.0
                issued = 0.0
                if _n == "gemm":
                    m = kw.get("M") or a[0].shape[0]
                    n, k = a[1].shape
                    issued = 2.0 * m * n * k                     # MFMA work actually issued (split-precision planes count twice)
                    k = kw.get("a_kmod") or k                    # algorithmic K: the [W_hi | W_lo] planes are ONE weight
                    flops = 2.0 * m * n * k
                    esz = a[0].element_size()
                    # algorithmic bytes: A and W read once, every output written once, the residual read once
                    nbytes = esz * (m * k + n * k) + m * n * (4 * (kw.get("out32") is not None) + esz * (kw.get("out16") is not None)
                                                              + 4 * (kw.get("res") is not None and kw.get("res_mod", 0) == 0))
                    if kw.get("vt") is not None:
                        nbytes += esz * m * (n - kw.get("vt_col0", 0))
                    if kw.get("aux16") is not None and kw.get("nstat_out") is not None:
                        nbytes += esz * m * n                    # the lo plane written beside out16 (round 6: plane-pair stream)
                        if kw.get("out32") is None and kw.get("res") is None:
                            nbytes += 2 * esz * m * n            # ... and both planes read as the residual (in-place read-modify-write)
                elif _n in ("gemm_tn", "gemm_tn16"):
                    flops = issued = 2.0 * a[0].shape[0] * a[0].shape[1] * a[1].shape[1]
                elif _n == "attn_fwd":
                    b, heads, t = a[5], a[6], a[7]
                    flops = 4.0 * b * heads * t * t * 64
                elif _n == "attn_fwd_rows":                  # (qkv, out16, b, heads, t, ...): no V^T copy; windows in image order
                    b, heads, t = a[2], a[3], a[4]
                    flops = 4.0 * b * heads * t * t * 64
                elif _n == "attn_fwd_lse":
                    flops = 4.0 * a[4] * a[5] * a[6] * a[6] * 64
                elif _n == "attn_bwd":                       # 7 T x T x 64 products (S and dP twice: one kernel per output side)
                    flops = 14.0 * a[9] * a[10] * a[11] * a[11] * 64
                elif _n in ("twoway_t2i", "twoway_i2t"):     # the (groups, hw, D) fp32 stream: read once (t2i), read + written (i2t)
                    nbytes = a[0].numel() * 4.0 * (2 if _n == "twoway_i2t" else 1)
                tag = _n
                if _n == "gemm" and self.by_shape:
                    tag = f"gemm[{m}x{a[1].shape[0]}x{a[1].shape[1]},{str(a[0].dtype)[6:]}]"
                if _n in ("attn_small_bwd", "attn_small", "attn_small_lse", "layernorm_bwd", "colsum_acc", "add_cast", "act_bwd", "act_fwd") and self.by_shape:
                    shp = [x for x in a if isinstance(x, int)][:4]
                    tag = f"{_n}{[tuple(a[0].shape)] + shp}"
                if _n in ("gemm_tn", "gemm_tn16") and self.by_shape:
                    tag = f"{_n}[{a[0].shape[0]}x{a[0].shape[1]}x{a[1].shape[1]}]"
                if _n == "attn_fwd" and self.by_shape:
                    tag = f"attn_fwd[{a[5]}x{a[6]}x{a[7]}]"
                if _n == "attn_fwd_rows" and self.by_shape:
                    tag = f"attn_fwd_rows[{a[2]}x{a[3]}x{a[4]}]"
                self.records.append((tag, flops, s, e, nbytes, issued if _n in ("gemm", "gemm_tn", "gemm_tn16") else flops))
            setattr(L, n, wrapped)
        return self

    def __exit__(self, *exc):
        for n, fn in self.saved.items():
            setattr(self.L, n, fn)

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for n, fl, s, e, nb, iss in self.records:
            d = agg.setdefault(n, [0, 0.0, 0.0, 0.0, 0.0])
            d[0] += 1
            d[1] += s.elapsed_time(e) * 1e-3
            d[2] += fl
            d[3] += nb
            d[4] += iss
        return agg


def pmc_traffic(workload: str, precise, episodes=None):
    """HBM-side bytes per la_gemm launch from the committed PMC passes (profiles/r*_traffic.json, written by tools/collect_profiles.sh
    with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE).  A file only applies to the workload AND the numerics configuration it was
    collected on (it records both); anything else reports null instead of a stale number."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic*.json")), reverse=True):
        with open(path) as f:
            d = json.load(f)
        if d.get("workload", "cfg2") == workload and d.get("encoder_split_precision") == list(precise) and \
                (episodes is None or d.get("episodes_per_step", 32) == episodes):       # (files of rounds 1 - 2: 32 episodes per step)
            return d.get("bytes_per_launch")
    return None


def host_cpu():
    """(model name, physical cores, logical CPUs) of this box from /proc/cpuinfo."""
    model, cores, logical = "unknown", set(), 0
    try:
        phys = core = None
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                elif line.startswith("processor"):
                    logical += 1
                elif line.startswith("physical id"):
                    phys = line.split(":", 1)[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":", 1)[1].strip()
                    cores.add((phys, core))
    except OSError:
        pass
    return model, (len(cores) or logical), logical


def cpu_baseline(cfg, timed: int = 3):
    """Time the CPU oracle (kind "port": the checked restatement of the reference's torch path) on this box's host cores: one warm-up
    episode + the median of ``timed`` episodes of the bench workload, torch intra-op threads = physical cores."""
    from labelanything_amd.episodes import make_episode
    from labelanything_amd.weights import init_state_dict
    from oracle import lam_oracle as O
    from tests.cases import geometry_for
    sd = init_state_dict(cfg, 2)
    geo = geometry_for(cfg)
    model, phys, logical = host_cpu()
    torch.set_num_threads(max(1, phys))
    times = []
    with torch.no_grad():
        for i in range(timed + 1):
            batch = make_episode(batch=1, n_ways=1, k_shots=1, image_size=1024, seed=1234 + i, prompts=("mask",))
            t0 = time.perf_counter()
            O.lam_forward(sd, geo, batch)
            if i:
                times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"value": round(1.0 / med, 5), "unit": "episodes/s", "cores": torch.get_num_threads(), "kind": "port",
            "cpu": model, "physical_cores": phys, "logical_cpus": logical,
            "sample": f"1 warm-up + median of {timed} episodes (2 images 1024x1024 each) of the bench workload, fp32 torch CPU oracle, "
                      f"{med:.1f} s per episode (all: {', '.join(f'{t:.1f}' for t in times)})"}


def eager_baseline(cfg, dev, episodes: int = 8, warm: int = 2, it: int = 5):
    """north_star's comparator ("the reference single-GPU PyTorch-eager images/sec"): the reference is not on the GPU box, so the
    checked fp32 restatement of its eager torch op sequence (the oracle, pinned on the reference to 2e-6) runs ON this MI355X with
    stock torch / rocBLAS / MIOpen kernels, outside the timed region like cpu_baseline.  Two legs, as BASELINE.md 3 names them: fp32
    (the reference's own precision: `vs_baseline`) and the same op sequence under `torch.autocast(dtype=float16)` (the stronger
    comparator: `vs_eager_autocast16`).  A baseline leg only - never the product."""
    from labelanything_amd.episodes import make_episode
    from labelanything_amd.weights import init_state_dict
    from oracle import lam_oracle as O
    from tests.cases import geometry_for
    sd = {k: v.to(dev) for k, v in init_state_dict(cfg, 2).items()}
    geo = geometry_for(cfg)
    batch = {k: v.to(dev) for k, v in make_episode(batch=episodes, n_ways=1, k_shots=1, image_size=1024, seed=1234, prompts=("mask",)).items()}
    torch.set_default_device(dev)             # the restatement builds a few helper tensors on the default device

    def leg():
        for _ in range(warm):
            O.lam_forward(sd, geo, batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(it):
            O.lam_forward(sd, geo, batch)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / it

    dt16, err16 = None, None
    try:
        with torch.no_grad():
            dt = leg()
            try:
                with torch.autocast(device_type="cuda", dtype=torch.float16):
                    dt16 = leg()
            except Exception as ex:           # (the reference breaks under plain half casts; autocast has always run so far)
                err16 = f"{type(ex).__name__}: {ex}"
    finally:
        torch.set_default_device("cpu")
    batch_note = (f"{episodes} episodes per eager forward (the product's timed step runs its own, larger batch - `episodes_per_step_per_gpu`; "
                  f"eager fp32 at 1024 px keeps ~2.4 GB of activations per image and gains nothing from larger batches)")
    out = {"value": round(episodes / dt, 3), "unit": "episodes/s", "images_per_sec": round(2 * episodes / dt, 2),
           "kind": "torch-eager fp32 restatement of the reference's op sequence on the same MI355X (stock torch kernels)",
           "sample": f"{warm} warm-up + {it} forwards of {episodes} episodes (2 images 1024x1024 each), {dt * 1e3:.1f} ms per forward; {batch_note}"}
    if dt16 is not None:
        out["autocast16"] = {"value": round(episodes / dt16, 3), "unit": "episodes/s", "images_per_sec": round(2 * episodes / dt16, 2),
                             "kind": "the same op sequence under torch.autocast(device_type='cuda', dtype=torch.float16), stock torch kernels, same MI355X",
                             "sample": f"{warm} warm-up + {it} forwards of {episodes} episodes, {dt16 * 1e3:.1f} ms per forward"}
    elif err16:
        out["autocast16"] = {"value": None, "error": err16}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--episodes", type=int, default=None, help="episodes per step per GPU (cfg2 default 48 = 96 images, 45 GB of the 288 GB; rounds 1 / 2 used 16 / 32; same box, round 3: "
                    "335 / 341 / 340 episodes/s at 32 / 48 / 64)")
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS), help="cfg2 = the BASELINE metric (default); the others are "
                    "extra data points with the geometry of the other BASELINE configs")
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16"])
    ap.add_argument("--decoder", default="f32", choices=["f32", "f16x2", "same"], help="operand type of the decoder-side GEMMs: exact-fp32 "
                    "MFMA, fp16 plane pairs on the image side (3 fast-MFMA products, fp32-level accuracy), or the encoder's 16-bit type")
    ap.add_argument("--precise", default="default", help="encoder GEMM groups in split precision: 'default' (the parity-tested "
                    "configuration, engine.resolve_precise), 'none' (plain 16-bit operands everywhere: faster, misses the 1e-3 logit tolerance), "
                    "or a comma list of groups")
    ap.add_argument("--train-encoder", action="store_true", help="cfg3_train only: train the ViT backbone too (mae_noembs.yaml has no "
                    "freeze_backbone): forward with saved activations + encoder backward + 96 M-parameter gradient all-reduce")
    ap.add_argument("--attn-fp8", action="store_true", help="HF encoders (cfg1/3/5): QK^T of the attention on the fp8 (e4m3) MFMA, as BASELINE "
                    "configs[4] words cfg5; 3e-3 on the logits, i.e. outside the 1e-3 tolerance - never the default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-eager-baseline", action="store_true", help="skip the torch-eager-on-GPU comparator that fills vs_baseline (cfg2, 1 GPU)")
    ap.add_argument("--no-graphs", action="store_true", help="launch every kernel eagerly instead of replaying a HIP graph")
    ap.add_argument("--no-conv-implicit", action="store_true", help="A/B: the necks' 3 x 3 convolution as im2col + GEMM instead of the implicit GEMM on "
                    "zero-bordered plane-pair maps (LamEngine.conv_implicit, round 6)")
    ap.add_argument("--no-win-fused-cs", action="store_true", help="A/B: the window blocks' output token means from a la_colmean16 pass instead of "
                    "the attention epilogue's column sums (LamEngine.win_fused_cs, round 6)")
    ap.add_argument("--no-norm-fold", action="store_true", help="A/B: keep the LayerNorm kernels of the encoder blocks instead of folding them into "
                    "the neighbour GEMMs (LamEngine.norm_fold, round 6)")
    ap.add_argument("--gemm-shapes", action="store_true", help="print a per-shape breakdown of the GEMM launches to stderr")
    a = ap.parse_args()
    if a.episodes is None:
        a.episodes = WORKLOADS[a.workload]["default_episodes"]

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (torch.cuda.is_available() is False)")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("LA_BENCH_BACKEND", "nccl")      # "nccl" is RCCL on ROCm; "gloo" only for the 1-GPU smoke test
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: launch the N ranks ourselves (one process per GPU, rendezvous on the loopback address) - the
        # reference starts its ranks the same way (`accelerate launch --multi_gpu`, /root/reference/slurm/launch_run_exe:5)
        if a.gpus > ndev and backend != "gloo":
            raise SystemExit(f"bench.py --gpus {a.gpus}: only {ndev} GPU(s) visible (LA_BENCH_BACKEND=gloo shares one GPU between the ranks: smoke tests only)")
        import subprocess
        port = os.environ.get("MASTER_PORT", "29517")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus N` or under torch.distributed.run "
                         f"with --nproc-per-node N")
    torch.cuda.set_device(local % ndev)              # one rank per GPU on a real node; wraps only in the 1-GPU smoke test
    dev = torch.device("cuda", local % ndev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    dtype = torch.float16 if a.dtype == "f16" else torch.bfloat16
    precise = None if a.precise == "default" else (() if a.precise == "none" else tuple(a.precise.split(",")))
    lam, cfg = build_model(dtype, {"f32": torch.float32, "f16x2": "f16x2", "same": None}[a.decoder], a.workload, precise)
    lam = lam.to(dev)
    train = bool(WORKLOADS[a.workload].get("train"))
    lam.attn_fp8 = bool(a.attn_fp8)
    lam.norm_fold = not a.no_norm_fold
    if a.no_conv_implicit:
        lam.engine().conv_implicit = False
    if a.no_win_fused_cs:
        lam.engine().win_fused_cs = False
    lam.use_graphs = not a.no_graphs and not train
    batch = make_inputs(a.episodes, 1234 + rank, dev, a.workload)
    if train:
        from labelanything_amd.train import LamTrainer
        trainer = LamTrainer(lam, lr=5e-5, num_warmup_steps=1000, train_encoder=a.train_encoder)           # mae_noembs.yaml:30-37
        c = batch["flag_examples"].shape[2]
        gt = torch.randint(0, c, (a.episodes, WORKLOADS[a.workload]["episode"]["image_size"], WORKLOADS[a.workload]["episode"]["image_size"]),
                           generator=torch.Generator().manual_seed(99 + rank)).to(dev)
        lam_fwd = lam
        lam = lambda bt: trainer.step(bt, gt)                               # noqa: E731  one full optimizer step

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        lam(batch)
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = lam(batch)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert torch.isfinite(out["logits"][out["logits"] > float("-inf")]).all()

    # instrumented step: per-kernel HIP-event timing (outside the timed region)
    roof = None
    kernels = None
    if rank == 0:
        if not train:
            lam.use_graphs = False      # per-kernel events need eager launches
        with KernelTimer() as kt:
            lam(batch)
        agg = kt.summary()
        if a.gemm_shapes:
            with KernelTimer(by_shape=True) as kt2:
                lam(batch)
            for n, v in sorted(kt2.summary().items(), key=lambda kv: -kv[1][1]):
                if n.startswith("gemm") or n.startswith("attn") or n.startswith("layernorm_bwd") or n.startswith("colsum") or n.startswith("add_cast") or n.startswith("act_"):
                    print(f"{n:44s} x{v[0]:3d}  {v[1]*1e3:8.3f} ms  {v[2]/max(v[1],1e-12)/1e12:7.1f} TF/s", file=sys.stderr)
        g = agg.get("gemm")
        tot = sum(v[1] for v in agg.values())
        kernels = {n: {"launches": v[0], "ms": round(v[1] * 1e3, 3)} for n, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}
        if g:
            ach = g[2] / g[1] / 1e12
            roof = {"kernel": "la_gemm (gemm_t256w_kernel - four waves x 512 registers, 256 x 256 x 64 - on the encoder shapes, gemm_t256p_kernel for two-plane weights; gemm_t256q_kernel / gemm_t256_kernel / gemm_dma4_kernel / gemm_pp_kernel / gemm_dma_kernel / gemm_f32_kernel / gemm_skinny_kernel elsewhere)", "bound": "mfma",
                    "achieved": round(ach, 1), "peak": PEAK_MFMA_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(ach / PEAK_MFMA_TFLOPS, 4), "traffic": pmc_traffic(a.workload, (lam_fwd if train else lam).precise, a.episodes),
                    "flop_per_launch": round(g[2] / g[0]), "algorithmic_bytes_per_launch": round(g[3] / g[0]), "launches_per_step": g[0], "avg_launch_us": round(g[1] / g[0] * 1e6, 2),
                    "share_of_kernel_time": round(g[1] / tot, 3),
                    "peak_note": ("2.5 PFLOP/s is the dense fp16 figure at the 2.4 GHz peak clock; under this kernel the part holds 1.45 - 1.5 GHz "
                                  "(s_memtime stamps against wall time, profiles/r05_notes.md 1; matrix pipe busy 0.73 - 0.77 of the REAL cycles) - the "
                                  "vendor GEMM without any epilogue reaches 1.54 PFLOP/s = 0.61 on the same 8192^3 product, la_gemm 1.32"),
                    # split-precision weights ([W_hi | W_lo]) issue two MFMA passes for one algorithmic product
                    "mfma_issued_tflops": round(g[4] / g[1] / 1e12, 1)}
            # round 6: with LamEngine.norm_fold the encoder's LayerNorm passes live in these launches' epilogues (16-bit copy of the stream,
            # row sums, normalisation of the product) - the same MFMA work over a longer time.  A second instrumented step with the
            # LayerNorm kernels puts the two accountings side by side on this box (outside the timed region, like the first)
            if not train and getattr(lam._engine, "norm_fold", False):
                lam.norm_fold = False
                lam(batch)                                   # (the other launch sequence once untimed: arena buffers, packed weights in cache)
                with KernelTimer() as kt3:
                    lam(batch)
                lam.norm_fold = True
                lam.engine()                                 # (re-syncs the engine's switch: the line below reports the timed configuration)
                a3 = kt3.summary()
                g3 = a3.get("gemm")
                ln_ms = lambda ag: round(sum(v[1] for n, v in ag.items() if n in ("layernorm_g", "norm_finalize", "norm_stats", "add_rowvec", "add_rowvec_split")) * 1e3, 3)
                roof["norm_fold"] = {
                    "note": "these launches also carry the encoder's LayerNorm since round 6 (LamEngine.norm_fold): same MFMA work, longer epilogues, no LayerNorm pass",
                    "folded": {"la_gemm_ms": round(g[1] * 1e3, 3), "layernorm_side_ms": ln_ms(agg), "kernel_ms_per_step": round(tot * 1e3, 3)},
                    "layernorm_kernels": {"la_gemm_ms": round(g3[1] * 1e3, 3), "layernorm_side_ms": ln_ms(a3), "kernel_ms_per_step": round(sum(v[1] for v in a3.values()) * 1e3, 3),
                                          "la_gemm_tflops": round(g3[2] / g3[1] / 1e12, 1), "la_gemm_frac": round(g3[2] / g3[1] / 1e12 / PEAK_MFMA_TFLOPS, 4)}}

        top = max(agg.items(), key=lambda kv: kv[1][1])
        if top[0].startswith("twoway_") and (not g or top[1][1] > g[1]):
            # decoder-only workloads (cfg4): the dominant kernel is a fused pass over the image-side stream, bounded by HBM
            i2t = agg.get("twoway_i2t", top[1])
            ach = i2t[3] / i2t[1] / 1e9
            tw = None
            import glob
            for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic*.json")), reverse=True):
                with open(path) as fh:
                    dd = json.load(fh)
                if dd.get("workload") == a.workload and dd.get("twoway"):
                    tw = next((v["bytes_per_launch"] for k, v in dd["twoway"].items() if "i2t" in k), None)
                    break
            roof = {"kernel": "twoway_i2t_kernel (q-proj + image->token attention + out_proj + residual + LayerNorm, in place on the stream)",
                    "bound": "hbm", "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4), "traffic": tw,
                    "algorithmic_bytes_per_launch": round(i2t[3] / i2t[0]), "launches_per_step": i2t[0],
                    "avg_launch_us": round(i2t[1] / i2t[0] * 1e6, 2), "share_of_kernel_time": round(i2t[1] / tot, 3)}

    if rank == 0:
        eps = a.episodes * world * a.steps / elapsed
        line = {
            "metric": "episodes/sec (query+support fwd) ViT-B 1024px 1-shot" if a.workload == "cfg2" else
                      (f"episodes/sec (training steps{', trainable encoder' if a.train_encoder else ''}) {a.workload}" if train else f"episodes/sec (forward) {a.workload}"),
            "value": round(eps, 3), "unit": "episodes/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype,
            "dtype_note": ("BASELINE configs[1] says bf16, north_star's tolerance says 'fp16/bf16': the path is served with fp16 MFMA operands because "
                           "bf16 operands (8 mantissa bits on every activation) measure 4.4e-3 on the cfg2 logits against the 1e-3 tolerance "
                           "(fp16: 7e-4) and run 15 % slower with the full weight-plane set they need (profiles/r03_bench_cfg2_bf16.json); "
                           "--dtype bf16 runs that configuration") if a.dtype == "f16" else
                          "bf16 operands: outside the 1e-3 logit tolerance (4.4e-3 measured on cfg2); the parity configuration is --dtype f16",
            "attn_fp8": bool(a.attn_fp8), "norm_fold": bool(getattr((lam_fwd if train else lam)._engine, "norm_fold", False)) if getattr((lam_fwd if train else lam), "_engine", None) is not None else None,
            "encoder_split_precision": list((lam_fwd if train else lam).precise), "decoder_gemm_dtype": {"f32": "f32", "f16x2": "f16x2 (fp16 plane pairs, 3 products)", "same": a.dtype}[a.decoder], "data": "synthetic",
            "config": {"workload": WORKLOADS[a.workload]["desc"] + ", random-init weights, full-resolution logits",
                       "episodes_per_step_per_gpu": a.episodes, "global_episodes_per_step": a.episodes * world,
                       "images_per_sec": round(eps * (1 + WORKLOADS[a.workload]["episode"]["n_ways"] * WORKLOADS[a.workload]["episode"]["k_shots"]), 2),
                       "parallelism": (f"data-parallel x{world}, one flat-gradient SUM all-reduce (RCCL) per step" if train else
                                       f"episode-sharded x{world}, no collective"),
                       "launch": "eager" if (a.no_graphs or train) else "hipGraph replay"},
            "roofline": roof, "kernels_ms_per_step": kernels,
        }
        if world == 1 and not a.no_eager_baseline and a.workload == "cfg2":
            # BASELINE.md publishes no number; north_star's target is stated against the reference's eager path on the same GPU
            eb = eager_baseline(cfg, dev)
            line["vs_baseline"] = round(eps / eb["value"], 2)
            if eb.get("autocast16", {}).get("value"):
                line["vs_eager_autocast16"] = round(eps / eb["autocast16"]["value"], 2)
            line["baseline_kind"] = eb["kind"]
            line["eager_baseline"] = eb
        if world == 1 and not a.no_cpu_baseline and a.workload == "cfg2":
            line["cpu_baseline"] = cpu_baseline(cfg)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
