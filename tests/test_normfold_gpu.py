"""GPU: LayerNorm folded into its neighbour GEMMs (LaGemmEpilogue.nstat_out / nstat_in, la_norm_finalize, la_norm_stats) against torch.

Reference ops: image_encoder.py:181-197 (`x = x + attn(norm1(x))`, `x = x + mlp(norm2(x))`), models/common.py:19-37 (MLPBlock),
transformers ViTLayer layernorm_before / layernorm_after.  Producer: out32 = a W^T + b + res (+ group vector), out16 its rounding, partial
row sums; consumer: act(LayerNorm(x16) W^T + b) from the un-normalised x16, the gamma-folded weight and the row statistics.
"""
import math

import pytest
import torch
import torch.nn.functional as F

from tests.helpers import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    from labelanything_amd import _lib
    _lib.lib()
    return _lib


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).cuda()


def _producer(L, m, n, k, rpg, res_mod=0, a_kmod=0, with_rvec=True, with_res=True):
    a = rnd(m, a_kmod or k, seed=1).half()
    w = (rnd(n, k, seed=2) / math.sqrt(k)).half()
    bias = rnd(n, seed=3)
    res = rnd(res_mod or m, n, seed=4) if with_res else None
    groups = -(-m // rpg)
    rvec = rnd(groups, n, seed=5, scale=0.3) if with_rvec else None
    af = a.float() if not a_kmod else a.float().repeat(1, -(-k // a_kmod))[:, :k]      # the A columns repeat with period a_kmod
    ref = af @ w.float().t() + bias
    if with_res:
        ref = ref + (res.repeat(m // res_mod, 1) if res_mod else res)
    if with_rvec:
        ref = ref + rvec.repeat_interleave(rpg, dim=0)[:m]
    o32 = torch.empty(m, n, device="cuda") if res_mod or not with_res else res.clone()
    o16 = torch.zeros(m, n, device="cuda", dtype=torch.float16)
    part = torch.full((m, n // 64, 2), float("nan"), device="cuda")
    L.gemm(a, w, bias=bias, res=(res if res_mod or not with_res else o32), res_mod=res_mod, out32=o32, out16=o16, nstat_out=part, rvec=rvec,
           rvec_rpg=rpg if with_rvec else 0, a_kmod=a_kmod)
    torch.cuda.synchronize()
    return ref, o32, o16, part


@pytest.mark.parametrize("shape", [(8192, 768, 768, 4096), (2 * 4096, 768, 3072, 4096), (3 * 901, 768, 768, 901), (5 * 901 + 0, 1024, 1024, 901),
                                   (640, 256, 128, 128)])
def test_producer_gemm_writes_stream_copy_and_row_sums(L, shape):
    m, n, k, rpg = shape
    ref, o32, o16, part = _producer(L, m, n, k, rpg)
    assert rel_err(o32, ref) < 1e-3
    assert torch.equal(o16, o32.half())                      # the 16-bit copy is the rounding of exactly what went to the stream
    assert torch.isfinite(part).all()
    s1, s2 = part[..., 0].sum(1), part[..., 1].sum(1)
    assert rel_err(s1, o32.sum(1)) < 1e-5
    assert rel_err(s2, (o32 * o32).sum(1)) < 1e-5
    # per 64-column slot
    assert rel_err(part[..., 0], o32.view(m, n // 64, 64).sum(2)) < 1e-5


def test_producer_gemm_periodic_residual_and_planes(L):
    """The patch embedding's form: [A_hi | A_lo] against [W_hi | W_hi | W_lo], residual = the position table modulo its rows."""
    m, n, k0, period = 4 * 1024, 768, 256, 1024
    ref, o32, o16, part = _producer(L, m, n, 3 * k0, period, res_mod=period, a_kmod=2 * k0, with_rvec=False)
    assert rel_err(o32, ref) < 1e-3
    assert torch.equal(o16, o32.half())
    assert rel_err(part[..., 0].sum(1), o32.sum(1)) < 1e-5
    # the engine's form: plane pairs only (out16 = hi, aux16 = lo of one [rows, 2 N] buffer), no fp32 matrix at all
    a = rnd(m, 2 * k0, seed=1).half()
    w = (rnd(n, 3 * k0, seed=2) / math.sqrt(3 * k0)).half()
    xs = torch.zeros(m, 2 * n, device="cuda", dtype=torch.float16)
    part2 = torch.zeros_like(part)
    L.gemm(a, w, bias=rnd(n, seed=3), res=rnd(period, n, seed=4), res_mod=period, out16=xs[:, :n], aux16=xs[:, n:], nstat_out=part2, a_kmod=2 * k0)
    torch.cuda.synchronize()
    assert torch.equal(xs[:, :n], o16) and torch.equal(part2, part)
    assert float((xs[:, :n].float() + xs[:, n:].float() - o32).abs().max()) <= 2e-6 * float(o32.abs().max())


def test_producer_gemm_without_group_vector_or_residual(L):
    ref, o32, o16, part = _producer(L, 1024, 768, 768, 4096, with_rvec=False)
    assert rel_err(o32, ref) < 1e-3 and torch.equal(o16, o32.half())
    ref, o32, o16, part = _producer(L, 1024, 768, 768, 4096, with_rvec=False, with_res=False)
    assert rel_err(o32, ref) < 1e-3 and torch.equal(o16, o32.half())


def test_producer_results_do_not_depend_on_the_batch(L):
    """Rows are independent and the partial sums have a fixed order: the first two images of a four-image launch come out bit-identical
    to a two-image launch of the same rows (also with groups that do not end on tile edges: two groups inside one tile)."""
    for rpg, n, k in ((4096, 768, 768), (901, 768, 768)):
        m4, m2 = 4 * rpg, 2 * rpg
        a = rnd(m4, k, seed=41).half()
        w = (rnd(n, k, seed=42) / math.sqrt(k)).half()
        bias, res, rvec = rnd(n, seed=43), rnd(m4, n, seed=44), rnd(4, n, seed=45, scale=0.3)
        outs = []
        for m in (m4, m2):
            o32 = res[:m].clone()
            o16 = torch.zeros(m, n, device="cuda", dtype=torch.float16)
            part = torch.zeros(m, n // 64, 2, device="cuda")
            L.gemm(a[:m], w, bias=bias, res=o32, out32=o32, out16=o16, nstat_out=part, rvec=rvec[: m // rpg], rvec_rpg=rpg)
            torch.cuda.synchronize()
            outs.append((o32, o16, part))
        for big, small in zip(outs[0], outs[1]):
            assert torch.equal(big[:m2], small)


@pytest.mark.parametrize("m,e,eps", [(8192, 768, 1e-6), (2703, 768, 1e-12), (1000, 1024, 1e-6)])
def test_norm_finalize_and_norm_stats(L, m, e, eps):
    x = rnd(m, e, seed=7) * (1.0 + rnd(m, 1, seed=8).abs()) + rnd(m, 1, seed=9)
    mean, var = x.mean(1), x.var(1, unbiased=False)
    part = torch.stack([x.view(m, e // 64, 64).sum(2), (x * x).view(m, e // 64, 64).sum(2)], dim=2).contiguous()
    mpad = -(-m // 256) * 256
    mr = torch.full((mpad, 2), float("nan"), device="cuda")
    L.norm_finalize(part, m, e, eps, mr)
    torch.cuda.synchronize()
    assert rel_err(mr[:m, 0], mean) < 1e-5
    assert rel_err(mr[:m, 1], (var + eps).rsqrt()) < 1e-4
    assert torch.equal(mr[m:], torch.zeros_like(mr[m:]))
    x16 = torch.empty(m, e, device="cuda", dtype=torch.float16)
    mr2 = torch.zeros(mpad, 2, device="cuda")
    L.norm_stats(x, eps, x16, mr2)
    torch.cuda.synchronize()
    assert torch.equal(x16, x.half())
    assert rel_err(mr2[:m, 0], mean) < 1e-6
    assert rel_err(mr2[:m, 1], (var + eps).rsqrt()) < 1e-5


@pytest.mark.parametrize("groups,rpg,e", [(3, 4096, 768), (4, 901, 768), (2, 1025, 1024)])
def test_norm_finalize_column_sums(L, groups, rpg, e):
    m = groups * rpg
    x = rnd(m, e, seed=11) * 1.5 + 0.2
    x16 = x.half()
    part = torch.stack([x.view(m, e // 64, 64).sum(2), (x * x).view(m, e // 64, 64).sum(2)], dim=2).contiguous()
    mr = torch.zeros(-(-m // 256) * 256, 2, device="cuda")
    chunks = L.norm_cs_chunks(rpg)
    cs = torch.full((groups * chunks, e), float("nan"), device="cuda")
    L.norm_finalize(part, m, e, 1e-6, mr, x16=x16, rpg=rpg, cs_part=cs)
    bar = torch.empty(groups, e, device="cuda")
    L.colsum_fold(cs, groups, chunks, e, 1.0 / rpg, bar)
    torch.cuda.synchronize()
    z = (x16.float() - x.mean(1, keepdim=True)) * (x.var(1, unbiased=False, keepdim=True) + 1e-6).rsqrt()
    ref = z.view(groups, rpg, e).mean(1)
    assert float((bar - ref).abs().max()) < 2e-5
    # from ready-made statistics (la_norm_stats): the same sums
    cs2 = torch.zeros_like(cs)
    L.norm_finalize(None, m, e, 1e-6, mr, x16=x16, rpg=rpg, cs_part=cs2)
    torch.cuda.synchronize()
    assert torch.equal(cs, cs2)


@pytest.mark.parametrize("m,n,k,act", [(8192, 2304, 768, 0), (8192, 3072, 768, 1), (2703, 3072, 768, 1), (2703, 2304, 768, 0), (512, 256, 128, 1)])
def test_consumer_gemm_applies_the_layernorm_to_the_product(L, m, n, k, act):
    x = rnd(m, k, seed=21) * (0.5 + rnd(m, 1, seed=22).abs()) + 0.3 * rnd(m, 1, seed=23)
    gamma, beta = 1.0 + 0.1 * rnd(k, seed=24), 0.05 * rnd(k, seed=25)
    w = rnd(n, k, seed=26) / math.sqrt(k)
    b = 0.02 * rnd(n, seed=27)
    x16 = x.half()
    wf = (w * gamma).half()
    ncol = wf.float().sum(1).contiguous()
    bf = (b + w @ beta).contiguous()
    mr = torch.zeros(-(-m // 256) * 256, 2, device="cuda")
    mr[:m, 0] = x.mean(1)
    mr[:m, 1] = (x.var(1, unbiased=False) + 1e-6).rsqrt()
    out = torch.empty(m, n, device="cuda", dtype=torch.float16)
    L.gemm(x16, wf, bias=bf, out16=out, act=L.ACT_GELU if act else L.ACT_NONE, nstat_in=mr, ncol=ncol)
    torch.cuda.synchronize()
    # exactly what the epilogue computes, in fp32
    pre = mr[:m, 1:2] * (x16.float() @ wf.float().t() - mr[:m, 0:1] * ncol) + bf
    ref = F.gelu(pre) if act else pre
    assert rel_err(out, ref) < 2e-3
    # ... which is LayerNorm(x16) W^T + b up to the 16-bit rounding of the folded weight
    ln = F.layer_norm(x16.float(), (k,), gamma, beta, 1e-6) @ w.t() + b
    assert rel_err(pre, ln) < 2e-3


def test_fold_chain_matches_layernorm_then_linear(L):
    """producer -> finalize -> consumer against residual add -> LayerNorm -> Linear in fp32."""
    m, e, n = 2 * 4096, 768, 2304
    a = rnd(m, e, seed=31).half()
    wo = (rnd(e, e, seed=32) / math.sqrt(e)).half()
    bo = 0.02 * rnd(e, seed=33)
    stream = rnd(m, e, seed=34)
    rvec = 0.1 * rnd(2, e, seed=35)
    gamma, beta = 1.0 + 0.1 * rnd(e, seed=36), 0.05 * rnd(e, seed=37)
    w = rnd(n, e, seed=38) / math.sqrt(e)
    b = 0.02 * rnd(n, seed=39)
    ref_stream = stream + a.float() @ wo.float().t() + bo + rvec.repeat_interleave(4096, dim=0)
    ref = F.layer_norm(ref_stream, (e,), gamma, beta, 1e-6) @ w.t() + b
    x16 = torch.empty(m, e, device="cuda", dtype=torch.float16)
    part = torch.empty(m, e // 64, 2, device="cuda")
    L.gemm(a, wo, bias=bo, res=stream, out32=stream, out16=x16, nstat_out=part, rvec=rvec, rvec_rpg=4096)
    mr = torch.zeros(m, 2, device="cuda")
    L.norm_finalize(part, m, e, 1e-6, mr)
    wf = (w * gamma).half()
    out = torch.empty(m, n, device="cuda", dtype=torch.float16)
    L.gemm(x16, wf, bias=(b + w @ beta).contiguous(), out16=out, nstat_in=mr, ncol=wf.float().sum(1).contiguous())
    torch.cuda.synchronize()
    assert rel_err(stream, ref_stream) < 1e-3
    assert rel_err(out, ref) < 3e-3


def test_fold_forms_refuse_what_the_direct_epilogue_cannot_do(L):
    a = rnd(512, 768, seed=1).half()
    w = rnd(200, 768, seed=2).half()
    o16 = torch.empty(512, 200, device="cuda", dtype=torch.float16)
    o32 = torch.empty(512, 200, device="cuda")
    part = torch.empty(512, 4, 2, device="cuda")
    with pytest.raises(RuntimeError):
        L.gemm(a, w, out32=o32, out16=o16, nstat_out=part)          # N % 256 != 0
    w = rnd(256, 768, seed=2).half()
    o16 = torch.empty(512, 256, device="cuda", dtype=torch.float16)
    with pytest.raises(RuntimeError):
        L.gemm(a, w, out16=o16, nstat_out=torch.empty(512, 4, 2, device="cuda"))       # producer needs out32
    with pytest.raises(RuntimeError):
        L.gemm(a.bfloat16(), w.bfloat16(), out16=o16.bfloat16(), nstat_in=torch.zeros(512, 2, device="cuda"), ncol=torch.zeros(256, device="cuda"))


def test_producer_gemm_leaves_plane_pairs_and_saturates(L):
    """aux16 with nstat_out: [hi | lo] rows of what went to the stream (the LA_F16X2 operand of the SAM neck's 1 x 1 convolution); the
    16-bit copy saturates at the fp16 range instead of becoming inf."""
    m, n, k = 1024, 768, 768
    a = rnd(m, k, seed=51).half()
    w = (rnd(n, k, seed=52) / math.sqrt(k)).half()
    res = rnd(m, n, seed=53)
    res[5, 7] = 3.0e5
    res[9, 700] = -7.0e4
    o32 = res.clone()
    xs = torch.zeros(m, 2 * n, device="cuda", dtype=torch.float16)
    part = torch.empty(m, n // 64, 2, device="cuda")
    L.gemm(a, w, res=o32, out32=o32, out16=xs[:, :n], aux16=xs[:, n:], nstat_out=part)
    torch.cuda.synchronize()
    ref = res + a.float() @ w.float().t()
    assert rel_err(o32, ref) < 1e-3
    clamped = o32.clamp(-65504.0, 65504.0)
    hi = clamped.half()
    assert torch.equal(xs[:, :n], hi) and torch.isfinite(xs.float()).all()
    assert torch.equal(xs[:, n:], (clamped - hi.float()).half())
    inr = o32.abs() < 6e4
    assert float(((xs[:, :n].float() + xs[:, n:].float()) - o32)[inr].abs().max()) < 2e-6 * float(o32[inr].abs().max()) + 1e-7


@pytest.mark.parametrize("shape", [(8192, 768, 768, 4096), (2 * 4096, 768, 3072, 4096), (3 * 901, 768, 768, 901), (640, 256, 128, 128)])
def test_producer_gemm_on_a_plane_pair_stream_in_place(L, shape):
    """out32 / res absent: the stream is [hi | lo] fp16 planes (out16 / aux16), read as the residual and written back in place; the
    hi plane is the next GEMM's operand.  Against fp32 arithmetic on hi + lo."""
    m, n, k, rpg = shape
    a = rnd(m, k, seed=71).half()
    w = (rnd(n, k, seed=72) / math.sqrt(k)).half()
    bias = rnd(n, seed=73)
    rvec = rnd(-(-m // rpg), n, seed=74, scale=0.3)
    x0 = rnd(m, n, seed=75) * 3.0
    xs = torch.empty(m, 2 * n, device="cuda", dtype=torch.float16)
    xs[:, :n] = x0.half()
    xs[:, n:] = (x0 - x0.half().float()).half()
    before = xs[:, :n].float() + xs[:, n:].float()
    ref = before + a.float() @ w.float().t() + bias + rvec.repeat_interleave(rpg, dim=0)[:m]
    part = torch.full((m, n // 64, 2), float("nan"), device="cuda")
    L.gemm(a, w, bias=bias, out16=xs[:, :n], aux16=xs[:, n:], nstat_out=part, rvec=rvec, rvec_rpg=rpg)
    torch.cuda.synchronize()
    got = xs[:, :n].float() + xs[:, n:].float()
    assert rel_err(got, ref) < 1e-3                      # (the product's fp16 operands; the pair itself resolves 2^-22)
    assert float((got - ref).abs().max()) < 3e-3 * float(ref.abs().max()) / 8
    hi = xs[:, :n]
    # hi is the 16-bit rounding of the pair's value - the consumers' operand (up to the rare double rounding: lo = rn16(x - hi) can land
    # exactly on half an ulp of hi)
    assert float((hi != (hi.float() + xs[:, n:].float()).half()).float().mean()) < 1e-3
    assert float((hi.float() - got).abs().max()) <= 2.0 ** -11 * float(got.abs().max())
    assert rel_err(part[..., 0].sum(1), got.sum(1)) < 1e-5
    assert rel_err(part[..., 1].sum(1), (got * got).sum(1)) < 1e-5
    # twenty more read-modify-writes of a zero product leave the pair where it is (no drift: hi + lo is re-split exactly)
    z = torch.zeros_like(a)
    snap = got.clone()
    for _ in range(20):
        L.gemm(z, w, out16=xs[:, :n], aux16=xs[:, n:], nstat_out=part)
    torch.cuda.synchronize()
    assert float(((xs[:, :n].float() + xs[:, n:].float()) - snap).abs().max()) <= 1e-6 * float(snap.abs().max())
