"""GPU: every C-ABI kernel against a plain torch fp32 reference of the same op.

Inputs are rounded to the 16-bit operand type first, so the only differences left are the
accumulation order and the 16-bit rounding of outputs: tolerances are 2e-3 (f16) / 1.6e-2 (bf16)
relative to the output's max magnitude for 16-bit outputs and 1e-3 / 2e-3 for fp32 outputs.
"""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import lam_oracle as O
from tests.helpers import rel_err

pytestmark = pytest.mark.gpu

DTYPES = [torch.float16, torch.bfloat16]
TOL16 = {torch.float16: 2e-3, torch.bfloat16: 1.6e-2}
TOL32 = {torch.float16: 1e-3, torch.bfloat16: 4e-3}


@pytest.fixture(scope="module")
def L():
    from labelanything_amd import _lib
    _lib.lib()
    return _lib


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).cuda()


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("mnk", [(300, 200, 768), (128, 128, 64), (1024, 2304, 768), (77, 40, 288), (5, 256, 256)])
def test_gemm_bias_act_residual(L, dt, mnk):
    m, n, k = mnk
    a = rnd(m, k, seed=1).to(dt)
    w = (rnd(n, k, seed=2) / math.sqrt(k)).to(dt)
    bias = rnd(n, seed=3)
    res = rnd(m, n, seed=4)
    ref = F.gelu(a.float() @ w.float().t() + bias) + res
    o32 = torch.empty(m, n, device="cuda")
    o16 = torch.empty(m, n, device="cuda", dtype=dt)
    L.gemm(a, w, bias=bias, res=res, out32=o32, out16=o16, act=L.ACT_GELU)
    torch.cuda.synchronize()
    assert rel_err(o32, ref) < TOL32[dt]
    assert rel_err(o16, ref) < TOL16[dt]


@pytest.mark.parametrize("dt", DTYPES)
def test_gemm_relu_and_res_mod(L, dt):
    m, n, k = 4 * 196, 128, 768
    a = rnd(m, k, seed=5).to(dt)
    w = (rnd(n, k, seed=6) / math.sqrt(k)).to(dt)
    pos = rnd(196, n, seed=7)
    ref = torch.relu(a.float() @ w.float().t()) + pos.repeat(4, 1)
    o32 = torch.empty(m, n, device="cuda")
    L.gemm(a, w, res=pos, res_mod=196, out32=o32, act=L.ACT_RELU)
    torch.cuda.synchronize()
    assert rel_err(o32, ref) < TOL32[dt]


@pytest.mark.parametrize("dt", DTYPES)
def test_gemm_group_map_cls_gap(L, dt):
    """HF ViT patch embedding: rows land after a CLS slot, pos-embed added by (dst_row % (hw+1))."""
    bn, hw, n, k = 3, 25, 128, 768
    a = rnd(bn * hw, k, seed=8).to(dt)
    w = (rnd(n, k, seed=9) / math.sqrt(k)).to(dt)
    pos = rnd(hw + 1, n, seed=10)
    out = torch.zeros(bn * (hw + 1), n, device="cuda")
    L.gemm(a, w, res=pos, res_mod=hw + 1, out32=out, map=L.MAP_GROUP, p=(hw, hw + 1, 1, 0, 0))
    torch.cuda.synchronize()
    ref = torch.zeros(bn, hw + 1, n, device="cuda")
    ref[:, 1:] = (a.float() @ w.float().t()).view(bn, hw, n) + pos[1:]
    assert rel_err(out.view(bn, hw + 1, n)[:, 1:], ref[:, 1:]) < TOL32[dt]
    assert float(out.view(bn, hw + 1, n)[:, 0].abs().max()) == 0.0


@pytest.mark.parametrize("dt", DTYPES)
def test_gemm_window_merge(L, dt):
    """proj GEMM over window-partitioned rows + un-partition + shortcut add (image_encoder.py:189-194)."""
    b, h, w_, ws, e = 2, 14, 14, 8, 128
    xw_tokens = rnd(b * 4 * ws * ws, e, seed=11).to(dt)        # 2x2 windows of 8x8 per image
    wt = (rnd(e, e, seed=12) / math.sqrt(e)).to(dt)
    bias = rnd(e, seed=13)
    shortcut = rnd(b * h * w_, e, seed=14)
    out = torch.empty(b * h * w_, e, device="cuda")
    L.gemm(xw_tokens, wt, bias=bias, res=shortcut, out32=out, map=L.MAP_WINDOW_MERGE, p=(ws, 2, 2, h, w_))
    torch.cuda.synchronize()
    y = (xw_tokens.float() @ wt.float().t() + bias).view(b * 4, ws, ws, e)
    ref = O.window_merge(y.cpu(), ws, (16, 16), (h, w_)).reshape(b * h * w_, e).cuda() + shortcut
    assert rel_err(out, ref) < TOL32[dt]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("cin,cout", [(256, 64), (64, 32), (64, 16)])
def test_gemm_conv_transpose_pixel_shuffle(L, dt, cin, cout):
    b, g = 2, 12
    x = rnd(b, cin, g, g, seed=15).to(dt)
    wt = (rnd(cin, cout, 2, 2, seed=16) / math.sqrt(cin)).to(dt)
    bias = rnd(cout, seed=17)
    ref = F.conv_transpose2d(x.float(), wt.float(), bias, stride=2).permute(0, 2, 3, 1).reshape(-1, cout)
    a = x.permute(0, 2, 3, 1).reshape(b * g * g, cin).contiguous()
    wg = wt.permute(2, 3, 1, 0).reshape(4 * cout, cin).contiguous()       # rows (ky, kx, cout)
    out = torch.empty(b * 4 * g * g, cout, device="cuda")
    L.gemm(a, wg, bias=bias, out32=out, map=L.MAP_CONVT2X2, p=(g, g, cout, 0, 0))
    torch.cuda.synchronize()
    assert rel_err(out, ref) < TOL32[dt]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("T", [196, 901])
@pytest.mark.parametrize("geom", [(2, 2, 128), (5, 12, 768)])           # (5 images x 901 tokens: every alignment of an image start in a 4-row quad)
def test_gemm_v_transposed_epilogue(L, dt, T, geom):
    b, heads, e = geom
    tpad = (T + 63) // 64 * 64
    a = rnd(b * T, e, seed=18).to(dt)
    w = (rnd(3 * e, e, seed=19) / math.sqrt(e)).to(dt)
    bias = rnd(3 * e, seed=20)
    qkv = torch.zeros(b * T, 3 * e, device="cuda", dtype=dt)
    vt = torch.zeros(b * heads, 64, tpad, device="cuda", dtype=dt)
    L.gemm(a, w, bias=bias, out16=qkv, vt=vt, vt_col0=2 * e, vt_T=T, vt_Tpad=tpad, vt_hd=64, vt_heads=heads)
    torch.cuda.synchronize()
    ref = a.float() @ w.float().t() + bias
    assert rel_err(qkv[:, : 2 * e], ref[:, : 2 * e]) < TOL16[dt]
    v = ref[:, 2 * e:].view(b, T, heads, 64).permute(0, 2, 3, 1).reshape(b * heads, 64, T)
    assert rel_err(vt[:, :, :T], v) < TOL16[dt]
    assert float(vt[:, :, T:].float().abs().max()) == 0.0 if tpad > T else True


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("e", [768, 256, 128, 64, 32, 1024, 16, 8])
def test_layernorm_variants(L, dt, e):
    rows = 1000
    x = rnd(rows, e, seed=21, scale=3.0) + 0.5
    x2 = rnd(rows, e, seed=22)
    g_, b_ = rnd(e, seed=23) * 0.1 + 1.0, rnd(e, seed=24) * 0.1
    pe = rnd(250, e, seed=25)
    ref = F.gelu(F.layer_norm(x + x2, (e,), g_, b_, 1e-6))
    o32 = torch.empty(rows, e, device="cuda")
    o16 = torch.empty(rows, e, device="cuda", dtype=dt)
    o16pe = torch.empty(rows, e, device="cuda", dtype=dt)
    L.layernorm(x, g_, b_, 1e-6, x2=x2, gelu=True, out32=o32, out16=o16, out16_pe=o16pe, pe=pe, pe_mod=250, dt=L._DT[dt])
    torch.cuda.synchronize()
    assert rel_err(o32, ref) < 1e-5
    assert rel_err(o16, ref) < TOL16[dt]
    assert rel_err(o16pe, ref + pe.repeat(4, 1)) < TOL16[dt]


@pytest.mark.parametrize("dt", DTYPES)
def test_layernorm_window_partition(L, dt):
    b, h, w_, ws, e = 2, 14, 14, 8, 128
    x = rnd(b * h * w_, e, seed=26)
    g_, b_ = rnd(e, seed=27) * 0.1 + 1.0, rnd(e, seed=28) * 0.1
    out = torch.zeros(b * 4 * ws * ws, e, device="cuda", dtype=dt)
    L.layernorm(x, g_, b_, 1e-6, out16=out, window=ws, H=h, W=w_, dt=L._DT[dt])
    torch.cuda.synchronize()
    y = F.layer_norm(x, (e,), g_, b_, 1e-6).view(b, h, w_, e).cpu()
    ref, _ = O.window_split(y, ws)
    assert rel_err(out.view(-1, ws, ws, e), ref) < TOL16[dt]


@pytest.mark.parametrize("dt", DTYPES)
def test_patch_embed_as_gemm(L, dt):
    bn, s, p, e = 2, 64, 16, 128
    img = rnd(bn, 3, s, s, seed=29)
    wt = (rnd(e, 3, p, p, seed=30) / math.sqrt(3 * p * p)).to(dt)
    bias = rnd(e, seed=31)
    g = s // p
    a = torch.empty(bn * g * g, 3 * p * p, device="cuda", dtype=dt)
    L.im2col_patch(img, p, a)
    out = torch.empty(bn * g * g, e, device="cuda")
    L.gemm(a, wt.reshape(e, -1), bias=bias, out32=out)
    torch.cuda.synchronize()
    ref = F.conv2d(img.to(dt).float(), wt.float(), bias, stride=p).permute(0, 2, 3, 1).reshape(-1, e)
    assert rel_err(out, ref) < TOL32[dt]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("c,co", [(256, 256), (32, 32), (8, 8)])
def test_conv3x3_as_gemm(L, dt, c, co):
    b, h, w_ = 2, 10, 12
    x = rnd(b, c, h, w_, seed=32).to(dt)
    wt = (rnd(co, c, 3, 3, seed=33) / math.sqrt(9 * c)).to(dt)
    bias = rnd(co, seed=34)
    xn = x.permute(0, 2, 3, 1).contiguous()
    a = torch.empty(b * h * w_, 9 * c, device="cuda", dtype=dt)
    L.im2col_3x3(xn, b, h, w_, c, a)
    out = torch.empty(b * h * w_, co, device="cuda")
    L.gemm(a, wt.permute(0, 2, 3, 1).reshape(co, 9 * c).contiguous(), bias=bias, out32=out)
    torch.cuda.synchronize()
    ref = F.conv2d(x.float(), wt.float(), bias, padding=1).permute(0, 2, 3, 1).reshape(-1, co)
    assert rel_err(out, ref) < TOL32[dt]


def _attn_inputs(b, heads, t, dt, seed):
    e = heads * 64
    qkv = (rnd(b * t, 3 * e, seed=seed)).to(dt)
    tpad = (t + 63) // 64 * 64
    v = qkv[:, 2 * e:].view(b, t, heads, 64).permute(0, 2, 3, 1)           # (b, heads, 64, t)
    vt = torch.zeros(b * heads, 64, tpad, device="cuda", dtype=dt)
    vt[:, :, :t] = v.reshape(b * heads, 64, t)
    return qkv, vt, e, tpad


def _attn_ref(qkv, b, heads, t, bias=None):
    e = heads * 64
    q, k, v = [z.view(b, t, heads, 64).transpose(1, 2).float() for z in qkv.float().split(e, dim=1)]
    s = (q * 0.125) @ k.transpose(-1, -2)
    if bias is not None:
        s = s + bias
    return (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(b * t, e)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("t", [901, 196, 64, 1000])
def test_attention_plain(L, dt, t):
    b, heads = 2, 2
    qkv, vt, e, tpad = _attn_inputs(b, heads, t, dt, 40)
    out = torch.empty(b * t, e, device="cuda", dtype=dt)
    L.attn_fwd(qkv, vt, out, None, None, b, heads, t, tpad, 0, e, 0.125, L.ATTN_PLAIN)
    torch.cuda.synchronize()
    assert rel_err(out, _attn_ref(qkv, b, heads, t)) < TOL16[dt]


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("g", [64, 14, 8, 30])
def test_relpos_terms_and_attention(L, dt, g):
    b, heads = (2, 2) if g == 64 else (3, 2)
    t = g * g
    qkv, vt, e, tpad = _attn_inputs(b, heads, t, dt, 41)
    tabh = (rnd(2 * g - 1, 64, seed=42) * 0.3).to(dt)
    tabw = (rnd(2 * g - 1, 64, seed=43) * 0.3).to(dt)
    relh = torch.zeros(b * heads, t, g, device="cuda")
    relw = torch.zeros(b * heads, t, g, device="cuda")
    L.relpos_terms(qkv, b, heads, g, e, tabh, tabw, relh, relw)
    torch.cuda.synchronize()
    q = qkv[:, :e].float().view(b, t, heads, 64).transpose(1, 2).reshape(b * heads, g, g, 64)
    rh = O.rel_pos_table(g, g, tabh.float().cpu()).cuda()
    rw = O.rel_pos_table(g, g, tabw.float().cpu()).cuda()
    ref_h = torch.einsum("nyxc,ykc->nyxk", q, rh).reshape(b * heads, t, g)
    ref_w = torch.einsum("nyxc,xkc->nyxk", q, rw).reshape(b * heads, t, g)
    assert rel_err(relh, ref_h) < 1e-3
    assert rel_err(relw, ref_w) < 1e-3
    bias = (ref_h.view(b, heads, t, g, 1) + ref_w.view(b, heads, t, 1, g)).reshape(b, heads, t, t)
    out = torch.empty(b * t, e, device="cuda", dtype=dt)
    L.attn_fwd(qkv, vt, out, relh, relw, b, heads, t, tpad, g, e, 0.125, L.ATTN_RELPOS)
    torch.cuda.synchronize()
    assert rel_err(out, _attn_ref(qkv, b, heads, t, bias)) < TOL16[dt]
    if g <= 16 or g == 64:   # the same bias computed inside the attention kernel from the tables
        out2 = torch.zeros_like(out)
        L.attn_fwd(qkv, vt, out2, None, None, b, heads, t, tpad, g, e, 0.125, L.ATTN_RELPOS, tabh=tabh, tabw=tabw)
        torch.cuda.synchronize()
        assert rel_err(out2, _attn_ref(qkv, b, heads, t, bias)) < TOL16[dt]


def test_bad_arguments_raise(L):
    a = torch.zeros(4, 12, device="cuda", dtype=torch.float16)
    w = torch.zeros(4, 12, device="cuda", dtype=torch.float16)
    with pytest.raises(RuntimeError, match="multiples of 8"):
        L.gemm(a, w, out32=torch.zeros(4, 4, device="cuda"))


@pytest.mark.parametrize("mnk", [(300, 200, 256), (4096, 128, 256), (1000, 32, 288), (8, 2048, 256), (20, 256, 2048), (3, 32, 256), (64, 8, 72),
                                 (240, 2048, 256), (240, 256, 2048), (300, 40, 136), (512, 64, 1032), (129, 8, 8)])
def test_gemm_fp32_exact_mfma_and_skinny(L, mnk):
    """LA_F32: exact-fp32 MFMA (128 x 128 tiles; 32 x 32 wave tiles for 128 < M <= 512 with and without the in-workgroup K split; the VALU
    skinny kernel for M <= 128) vs torch fp32 matmul."""
    m, n, k = mnk
    a = rnd(m, k, seed=50)
    w = rnd(n, k, seed=51) / math.sqrt(k)
    bias = rnd(n, seed=52)
    res = rnd(m, n, seed=53)
    ref = torch.relu(a.double() @ w.double().t() + bias.double()).float() + res
    o32 = torch.empty(m, n, device="cuda")
    oT = torch.empty(m, n, device="cuda")
    L.gemm(a, w, bias=bias, res=res, out32=o32, out16=oT, act=L.ACT_RELU)
    torch.cuda.synchronize()
    assert rel_err(o32, ref) < 2e-6
    assert torch.equal(o32, oT)


@pytest.mark.parametrize("dt", DTYPES)
def test_gemm_skinny_16bit(L, dt):
    m, n, k = 7, 384, 256
    a = rnd(m, k, seed=54).to(dt)
    w = (rnd(n, k, seed=55) / math.sqrt(k)).to(dt)
    bias = rnd(n, seed=56)
    o32 = torch.empty(m, n, device="cuda")
    L.gemm(a, w, bias=bias, out32=o32, act=L.ACT_GELU)
    torch.cuda.synchronize()
    assert rel_err(o32, F.gelu(a.float() @ w.float().t() + bias)) < 1e-5


@pytest.mark.parametrize("c,co", [(32, 32), (64, 32), (32, 48)])
def test_conv3x3_fp32_implicit_gemm(L, c, co):
    b, h, w_ = 2, 20, 24
    x = rnd(b, c, h, w_, seed=57)
    wt = rnd(co, c, 3, 3, seed=58) / math.sqrt(9 * c)
    bias = rnd(co, seed=59)
    xn = x.permute(0, 2, 3, 1).contiguous()
    out = torch.empty(b * h * w_, co, device="cuda")
    L.conv3x3_f32(xn, b, h, w_, c, wt.permute(0, 2, 3, 1).reshape(co, 9 * c).contiguous(), bias, co, out)
    torch.cuda.synchronize()
    ref = F.conv2d(x.double(), wt.double(), bias.double(), padding=1).float().permute(0, 2, 3, 1).reshape(-1, co)
    assert rel_err(out, ref) < 2e-6


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("g", [14, 8])
def test_window_attention_in_padded_slot_order(L, dt, g):
    """LA_ATTN_RELPOS_WIN16: V^T written by the GEMM epilogue in 16-wide slot order (vt_ws), K gathered by the kernel."""
    b, heads, e = 5, 2, 128
    t = g * g
    tpad = (16 * g + 63) // 64 * 64
    x = rnd(b * t, e, seed=60).to(dt)
    wqkv = (rnd(3 * e, e, seed=61) / math.sqrt(e)).to(dt)
    bias = rnd(3 * e, seed=62) * 0.1
    qkv = torch.zeros(b * t, 3 * e, device="cuda", dtype=dt)
    vt = torch.zeros(b * heads, 64, tpad, device="cuda", dtype=dt)
    L.gemm(x, wqkv, bias=bias, out16=qkv, vt=vt, vt_col0=2 * e, vt_T=t, vt_Tpad=tpad, vt_hd=64, vt_heads=heads, vt_ws=g)
    torch.cuda.synchronize()
    ref_qkv = x.float() @ wqkv.float().t() + bias
    v = ref_qkv[:, 2 * e:].view(b, g, g, heads, 64).permute(0, 3, 4, 1, 2)             # (b, heads, 64, kh, kw)
    slots = vt.view(b, heads, 64, tpad)[..., : 16 * g].reshape(b, heads, 64, g, 16)
    assert rel_err(slots[..., :g], v) < TOL16[dt]
    assert float(slots[..., g:].float().abs().max()) == 0.0
    tabh = (rnd(2 * g - 1, 64, seed=63) * 0.3).to(dt)
    tabw = (rnd(2 * g - 1, 64, seed=64) * 0.3).to(dt)
    qkv[:, 2 * e:] = ref_qkv[:, 2 * e:].to(dt)       # reference path reads V from the qkv buffer
    q = qkv[:, :e].float().view(b, t, heads, 64).transpose(1, 2).reshape(b * heads, g, g, 64)
    rh = O.rel_pos_table(g, g, tabh.float().cpu()).cuda()
    rw = O.rel_pos_table(g, g, tabw.float().cpu()).cuda()
    bh = torch.einsum("nyxc,ykc->nyxk", q, rh).reshape(b, heads, t, g, 1)
    bw = torch.einsum("nyxc,xkc->nyxk", q, rw).reshape(b, heads, t, 1, g)
    ref = _attn_ref(qkv, b, heads, t, (bh + bw).reshape(b, heads, t, t))
    out = torch.zeros(b * t, e, device="cuda", dtype=dt)
    L.attn_fwd(qkv, vt, out, None, None, b, heads, t, tpad, g, e, 0.125, L.ATTN_RELPOS_WIN16, tabh=tabh, tabw=tabw)
    torch.cuda.synchronize()
    assert rel_err(out, ref) < TOL16[dt]


def _wide_inputs(b, heads, t, hd_true, dt, seed):
    """q/k/v with hd_true real dims per head, zero-padded to 128 (what the host does for SAM ViT-H's 80-wide heads)."""
    e = heads * 128
    qkv = torch.zeros(b * t, 3, heads, 128, device="cuda")
    qkv[..., :hd_true] = rnd(b * t, 3, heads, hd_true, seed=seed)
    qkv = qkv.reshape(b * t, 3 * e).to(dt)
    tpad = (t + 63) // 64 * 64
    v = qkv[:, 2 * e:].view(b, t, heads, 128).permute(0, 2, 3, 1)
    vt = torch.zeros(b * heads, 128, tpad, device="cuda", dtype=dt)
    vt[:, :, :t] = v.reshape(b * heads, 128, t)
    return qkv, vt, e, tpad


def _wide_ref(qkv, b, heads, t, scale, bias=None):
    e = heads * 128
    q, k, v = [z.view(b, t, heads, 128).transpose(1, 2).float() for z in qkv.float().split(e, dim=1)]
    s = (q * scale) @ k.transpose(-1, -2)
    if bias is not None:
        s = s + bias
    return (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(b * t, e)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("hd_true", [128, 80])
def test_attention_128_wide_heads_all_modes(L, dt, hd_true):
    """head_dim 128 (two 64-wide halves in every tile), also as the zero-padded image of SAM ViT-H's 80-wide heads
    (image_encoder.py:200-255 with hd = 1280 / 16): plain, rel-pos from la_relpos_terms, in-kernel G = 64, windows."""
    scale = hd_true ** -0.5
    # plain
    b, heads, t = 2, 2, 333
    qkv, vt, e, tpad = _wide_inputs(b, heads, t, hd_true, dt, 70)
    out = torch.empty(b * t, e, device="cuda", dtype=dt)
    L.attn_fwd(qkv, vt, out, None, None, b, heads, t, tpad, 0, e, scale, L.ATTN_PLAIN)
    torch.cuda.synchronize()
    assert rel_err(out, _wide_ref(qkv, b, heads, t, scale)) < TOL16[dt]
    assert hd_true == 128 or float(out.view(b * t, heads, 128)[..., hd_true:].float().abs().max()) == 0.0
    # rel-pos: generic grid via la_relpos_terms (G = 20), in-kernel for G = 64, windows G = 14 (both layouts)
    for g, b, heads in ((20, 2, 2), (64, 1, 2), (14, 3, 2)):
        t = g * g
        qkv, vt, e, tpad = _wide_inputs(b, heads, t, hd_true, dt, 71 + g)
        tabh = torch.zeros(2 * g - 1, 128, device="cuda")
        tabw = torch.zeros(2 * g - 1, 128, device="cuda")
        tabh[:, :hd_true] = rnd(2 * g - 1, hd_true, seed=72) * 0.3
        tabw[:, :hd_true] = rnd(2 * g - 1, hd_true, seed=73) * 0.3
        tabh, tabw = tabh.to(dt), tabw.to(dt)
        q = qkv[:, :e].float().view(b, t, heads, 128).transpose(1, 2).reshape(b * heads, g, g, 128)
        rh = O.rel_pos_table(g, g, tabh.float().cpu()).cuda()
        rw = O.rel_pos_table(g, g, tabw.float().cpu()).cuda()
        ref_h = torch.einsum("nyxc,ykc->nyxk", q, rh).reshape(b * heads, t, g)
        ref_w = torch.einsum("nyxc,xkc->nyxk", q, rw).reshape(b * heads, t, g)
        bias = (ref_h.view(b, heads, t, g, 1) + ref_w.view(b, heads, t, 1, g)).reshape(b, heads, t, t)
        ref = _wide_ref(qkv, b, heads, t, scale, bias)
        relh = torch.zeros(b * heads, t, g, device="cuda")
        relw = torch.zeros(b * heads, t, g, device="cuda")
        L.relpos_terms(qkv, b, heads, g, e, tabh, tabw, relh, relw)
        torch.cuda.synchronize()
        assert rel_err(relh, ref_h) < 1e-3 and rel_err(relw, ref_w) < 1e-3
        out = torch.zeros(b * t, e, device="cuda", dtype=dt)
        L.attn_fwd(qkv, vt, out, relh, relw, b, heads, t, tpad, g, e, scale, L.ATTN_RELPOS)
        torch.cuda.synchronize()
        assert rel_err(out, ref) < TOL16[dt], f"terms from global, G={g}"
        if g <= 16 or g == 64:
            out2 = torch.zeros_like(out)
            L.attn_fwd(qkv, vt, out2, None, None, b, heads, t, tpad, g, e, scale, L.ATTN_RELPOS, tabh=tabh, tabw=tabw)
            torch.cuda.synchronize()
            assert rel_err(out2, ref) < TOL16[dt], f"terms in-kernel, G={g}"
        if g <= 16:          # 16-wide slot order: V^T re-laid out like the GEMM epilogue does (vt_ws)
            tp16 = (16 * g + 63) // 64 * 64
            vt16 = torch.zeros(b * heads, 128, tp16, device="cuda", dtype=dt)
            vt16[..., : 16 * g].view(b * heads, 128, g, 16)[..., :g] = vt[:, :, :t].reshape(b * heads, 128, g, g)
            out3 = torch.zeros_like(out)
            L.attn_fwd(qkv, vt16, out3, None, None, b, heads, t, tp16, g, e, scale, L.ATTN_RELPOS_WIN16, tabh=tabh, tabw=tabw)
            torch.cuda.synchronize()
            assert rel_err(out3, ref) < TOL16[dt], f"WIN16, G={g}"


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", [(2, 20, 20, 8, 256), (1, 64, 64, 14, 768), (3, 9, 13, 4, 128)])
def test_gemm_window_partition_output_and_gather(L, dt, shape):
    """LA_MAP_WINDOW_PART as output map (qkv of a window block: image-order tokens in, window-ordered q/k rows and V^T in
    16-slot order out, padded tokens untouched) and as source map (proj: window-ordered rows gathered back to image order)."""
    b, h, w, ws, e = shape
    heads = e // 64
    nwy, nwx = -(-h // ws), -(-w // ws)
    nb, t = b * nwy * nwx, ws * ws
    rows, arows = b * h * w, nb * t
    x = rnd(rows, e, seed=80).to(dt)
    wqkv = (rnd(3 * e, e, seed=81) / math.sqrt(e)).to(dt)
    bias = rnd(3 * e, seed=82) * 0.1
    tpad = (16 * ws + 63) // 64 * 64
    fill = 7.0
    qkv = torch.full((arows, 3 * e), fill, device="cuda", dtype=dt)
    vt = torch.full((nb * heads, 64, tpad), fill, device="cuda", dtype=dt)
    L.gemm(x, wqkv, bias=bias, out16=qkv, vt=vt, vt_col0=2 * e, vt_T=t, vt_Tpad=tpad, vt_hd=64, vt_heads=heads, vt_ws=ws,
           map=L.MAP_WINDOW_PART, p=(ws, nwy, nwx, h, w))
    torch.cuda.synchronize()
    ref = (x.float() @ wqkv.float().t() + bias).view(b, h, w, 3 * e)
    # reference window partition with padding (image_encoder.py:258-279); padded positions keep the fill value
    pad = torch.full((b, nwy * ws, nwx * ws, 3 * e), fill, device="cuda")
    pad[:, :h, :w] = ref
    win = pad.view(b, nwy, ws, nwx, ws, 3 * e).permute(0, 1, 3, 2, 4, 5).reshape(arows, 3 * e)
    assert rel_err(qkv[:, : 2 * e], win[:, : 2 * e]) < TOL16[dt]
    v = win[:, 2 * e:].view(nb, ws, ws, heads, 64).permute(0, 3, 4, 1, 2)                      # (nb, heads, 64, kh, kw)
    slots = vt.view(nb, heads, 64, tpad)[..., : 16 * ws].reshape(nb, heads, 64, ws, 16)
    assert rel_err(slots[..., :ws], v) < TOL16[dt]
    assert bool((slots[..., ws:] == fill).all()) and bool((vt.view(nb, heads, 64, tpad)[..., 16 * ws:] == fill).all())
    assert bool((qkv[:, 2 * e:] == fill).all())                                                # V columns only go to vt
    # gather: rows of a window-ordered activation back in image order, with bias + residual like the proj GEMM
    ao = rnd(arows, e, seed=83).to(dt)
    wp = (rnd(e, e, seed=84) / math.sqrt(e)).to(dt)
    res = rnd(rows, e, seed=85)
    out = torch.zeros(rows, e, device="cuda")
    L.gemm(ao, wp, bias=bias[:e].contiguous(), res=res, out32=out, M=rows, amap=L.MAP_WINDOW_PART, p=(ws, nwy, nwx, h, w))
    torch.cuda.synchronize()
    merged = ao.float().view(b, nwy, nwx, ws, ws, e).permute(0, 1, 3, 2, 4, 5).reshape(b, nwy * ws, nwx * ws, e)[:, :h, :w].reshape(rows, e)
    assert rel_err(out, merged @ wp.float().t() + bias[:e] + res) < TOL16[dt]


def test_gemm_shape_fuzz_all_kernel_paths(L):
    """Seeded sweep over ragged shapes (M / N tails, every K granularity, fp32 and 16-bit, strided A) through every la_gemm
    kernel (LA_GEMM_PATH is not forced: the dispatcher picks skinny / fallback / 128x128 / 256x128 / ping-pong by shape)."""
    g = torch.Generator().manual_seed(1234)
    shapes = []
    for _ in range(36):
        m = int(torch.randint(1, 700, (1,), generator=g))
        n = int(torch.randint(1, 80, (1,), generator=g)) * 8
        k = int(torch.randint(1, 24, (1,), generator=g)) * [8, 32, 64][int(torch.randint(0, 3, (1,), generator=g))]
        shapes.append((m, n, k))
    shapes += [(33, 8, 8), (257, 136, 72), (512, 512, 2048), (1025, 264, 64), (131072, 256, 64), (65536 + 17, 384, 128),
               (4096, 1024, 2048), (131072, 1024, 2048)]       # the last: 512 ping-pong tiles (K >= 2048) -> gemm_pp_kernel
    for idx, (m, n, k) in enumerate(shapes):
        for dt in (torch.float16, torch.float32):
            if dt == torch.float32 and (m * n * k > 3e9):
                continue
            lda = k + (8 if idx % 3 == 0 else 0)                   # padded row stride every third case
            a_full = rnd(m, lda, seed=100 + idx).to(dt)
            a = a_full[:, :k]
            w = (rnd(n, k, seed=200 + idx) / math.sqrt(k)).to(dt)
            bias = rnd(n, seed=300 + idx)
            use_res = idx % 2 == 0
            res = rnd(m, n, seed=400 + idx) if use_res else None
            act = [L.ACT_NONE, L.ACT_RELU, L.ACT_GELU][idx % 3]
            ref = a.float() @ w.float().t() + bias
            ref = torch.relu(ref) if act == L.ACT_RELU else (F.gelu(ref) if act == L.ACT_GELU else ref)
            if use_res:
                ref = ref + res
            o32 = torch.full((m, n), float("nan"), device="cuda")
            L.gemm(a_full, w, bias=bias, res=res, out32=o32, act=act, lda=lda) if lda != k else L.gemm(a, w, bias=bias, res=res, out32=o32, act=act)
            torch.cuda.synchronize()
            tol = 2e-5 if dt == torch.float32 else 1e-3
            assert torch.isfinite(o32).all(), (m, n, k, dt)
            assert rel_err(o32, ref) < tol, (m, n, k, dt, float(rel_err(o32, ref)))


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("mnk", [(300, 200, 768),          # 128x128 LDS-DMA kernel
                                 (256 * 80, 1024, 768),     # >= 512 tiles of 256x128: 256x128 kernel (K = 1536 with two planes)
                                 (256 * 48, 3072, 1024),    # >= 512 tiles of 256x256, K = 2048: ping-pong kernel
                                 (40, 72, 64)])             # unaligned N: register-staged kernel
def test_gemm_split_precision_weights(L, dt, mnk):
    """a_kmod: W = [W_hi | W_lo] against ONE 16-bit A -> the weights enter with ~2x the mantissa bits.
    The result must match the fp32-weight product to accumulation accuracy, i.e. far below the 16-bit weight rounding."""
    m, n, k = mnk
    a = rnd(m, k, seed=21).to(dt)
    w32 = rnd(n, k, seed=22) / math.sqrt(k)
    hi = w32.to(dt)
    lo = (w32 - hi.float()).to(dt)
    w2 = torch.cat([hi, lo], dim=1).contiguous()
    bias = rnd(n, seed=23)
    ref = a.float() @ w32.t() + bias
    o32 = torch.empty(m, n, device="cuda")
    L.gemm(a, w2, bias=bias, out32=o32, a_kmod=k)
    o1 = torch.empty(m, n, device="cuda")
    L.gemm(a, hi, bias=bias, out32=o1)
    torch.cuda.synchronize()
    e2, e1 = rel_err(o32, ref), rel_err(o1, ref)
    bound = 2e-6 if dt == torch.float16 else 2e-5        # fp16 planes: 22 bits; bf16 planes: 16 bits
    assert e2 < bound, (e2, e1)
    assert e1 > 4 * e2                                    # and the single plane really is the coarser one
    with pytest.raises(RuntimeError, match="a_kmod"):
        L.gemm(a, w2, out32=o32, a_kmod=k + 8)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("planes", [1, 2])
def test_gemm_persistent_256x256_epilogues(L, dt, planes):
    """gemm_t256p_kernel (persistent workgroups, k-step stream running across tile seams, wave-private epilogue slabs) on its three
    epilogues: bias -> GELU -> 16 bit; bias + in-place fp32 residual (+ 16-bit copy) behind a window-gather A map; bias -> 16 bit with
    the V columns leaving transposed in 16-slot window order.  M is not a multiple of 256 (row-edge tiles) and there are more tiles
    than CUs (every workgroup crosses seams)."""
    k = 768

    def weights(n, seed):
        w32 = rnd(n, k, seed=seed) / math.sqrt(k)
        hi = w32.to(dt)
        if planes == 1:
            return hi.contiguous(), hi.float(), {}
        lo = (w32 - hi.float()).to(dt)
        return torch.cat([hi, lo], dim=1).contiguous(), hi.float() + lo.float(), {"a_kmod": k}

    tol = TOL16[dt]
    # --- GELU (lin1 shape class) ---------------------------------------------------------------------------------------------
    m, n = 256 * (43 if planes == 1 else 11) + 100, 3072
    a = rnd(m, k, seed=61).to(dt)
    w, wref, kw = weights(n, 62)
    bias = rnd(n, seed=63)
    out = torch.zeros(m, n, device="cuda", dtype=dt)
    L.gemm(a, w, bias=bias, out16=out, act=L.ACT_GELU, **kw)
    torch.cuda.synchronize()
    assert rel_err(out, F.gelu(a.float() @ wref.t() + bias)) < tol
    # --- residual in place, A rows gathered in window order (proj of a window block) ------------------------------------------------
    b, h, ws = 3, 64, 14
    nwy = -(-h // ws)
    rows, arows = b * h * h, b * nwy * nwy * ws * ws
    n = 768 if planes == 2 else 3072                 # >= 128 (two planes) / >= 512 (one plane) tiles of 256 x 256
    ao = rnd(arows, k, seed=64).to(dt)
    w, wref, kw = weights(n, 65)
    bias = rnd(n, seed=66)
    res = rnd(rows, n, seed=67)
    ref_res = res.clone()
    o16 = torch.zeros(rows, n, device="cuda", dtype=dt)
    L.gemm(ao, w, bias=bias, res=res, out32=res, out16=o16, M=rows, amap=L.MAP_WINDOW_PART, p=(ws, nwy, nwy, h, h), **kw)
    torch.cuda.synchronize()
    merged = ao.float().view(b, nwy, nwy, ws, ws, k).permute(0, 1, 3, 2, 4, 5).reshape(b, nwy * ws, nwy * ws, k)[:, :h, :h].reshape(rows, k)
    ref = merged @ wref.t() + bias + ref_res
    assert rel_err(res, ref) < 1e-5
    assert rel_err(o16, ref) < tol
    # --- qkv of window blocks: V^T in 16-slot order -----------------------------------------------------------------------------------
    e, heads, t = 768, 12, ws * ws
    nb = 75 if planes == 1 else 40
    tpad = (16 * ws + 63) // 64 * 64
    x = rnd(nb * t, k, seed=68).to(dt)
    w, wref, kw = weights(3 * e, 69)
    bias = rnd(3 * e, seed=70) * 0.1
    fill = 7.0
    qkv = torch.full((nb * t, 3 * e), fill, device="cuda", dtype=dt)
    vt = torch.full((nb * heads, 64, tpad), fill, device="cuda", dtype=dt)
    L.gemm(x, w, bias=bias, out16=qkv, vt=vt, vt_col0=2 * e, vt_T=t, vt_Tpad=tpad, vt_hd=64, vt_heads=heads, vt_ws=ws, **kw)
    torch.cuda.synchronize()
    ref = x.float() @ wref.t() + bias
    assert rel_err(qkv[:, : 2 * e], ref[:, : 2 * e]) < tol
    assert bool((qkv[:, 2 * e:] == fill).all())
    v = ref[:, 2 * e:].view(nb, ws, ws, heads, 64).permute(0, 3, 4, 1, 2)
    slots = vt.view(nb, heads, 64, tpad)[..., : 16 * ws].reshape(nb, heads, 64, ws, 16)
    assert rel_err(slots[..., :ws], v) < tol
    assert bool((slots[..., ws:] == fill).all()) and bool((vt.view(nb, heads, 64, tpad)[..., 16 * ws:] == fill).all())
    # --- and the global-attention form (identity slots) -------------------------------------------------------------------------------
    t, nb = 4096, 4 if planes == 1 else 2
    x = rnd(nb * t, k, seed=71).to(dt)
    qkv = torch.zeros(nb * t, 3 * e, device="cuda", dtype=dt)
    vt = torch.zeros(nb * heads, 64, t, device="cuda", dtype=dt)
    L.gemm(x, w, bias=bias, out16=qkv, vt=vt, vt_col0=2 * e, vt_T=t, vt_Tpad=t, vt_hd=64, vt_heads=heads, **kw)
    torch.cuda.synchronize()
    ref = x.float() @ wref.t() + bias
    assert rel_err(qkv[:, : 2 * e], ref[:, : 2 * e]) < tol
    assert rel_err(vt, ref[:, 2 * e:].view(nb, t, heads, 64).permute(0, 2, 3, 1).reshape(nb * heads, 64, t)) < tol


@pytest.mark.parametrize("dt", DTYPES)
def test_gemm_four_wave_kernel_bit_identical(L, dt):
    """gemm_t256w (four waves x 512 registers, la_gemm_variant 2: the default) against gemm_t256q (variant 1): the same
    MFMA order over k and the same epilogue arithmetic, so every output must be equal bit for bit - on interior tiles (its own
    epilogue through the fp32 slab), on row-edge tiles, behind a window-gather A map, with V^T columns and with the scatter map."""
    k = 768

    def run_all(fn, outs):
        res = {}
        for v in (1, 2):
            L.gemm_variant(v)
            for o in outs:
                o.fill_(3.0)
            fn()
            torch.cuda.synchronize()
            res[v] = [o.clone() for o in outs]
        L.gemm_variant(2)
        for a_, b_ in zip(res[1], res[2]):
            assert torch.equal(a_, b_)
        return res[1]

    try:
        for m in (256 * 44, 256 * 43 + 100):
            # GELU -> 16 bit (lin1)
            n = 3072
            a = rnd(m, k, seed=161).to(dt)
            w = (rnd(n, k, seed=162) / math.sqrt(k)).to(dt)
            bias = rnd(n, seed=163)
            out = torch.zeros(m, n, device="cuda", dtype=dt)
            got = run_all(lambda: L.gemm(a, w, bias=bias, out16=out, act=L.ACT_GELU), [out])[0]
            assert rel_err(got, F.gelu(a.float() @ w.float().t() + bias)) < TOL16[dt]
            # plain -> 16 bit (q | k)
            n = 1536
            w = (rnd(n, k, seed=164) / math.sqrt(k)).to(dt)
            bias = rnd(n, seed=165)
            out = torch.zeros(m, n, device="cuda", dtype=dt)
            got = run_all(lambda: L.gemm(a, w, bias=bias, out16=out), [out])[0]
            assert rel_err(got, a.float() @ w.float().t() + bias) < TOL16[dt]
        # fp32 residual in place (+ 16-bit copy), long K (lin2), then behind a window-gather A map (proj of a window block)
        m, n, kk = 256 * 40, 3072, 3072
        a = rnd(m, kk, seed=166).to(dt)
        w = (rnd(n, kk, seed=167) / math.sqrt(kk)).to(dt)
        bias = rnd(n, seed=168)
        res0 = rnd(m, n, seed=169)
        res = res0.clone()
        o16 = torch.zeros(m, n, device="cuda", dtype=dt)

        def lin2():
            res.copy_(res0)
            L.gemm(a, w, bias=bias, res=res, out32=res, out16=o16)
        got = run_all(lin2, [res, o16])
        assert rel_err(got[0], a.float() @ w.float().t() + bias + res0) < 1e-5
        b, h, ws = 4, 64, 14
        nwy = -(-h // ws)
        rows, arows = b * h * h, b * nwy * nwy * ws * ws
        n = 3072
        ao = rnd(arows, k, seed=170).to(dt)
        w = (rnd(n, k, seed=171) / math.sqrt(k)).to(dt)
        bias = rnd(n, seed=172)
        res0 = rnd(rows, n, seed=173)
        res = res0.clone()

        def proj():
            res.copy_(res0)
            L.gemm(ao, w, bias=bias, res=res, out32=res, M=rows, amap=L.MAP_WINDOW_PART, p=(ws, nwy, nwy, h, h))
        run_all(proj, [res])
        # q | k | v with V^T columns (global attention form)
        e, heads, t, nb = 768, 12, 4096, 4
        x = rnd(nb * t, k, seed=174).to(dt)
        w = (rnd(3 * e, k, seed=175) / math.sqrt(k)).to(dt)
        bias = rnd(3 * e, seed=176) * 0.1
        qkv = torch.zeros(nb * t, 3 * e, device="cuda", dtype=dt)
        vt = torch.zeros(nb * heads, 64, t, device="cuda", dtype=dt)
        run_all(lambda: L.gemm(x, w, bias=bias, out16=qkv, vt=vt, vt_col0=2 * e, vt_T=t, vt_Tpad=t, vt_hd=64, vt_heads=heads), [qkv, vt])
    finally:
        L.gemm_variant(2)


@pytest.mark.parametrize("dt", DTYPES)
def test_gemm_persistent_window_scatter(L, dt):
    """qkv of a SAM window block from IMAGE-order tokens (LA_MAP_WINDOW_PART as output map in gemm_t256p_kernel): the q / k rows land in
    window order, V^T in 16-slot order, the padded positions are never written (they hold the bias, filled once by the host) - as one
    one-plane launch over q | k | v, and as the one-plane q / k launch + two-plane V launch of the default numerics."""
    b, h, ws, e, heads = 4, 64, 14, 768, 12
    nwy = -(-h // ws)
    nb, t = b * nwy * nwy, ws * ws
    rows, arows = b * h * h, nb * t
    tpad = (16 * ws + 63) // 64 * 64
    x = rnd(rows, e, seed=90).to(dt)
    w32 = rnd(3 * e, e, seed=91) / math.sqrt(e)
    bias = rnd(3 * e, seed=92) * 0.1
    fill = 7.0
    hi = w32.to(dt)
    lo = (w32 - hi.float()).to(dt)

    def check(qkv, vt, wref):
        ref = (x.float() @ wref.t() + bias).view(b, h, h, 3 * e)
        pad = torch.full((b, nwy * ws, nwy * ws, 3 * e), fill, device="cuda")
        pad[:, :h, :h] = ref
        win = pad.view(b, nwy, ws, nwy, ws, 3 * e).permute(0, 1, 3, 2, 4, 5).reshape(arows, 3 * e)
        assert rel_err(qkv[:, : 2 * e], win[:, : 2 * e]) < TOL16[dt]
        assert bool((qkv[:, 2 * e:] == fill).all())
        v = win[:, 2 * e:].view(nb, ws, ws, heads, 64).permute(0, 3, 4, 1, 2)
        slots = vt.view(nb, heads, 64, tpad)[..., : 16 * ws].reshape(nb, heads, 64, ws, 16)
        assert rel_err(slots[..., :ws], v) < TOL16[dt]
        assert bool((slots[..., ws:] == fill).all()) and bool((vt.view(nb, heads, 64, tpad)[..., 16 * ws:] == fill).all())

    kw = dict(vt_T=t, vt_Tpad=tpad, vt_hd=64, vt_heads=heads, vt_ws=ws, map=L.MAP_WINDOW_PART, p=(ws, nwy, nwy, h, h))
    # one plane, q | k | v in one launch
    qkv = torch.full((arows, 3 * e), fill, device="cuda", dtype=dt)
    vt = torch.full((nb * heads, 64, tpad), fill, device="cuda", dtype=dt)
    L.gemm(x, hi.contiguous(), bias=bias, out16=qkv, vt=vt, vt_col0=2 * e, **kw)
    torch.cuda.synchronize()
    check(qkv, vt, hi.float())
    # q / k one plane + V two planes
    qkv = torch.full((arows, 3 * e), fill, device="cuda", dtype=dt)
    vt = torch.full((nb * heads, 64, tpad), fill, device="cuda", dtype=dt)
    L.gemm(x, hi[: 2 * e].contiguous(), bias=bias[: 2 * e], out16=qkv[:, : 2 * e], map=L.MAP_WINDOW_PART, p=(ws, nwy, nwy, h, h))
    wv = torch.cat([hi[2 * e:], lo[2 * e:]], dim=1).contiguous()
    L.gemm(x, wv, bias=bias[2 * e:], out16=qkv[:, 2 * e:], vt=vt, vt_col0=0, a_kmod=e, **kw)
    torch.cuda.synchronize()
    wref = hi.float().clone()
    wref[2 * e:] += lo[2 * e:].float()
    check(qkv, vt, wref)


def _planes(w):
    hi = w.to(torch.float16)
    return hi.contiguous(), (w - hi.float()).to(torch.float16).contiguous()


@pytest.mark.parametrize("d", [256, 512])
@pytest.mark.parametrize("g,hw,nt", [(3, 900, 5), (2, 4096, 1), (1, 128, 11), (5, 200, 32), (2, 50, 3)])
def test_fused_twoway_image_side_kernels(L, g, hw, nt, d):
    """la_twoway_t2i / la_twoway_i2t (D = 256: 8 heads of 16, two 64-row tiles in flight per CU; D = 512: 8 heads of 32, the published
    SAM-1024 decoder geometry) against the same mathematics in torch fp32: k / v / q projections from the (groups, hw, D) stream, softmax
    attention against / over a handful of tokens, out_proj + residual + LayerNorm."""
    di, heads, hd = d // 2, 8, d // 16
    x = rnd(g * hw, d, seed=31)
    pe = rnd(hw, d, seed=32)
    wk, wv, wq = (rnd(di, d, seed=33 + i) / 16 for i in range(3))
    wo = rnd(d, di, seed=36) / 11
    bk, bv, bq, bo = rnd(di, seed=37), rnd(di, seed=38), rnd(di, seed=39), rnd(d, seed=40)
    gamma, beta = 1 + 0.1 * rnd(d, seed=41), 0.1 * rnd(d, seed=42)
    qt, kt, vt = rnd(g * nt, di, seed=43), rnd(g * nt, di, seed=44), rnd(g * nt, di, seed=45)
    xp = (x.view(g, hw, d) + pe).reshape(g * hw, d)

    def heads_of(t, n):
        return t.view(g, n, heads, hd).transpose(1, 2)
    # tokens -> image
    kk, vv = xp @ wk.t() + bk, x @ wv.t() + bv
    att = torch.softmax(heads_of(qt, nt) @ heads_of(kk, hw).transpose(-1, -2) / math.sqrt(hd), dim=-1) @ heads_of(vv, hw)
    ref_t2i = att.transpose(1, 2).reshape(g * nt, di)
    part = torch.full((L.twoway_part_size(g, hw, nt, d),), float("nan"), device="cuda")
    out = torch.empty(g * nt, di, device="cuda")
    L.twoway_t2i(x, _planes(wk), _planes(wv), L.twoway_pe_layout((pe @ wk.t() + bk).contiguous()), bv, qt, g, hw, nt, heads, part, out)
    # image -> tokens, in place
    qq = xp @ wq.t() + bq
    o = torch.softmax(heads_of(qq, hw) @ heads_of(kt, nt).transpose(-1, -2) / math.sqrt(hd), dim=-1) @ heads_of(vt, nt)
    y = o.transpose(1, 2).reshape(g * hw, di) @ wo.t() + bo + x
    ref_i2t = F.layer_norm(y, (d,), gamma, beta, 1e-5)
    img = x.clone()
    peq = L.twoway_pe_layout((pe @ wq.t() + bq).contiguous())
    L.twoway_i2t(img, _planes(wq), peq, kt, vt, _planes(wo), bo, gamma, beta, 1e-5, g, hw, nt, heads)
    torch.cuda.synchronize()
    assert rel_err(out, ref_t2i) < 2e-5
    assert rel_err(img, ref_i2t) < 2e-5
    with pytest.raises(RuntimeError, match="nt="):
        L.twoway_i2t(img, _planes(wq), peq, rnd(g * 40, di), rnd(g * 40, di), _planes(wo), bo, gamma, beta, 1e-5, g, hw, 40, heads)


@pytest.mark.parametrize("shape", [(2, 2, 197), (1, 12, 901)])
def test_fp8_qk_attention_matches_torch_on_e4m3_rounded_operands(L, shape):
    """la_qk_fp8 + la_attn_fwd_fp8 (opt-in, BASELINE configs[4]): S = Q K^T on the scaled fp8 MFMA.  Against torch with q and k
    rounded to e4m3 the kernel is at 16-bit-P accuracy (this pins the operand layout of v_mfma_scale_f32_32x32x64_f8f6f4); against
    the UNROUNDED operands the output moves by a few percent - the price of 3 mantissa bits, reported, not asserted tight."""
    import math
    b, heads, t = shape
    e = heads * 64
    tpad = (t + 63) // 64 * 64
    g = torch.Generator().manual_seed(t)
    qkv = (torch.randn(b * t, 3 * e, generator=g) * 0.8).half().cuda()
    scale = 1.0 / math.sqrt(64)
    vt = torch.empty(b * heads, 64, tpad, dtype=torch.float16, device="cuda")
    L.head_transpose(qkv, 2 * e, b, heads, t, tpad, vt)
    qk8 = torch.empty(b * t, 2 * e, dtype=torch.uint8, device="cuda")
    L.qk_fp8(qkv, e, qk8)
    ref8 = qkv[:, :2 * e].float().to(torch.float8_e4m3fn)
    assert torch.equal(qk8.view(torch.float8_e4m3fn).float(), ref8.float())
    out = torch.empty(b * t, e, dtype=torch.float16, device="cuda")
    L.attn_fwd_fp8(qk8, vt, out, b, heads, t, tpad, e, scale)
    torch.cuda.synchronize()

    def ref_attn(qk):
        q = qk[:, :e].double().view(b, t, heads, 64).permute(0, 2, 1, 3)
        k = qk[:, e:].double().view(b, t, heads, 64).permute(0, 2, 1, 3)
        v = qkv[:, 2 * e:].double().cpu().view(b, t, heads, 64).permute(0, 2, 1, 3)
        o = torch.softmax(q @ k.transpose(-1, -2) * scale, -1) @ v
        return o.permute(0, 2, 1, 3).reshape(b * t, e)

    o8 = ref_attn(ref8.float().cpu())
    o16 = ref_attn(qkv[:, :2 * e].float().cpu())
    got = out.double().cpu()
    err8 = float((got - o8).abs().max() / o8.abs().max())
    err16 = float((got - o16).abs().max() / o16.abs().max())
    print(f"fp8 QK^T attention {shape}: vs e4m3-rounded operands {err8:.2e}, vs the 16-bit operands {err16:.2e}")
    assert err8 <= 3e-3
    assert err16 <= 0.2


@pytest.mark.parametrize("geom", [(3, 901, 768, 0), (2, 4096, 768, 0), (51, 901, 1024, 0), (2, 28 * 28, 128, 14), (1, 64 * 64, 768, 14)])
def test_token_mean_kernels(L, geom):
    """la_colmean16 (image order and window-partitioned sources), la_layernorm_g (+ column sums), la_colsum_fold, la_add_rowvec: the pieces of the token-mean
    correction of single-plane weights.  The means must not depend on the number of groups in the launch (bitwise)."""
    import torch.nn.functional as F
    groups, rpg, d, ws = geom
    gen = torch.Generator().manual_seed(groups * rpg)
    x = torch.randn(groups * rpg, d, generator=gen).half()
    g = int(round(rpg ** 0.5))
    if ws:
        nw = (g + ws - 1) // ws
        img = x.view(groups, g, g, d)
        pad = nw * ws - g
        win = F.pad(img, (0, 0, 0, pad, 0, pad), value=7.0)           # pad slots hold garbage that must be skipped
        win = win.view(groups, nw, ws, nw, ws, d).permute(0, 1, 3, 2, 4, 5).reshape(-1, d).contiguous().cuda()
        src = win
    else:
        src = x.cuda()
    scr = torch.empty(groups * ((rpg + 127) // 128) * d, device="cuda")
    out = torch.empty(groups, d, device="cuda")
    L.colmean16(src, groups, rpg, out, scr, ws, g if ws else 0, g if ws else 0)
    ref = x.float().view(groups, rpg, d).mean(1)
    assert float((out.cpu() - ref).abs().max()) <= 2e-6
    if not ws and groups > 1:                                         # one group alone gives the same bits
        out1 = torch.empty(1, d, device="cuda")
        L.colmean16(src[rpg:2 * rpg], 1, rpg, out1, scr)
        assert torch.equal(out1[0], out[1])
    # LayerNorm of x + per-group vector, and the in-place fold
    res = torch.randn(groups * rpg, d, generator=gen).cuda()
    rv = torch.randn(groups, d, generator=gen).cuda() * 0.1
    gamma, beta = (1 + 0.1 * torch.randn(d, generator=gen)).cuda(), (0.1 * torch.randn(d, generator=gen)).cuda()
    o32 = torch.empty_like(res)
    L.layernorm_g(res, rv, rpg, gamma, beta, 1e-6, out32=o32, dt=L.LA_F32)
    full = res + rv.repeat_interleave(rpg, 0)
    assert float((o32 - F.layer_norm(full, (d,), gamma, beta, 1e-6)).abs().max()) <= 2e-5
    L.add_rowvec(res, rv, rpg)
    assert torch.equal(res, full)
    # the same LayerNorm leaving the column sums of the 16-bit rows it stores (image order, or window-partitioned output)
    chunks = L.ln_cs_chunks(rpg)
    res2 = torch.randn(groups * rpg, d, generator=gen).cuda()
    part = torch.full((groups * chunks * d,), float("nan"), device="cuda")
    arows = groups * ((g + ws - 1) // ws * ws) ** 2 if ws else groups * rpg
    o16 = torch.zeros(arows, d, device="cuda", dtype=torch.float16)
    L.layernorm_g(res2, rv, rpg, gamma, beta, 1e-6, out16=o16, colsum_part=part, **(dict(window=ws, H=g, W=g) if ws else {}))
    plain = torch.zeros_like(o16)
    L.layernorm_g(res2, rv, rpg, gamma, beta, 1e-6, out16=plain, **(dict(window=ws, H=g, W=g) if ws else {}))
    assert torch.equal(o16, plain)
    bar = torch.zeros(groups, d + 8, device="cuda")
    L.colsum_fold(part, groups, chunks, d, 1.0 / rpg, bar[:, 8:])
    L.colmean16(o16, groups, rpg, out, scr, ws, g if ws else 0, g if ws else 0)        # the separate pass over the stored rows
    # (the sums are those of the fp32 values before the 16-bit store: equal to the stored rows' mean up to the mean rounding error)
    assert float((bar[:, 8:] - out).abs().max()) <= 1e-4 and float(bar[:, :8].abs().max()) == 0.0
    if groups > 1:                                                    # a group's sums do not depend on the launch
        part1 = torch.empty(chunks * d, device="cuda")
        o1 = torch.zeros(arows // groups, d, device="cuda", dtype=torch.float16)
        L.layernorm_g(res2[rpg:2 * rpg].contiguous(), rv[1:2].contiguous(), rpg, gamma, beta, 1e-6, out16=o1, colsum_part=part1,
                      **(dict(window=ws, H=g, W=g) if ws else {}))
        assert torch.equal(part1, part[chunks * d:2 * chunks * d])


@pytest.mark.parametrize("shape", [("plain", 3, 901, 0), ("global", 2, 4096, 64), ("win16", 2, 196, 14), ("generic", 2, 400, 20)])
def test_attention_column_sums(L, shape):
    """la_attn_fwd_cs: the output equals la_attn_fwd's bit for bit, and the per-block column sums fold to the token means of the stored
    rows (WIN16: rows of padded window slots left out)."""
    kind, nimg, t, gg = shape
    heads, e = 12, 768
    gen = torch.Generator(device="cuda").manual_seed(t)
    img_g = 64                                                                     # WIN16: 64 x 64 tokens in 5 x 5 windows of 14 x 14
    b = nimg * 25 if kind == "win16" else nimg
    qkv = (torch.randn(b * t, 3 * e, device="cuda", generator=gen) * 0.8).half()
    mode = {"plain": L.ATTN_PLAIN, "global": L.ATTN_RELPOS, "win16": L.ATTN_RELPOS_WIN16, "generic": L.ATTN_RELPOS}[kind]
    tpad = (16 * gg + 63) // 64 * 64 if kind == "win16" else (t + 63) // 64 * 64
    vt = torch.zeros(b * heads, 64, tpad, dtype=torch.float16, device="cuda")
    if kind == "win16":
        vt.view(b, heads, 64, tpad)[..., :16 * gg].unflatten(-1, (gg, 16))[..., :gg] = qkv[:, 2 * e:].view(b, gg, gg, heads, 64).permute(0, 3, 4, 1, 2)
    else:
        L.head_transpose(qkv, 2 * e, b, heads, t, tpad, vt)
    tab = dict(tabh=(torch.randn(2 * gg - 1, 64, device="cuda", generator=gen) * 0.3).half(),
               tabw=(torch.randn(2 * gg - 1, 64, device="cuda", generator=gen) * 0.3).half()) if kind in ("global", "win16") else {}
    relh = relw = None
    if kind == "generic":
        relh = torch.randn(b * heads, t, gg, device="cuda", generator=gen) * 0.3
        relw = torch.randn(b * heads, t, gg, device="cuda", generator=gen) * 0.3
    sc = 0.125
    ref = torch.empty(b * t, e, dtype=torch.float16, device="cuda")
    L.attn_fwd(qkv, vt, ref, relh, relw, b, heads, t, tpad, gg, e, sc, mode, **tab)
    out = torch.empty_like(ref)
    nq = (t + 127) // 128
    part = torch.full((b * nq * e,), float("nan"), device="cuda")
    L.attn_fwd_cs(qkv, vt, out, relh, relw, b, heads, t, tpad, gg, e, sc, mode, part, img_g if kind == "win16" else 0,
                  img_g if kind == "win16" else 0, **tab)
    assert torch.equal(out, ref)
    if kind == "win16":
        o = out.float().view(nimg, 5, 5, gg, gg, e)                                # (image, wy, wx, ty, tx): keep tokens inside the 64 x 64 grid
        yy = (torch.arange(5, device="cuda")[:, None] * gg + torch.arange(gg, device="cuda")[None]) < img_g
        keep = (yy[:, None, :, None] & yy[None, :, None, :]).float()               # [wy, wx, ty, tx]
        want = (o * keep[None, ..., None]).sum((1, 2, 3, 4)) / (img_g * img_g)
        bar = torch.empty(nimg, e, device="cuda")
        L.colsum_fold(part, nimg, 25 * nq, e, 1.0 / (img_g * img_g), bar)
    else:
        want = out.float().view(b, t, e).mean(1)
        bar = torch.empty(b, e, device="cuda")
        L.colsum_fold(part, b, nq, e, 1.0 / t, bar)
    assert float((bar - want).abs().max()) <= 3e-6 * max(1.0, float(want.abs().max())), float((bar - want).abs().max())


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("shape", [("plain", 3, 901, 0, 64), ("plain", 2, 4096, 0, 64), ("global", 2, 4096, 64, 64), ("win16", 7, 196, 14, 64),
                                   ("win16", 5, 64, 8, 64), ("plain", 2, 300, 0, 128), ("global", 1, 4096, 64, 128), ("win16", 4, 196, 14, 128)])
def test_attention_without_a_transposed_v_copy_is_bit_identical(L, dt, shape):
    """la_attn_fwd_rows: V tiles staged row-major from the v columns of qkv and fed to the MFMA by LDS transpose reads - the same products
    in the same order as la_attn_fwd on the V^T copy, so the outputs (and the per-block column sums) are equal bit for bit."""
    kind, b, t, gg, hd = shape
    heads = 3
    e = heads * hd
    gen = torch.Generator(device="cuda").manual_seed(t + hd)
    qkv = (torch.randn(b * t, 3 * e, device="cuda", generator=gen) * 0.8).to(dt)
    mode = {"plain": L.ATTN_PLAIN, "global": L.ATTN_RELPOS, "win16": L.ATTN_RELPOS_WIN16}[kind]
    tpad = (16 * gg + 63) // 64 * 64 if kind == "win16" else (t + 63) // 64 * 64
    vt = torch.zeros(b * heads, hd, tpad, dtype=dt, device="cuda")
    if kind == "win16":
        vt.view(b, heads, hd, tpad)[..., :16 * gg].unflatten(-1, (gg, 16))[..., :gg] = qkv[:, 2 * e:].view(b, gg, gg, heads, hd).permute(0, 3, 4, 1, 2)
    else:
        vt.view(b, heads, hd, tpad)[..., :t] = qkv[:, 2 * e:].view(b, t, heads, hd).permute(0, 2, 3, 1)
    tab = dict(tabh=(torch.randn(2 * gg - 1, hd, device="cuda", generator=gen) * 0.3).to(dt),
               tabw=(torch.randn(2 * gg - 1, hd, device="cuda", generator=gen) * 0.3).to(dt)) if kind != "plain" else {}
    sc = hd ** -0.5
    nq = (t + 127) // 128
    ref = torch.empty(b * t, e, dtype=dt, device="cuda")
    part_ref = torch.zeros(b * nq * e, device="cuda")
    L.attn_fwd_cs(qkv, vt, ref, None, None, b, heads, t, tpad, gg, e, sc, mode, part_ref, **tab)
    out = torch.full_like(ref, float("nan"))
    part = torch.full((b * nq * e,), float("nan"), device="cuda")
    L.attn_fwd_rows(qkv, out, b, heads, t, tpad, gg, e, sc, mode, cspart=part, **tab)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    assert torch.equal(part, part_ref)


@pytest.mark.parametrize("dt", DTYPES)
@pytest.mark.parametrize("geom", [(2, 64, 64, 14), (1, 20, 27, 8), (3, 14, 14, 14), (4, 64, 64, 14, 4), (8, 64, 64, 14, 12), (5, 33, 50, 14, 6)])
def test_window_attention_addressed_in_image_order(L, dt, geom):
    """la_attn_fwd_rows, LA_ATTN_RELPOS_WIN16 with an image grid: window_partition / window_unpartition (image_encoder.py:258-304) as address
    arithmetic.  Reference: the windows cut out explicitly - tokens beyond the image are the pad row, as pad-after-norm makes them - and run
    through la_attn_fwd on window buffers; every real token's output must be equal bit for bit, and the column sums fold to the image's
    token means.  Without column sums asked for, 64-wide heads run attn_win_stream_kernel (persistent workgroups walk the (window, head,
    query block) items as one tile stream; the larger geometries give a workgroup 2 / 7 / 2 items): equal bit for bit as well."""
    nimg, ih, iw, gg = geom[:4]
    heads, hd = (geom[4] if len(geom) > 4 else 2), 64
    e = heads * hd
    t = gg * gg
    nwy, nwx = -(-ih // gg), -(-iw // gg)
    b = nimg * nwy * nwx
    gen = torch.Generator(device="cuda").manual_seed(ih * 100 + iw)
    qkv = (torch.randn(nimg * ih * iw, 3 * e, device="cuda", generator=gen) * 0.8).to(dt)
    padrow = (torch.randn(3 * e, device="cuda", generator=gen) * 0.5).to(dt)
    tabh = (torch.randn(2 * gg - 1, hd, device="cuda", generator=gen) * 0.3).to(dt)
    tabw = (torch.randn(2 * gg - 1, hd, device="cuda", generator=gen) * 0.3).to(dt)
    # explicit windows: [image, wy, wx, ty, tx] -> image row or -1
    yy = torch.arange(nwy, device="cuda")[:, None] * gg + torch.arange(gg, device="cuda")[None]          # [wy, ty]
    xx = torch.arange(nwx, device="cuda")[:, None] * gg + torch.arange(gg, device="cuda")[None]          # [wx, tx]
    inside = (yy < ih)[:, None, :, None] & (xx < iw)[None, :, None, :]                                    # [wy, wx, ty, tx]
    row = (yy.clamp(max=ih - 1)[:, None, :, None] * iw + xx.clamp(max=iw - 1)[None, :, None, :])
    rows = (torch.arange(nimg, device="cuda").view(nimg, 1, 1, 1, 1) * ih * iw + row[None]).reshape(-1)
    ins = inside[None].expand(nimg, -1, -1, -1, -1).reshape(-1)
    qkv_w = torch.where(ins[:, None], qkv[rows], padrow[None].expand(rows.numel(), -1)).contiguous()     # [b * t, 3E]
    tpad = (16 * gg + 63) // 64 * 64
    vt = torch.zeros(b * heads, hd, tpad, dtype=dt, device="cuda")
    vt.view(b, heads, hd, tpad)[..., :16 * gg].unflatten(-1, (gg, 16))[..., :gg] = qkv_w[:, 2 * e:].view(b, gg, gg, heads, hd).permute(0, 3, 4, 1, 2)
    ref_w = torch.empty(b * t, e, dtype=dt, device="cuda")
    L.attn_fwd(qkv_w, vt, ref_w, None, None, b, heads, t, tpad, gg, e, 0.125, L.ATTN_RELPOS_WIN16, tabh=tabh, tabw=tabw)
    out = torch.full((nimg * ih * iw, e), float("nan"), dtype=dt, device="cuda")
    nq = (t + 127) // 128
    part = torch.full((b * nq * e,), float("nan"), device="cuda")
    L.attn_fwd_rows(qkv, out, b, heads, t, tpad, gg, e, 0.125, L.ATTN_RELPOS_WIN16, tabh=tabh, tabw=tabw, cspart=part, img_hw=(ih, iw),
                    padrow=padrow)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out.float()).all())                                 # every image token was written
    assert torch.equal(out[rows[ins]], ref_w[ins])
    bar = torch.empty(nimg, e, device="cuda")
    L.colsum_fold(part, nimg, nwy * nwx * nq, e, 1.0 / (ih * iw), bar)
    want = out.float().view(nimg, ih * iw, e).mean(1)
    assert float((bar - want).abs().max()) <= 3e-6 * max(1.0, float(want.abs().max()))
    out2 = torch.full_like(out, float("nan"))
    L.attn_fwd_rows(qkv, out2, b, heads, t, tpad, gg, e, 0.125, L.ATTN_RELPOS_WIN16, tabh=tabh, tabw=tabw, img_hw=(ih, iw), padrow=padrow)
    torch.cuda.synchronize()
    assert torch.equal(out2, out)


@pytest.mark.parametrize("with_v", [True, False])
def test_add_rowvec_split_leaves_the_stream_as_plane_pairs(L, with_v):
    """la_add_rowvec_split: x[r] += v[r / rows_per_group] in place (v optional) and the result as LA_F16X2 rows [hi | lo] with hi = rn(x),
    lo = rn(x - hi) - the A operand of the three-product fp16 GEMM (the SAM neck's 1 x 1 conv); that GEMM against [W_hi | W_hi | W_lo]
    reproduces the exact-fp32 product to 2e-6."""
    rows, d, rpg, n = 1024, 768, 256, 256
    x = rnd(rows, d, seed=70) * 3.0
    v = rnd(rows // rpg, d, seed=71) if with_v else None
    want = x + (v.repeat_interleave(rpg, 0) if with_v else 0.0)
    xs = torch.full((rows, 2 * d), float("nan"), device="cuda", dtype=torch.float16)
    x_in = x.clone()
    L.add_rowvec_split(x_in, v, rpg, xs)
    torch.cuda.synchronize()
    assert torch.equal(x_in, want)
    hi = want.to(torch.float16)
    assert torch.equal(xs[:, :d], hi) and torch.equal(xs[:, d:], (want - hi.float()).to(torch.float16))
    w = rnd(n, d, seed=72) / math.sqrt(d)
    whi = w.to(torch.float16)
    ws = torch.cat([whi, whi, (w - whi.float()).to(torch.float16)], dim=1).contiguous()
    out = torch.empty(rows, n, device="cuda")
    L.gemm(xs, ws, out32=out, a_kmod=2 * d)
    ref = (want.double() @ w.double().t()).float()
    torch.cuda.synchronize()
    assert rel_err(out, ref) < 2e-6
    # beyond the fp16 range the plane pair saturates (hi at 65504, lo takes the rest up to 131008) instead of becoming (inf, -inf):
    # the stream itself keeps its fp32 values, the planes stay finite
    big = torch.zeros(4, d, device="cuda")
    big[0, 0], big[1, 1], big[2, 2], big[3, 3] = 7.0e4, -1.2e5, 3.0e5, -65504.0
    keep = big.clone()
    bs = torch.full((4, 2 * d), float("nan"), device="cuda", dtype=torch.float16)
    L.add_rowvec_split(big, None, 1, bs)
    torch.cuda.synchronize()
    assert torch.equal(big, keep) and bool(torch.isfinite(bs.float()).all())
    pair = bs[:, :d].float() + bs[:, d:].float()
    assert abs(float(pair[0, 0]) - 7.0e4) <= 32 and abs(float(pair[1, 1]) + 1.2e5) <= 64 and float(pair[2, 2]) == 131008.0 and float(pair[3, 3]) == -65504.0


def test_gemm_residual_that_repeats_every_res_mod_rows_on_the_four_wave_kernel(L):
    """The patch embedding's GEMM (residual = the position table, one row per token of the image: res_mod) stays on the persistent
    four-wave kernel when whole 256-row tiles sit inside one period; compared with the exact product."""
    m, n, k, period = 2048, 512, 768, 512
    a = (rnd(m, k, seed=80) * 0.5).half()
    w = (rnd(n, k, seed=81) / math.sqrt(k)).half()
    bias = rnd(n, seed=82)
    pos = rnd(period, n, seed=83)
    out = torch.empty(m, n, device="cuda")
    L.gemm(a, w, bias=bias, res=pos, res_mod=period, out32=out)
    torch.cuda.synchronize()
    ref = (a.double() @ w.double().t() + bias.double() + pos.double().repeat(m // period, 1)).float()
    assert rel_err(out, ref) < 2e-6


@pytest.mark.parametrize("shape", [(7, 64, 64, 30, 30), (3, 64, 64, 64, 64), (5, 37, 50, 36, 17), (2, 128, 128, 64, 63), (4, 16, 16, 1, 1)])
def test_bilinear_backward_as_a_gather_matches_autograd(L, shape):
    """la_bilinear_bwd_set (reductions: every dx entry a sum of <= 2 x 2 taps, written, no atomics) against torch autograd of
    F.interpolate(mode="bilinear", align_corners=False) and against the scatter kernel la_bilinear_bwd on a zero-filled destination."""
    n, ih, iw, oh, ow = shape
    assert L.bilinear_bwd_set_ok(oh, ow, ih, iw) and not L.bilinear_bwd_set_ok(ih + 1, ow, ih, iw) and not L.bilinear_bwd_set_ok(oh, ow, 129, iw)
    x = rnd(n, ih, iw, seed=211).double().requires_grad_(True)
    dy = rnd(n, oh, ow, seed=212)
    F.interpolate(x.unsqueeze(0), size=(oh, ow), mode="bilinear", align_corners=False)[0].backward(dy.double())
    dx = torch.full((n, ih, iw), float("nan"), device="cuda")
    L.bilinear_bwd_set(dy, n, oh, ow, oh * ow, ow, dx, ih, iw, ih * iw, iw)
    dx2 = torch.zeros(n, ih, iw, device="cuda")
    L.bilinear_bwd(dy, n, oh, ow, oh * ow, ow, dx2, ih, iw, ih * iw, iw)
    torch.cuda.synchronize()
    ref = x.grad.float()
    # (tap weights are formed in fp32 from coordinates up to 128: a few 1e-6 of the fp64 reference; the two kernels share them)
    assert rel_err(dx, ref) < 3e-5 and rel_err(dx2, ref) < 3e-5 and rel_err(dx, dx2) < 1e-6
    # a destination frame with padding around the plane (explicit strides): only the ih x iw region is written
    frame = torch.full((n, ih + 3, iw + 5), 7.0, device="cuda")
    L.bilinear_bwd_set(dy, n, oh, ow, oh * ow, ow, frame, ih, iw, (ih + 3) * (iw + 5), iw + 5)
    torch.cuda.synchronize()
    assert torch.equal(frame[:, :ih, :iw], dx) and bool((frame[:, ih:, :] == 7.0).all()) and bool((frame[:, :, iw:] == 7.0).all())


@pytest.mark.parametrize("shape", [(3, 64, 64, 30, 30, 256), (2, 16, 24, 16, 24, 8), (2, 40, 33, 17, 20, 12), (2, 8, 8, 20, 20, 16)])
def test_bilinear_on_nhwc_rows_forward_and_backward(L, shape):
    """la_bilinear_rows against F.interpolate on the planes and - bit for bit - against la_bilinear on transposed planes (same taps, same
    blend); its autograd node (gathered adjoint for reductions, planar adjoint otherwise) against torch autograd."""
    from labelanything_amd import autograd_ops as A
    n, h, w, oh, ow, c = shape
    x = rnd(n * h * w, c, seed=221)
    out = torch.empty(n * oh * ow, c, device="cuda")
    L.bilinear_rows(x, n, h, w, c, oh, ow, out)
    planes = x.view(n, h * w, c).permute(0, 2, 1).reshape(n * c, h, w).contiguous()
    pout = torch.empty(n * c, oh, ow, device="cuda")
    L.bilinear(planes, n * c, h, w, oh, ow, pout)
    torch.cuda.synchronize()
    assert torch.equal(out, pout.view(n, c, oh * ow).permute(0, 2, 1).reshape(n * oh * ow, c))
    xd = planes.view(n, c, h, w).double().requires_grad_(True)
    ref = F.interpolate(xd, size=(oh, ow), mode="bilinear", align_corners=False)
    assert rel_err(pout.view(n, c, oh, ow), ref.detach().float()) < 3e-5
    r = rnd(n * oh * ow, c, seed=222)
    ref.backward(r.view(n, oh * ow, c).permute(0, 2, 1).reshape(n, c, oh, ow).double())
    xr = x.clone().requires_grad_(True)
    (A.bilinear_rows(xr, n, h, w, c, oh, ow) * r).sum().backward()
    torch.cuda.synchronize()
    gref = xd.grad.float().view(n, c, h * w).permute(0, 2, 1).reshape(n * h * w, c)
    assert rel_err(xr.grad, gref) < 3e-5


@pytest.mark.parametrize("bhw", [(2, 64, 64), (1, 37, 53), (3, 8, 32), (1, 256, 256)])
def test_conv3x3_split_precision_matches_torch_and_the_fp32_kernel(L, bhw):
    """la_conv3x3_split (mask_decoder.py:236-255 spatial convolutions at 32 channels): three fp16 products on plane pairs against
    F.conv2d in fp64 and against the exact-fp32 implicit GEMM it replaces in the inference engine."""
    b, h, w = bhw
    c = 32
    x = rnd(b * h * w, c, seed=61)
    wt = rnd(c, c, 3, 3, seed=62) / math.sqrt(9 * c)
    bias = rnd(c, seed=63) * 0.1
    wk = wt.permute(0, 2, 3, 1).flatten(1).contiguous()                 # [cout, (ky, kx, cin)]
    ref = F.conv2d(x.view(b, h, w, c).permute(0, 3, 1, 2).double(), wt.double(), bias.double(), padding=1).permute(0, 2, 3, 1).reshape(b * h * w, c)
    out = torch.full((b * h * w, c), float("nan"), device="cuda")
    out32 = torch.empty_like(out)
    assert L.conv3x3_split_ok(c, c) and not L.conv3x3_split_ok(64, 64)
    L.conv3x3_split(x, b, h, w, c, wk, bias, c, out)
    L.conv3x3_f32(x, b, h, w, c, wk, bias, c, out32)
    torch.cuda.synchronize()
    scale = float(ref.abs().max())
    e_split, e_f32 = float((out.double() - ref).abs().max()) / scale, float((out32.double() - ref).abs().max()) / scale
    print(f"[conv3x3 {bhw}] split precision {e_split:.2e}, exact fp32 {e_f32:.2e}")
    assert e_split < 2e-6 and e_f32 < 2e-6


@pytest.mark.parametrize("bhw", [(2, 64, 64), (3, 30, 30), (1, 24, 40)])
def test_implicit_3x3_convolution_on_padded_plane_pairs_equals_im2col(L, bhw):
    """LA_MAP_CONV3X3 (the SAM / LAM neck's second convolution, image_encoder.py:100-106): LayerNorm into zero-bordered plane-pair maps
    (window=-1), the GEMM's k-tiles reading the nine taps as shifts of its source base, LayerNorm gathering the interior (window=-2) -
    against LayerNorm -> im2col -> GEMM -> LayerNorm on the same operands: the same products in the same order, bit for bit."""
    b, h, w = bhw
    c = 256
    rows = b * h * w
    x = rnd(rows, c, seed=81) * 2.0
    g1, b1 = 1.0 + 0.1 * rnd(c, seed=82), 0.05 * rnd(c, seed=83)
    g3, b3 = 1.0 + 0.1 * rnd(c, seed=84), 0.05 * rnd(c, seed=85)
    wt = rnd(c, 9 * c, seed=86) / math.sqrt(9 * c)
    hi = wt.half()
    ws = torch.cat([hi, hi, (wt - hi.float()).half()], dim=1).contiguous()
    # reference path: plain plane pairs -> im2col -> GEMM (a_kmod) -> LayerNorm
    a1s = torch.empty(rows, 2 * c, device="cuda", dtype=torch.float16)
    L.layernorm(x, g1, b1, 1e-6, out16=a1s, dt=L.LA_F16X2)
    col = torch.empty(rows, 18 * c, device="cuda", dtype=torch.float16)
    L.im2col_3x3(a1s, b, h, w, c, col, split=True)
    y_ref = torch.empty(rows, c, device="cuda")
    L.gemm(col, ws, out32=y_ref, a_kmod=18 * c)
    out_ref = torch.empty(rows, c, device="cuda")
    L.layernorm(y_ref, g3, b3, 1e-6, out32=out_ref, dt=L.LA_F32)
    # implicit path
    hp, wp = h + 2, w + 2
    mp, guard = b * hp * wp, wp + 1
    a1p = torch.zeros(mp + 2 * guard, 2 * c, device="cuda", dtype=torch.float16)
    L.layernorm(x, g1, b1, 1e-6, out16=a1p[guard:], dt=L.LA_F16X2, window=-1, H=h, W=w)
    inner = a1p[guard:guard + mp].view(b, hp, wp, 2 * c)
    assert torch.equal(inner[:, 1:-1, 1:-1].reshape(rows, 2 * c), a1s)
    assert float(inner[:, 0].abs().max()) == 0 and float(inner[:, :, 0].abs().max()) == 0 and float(a1p[:guard].abs().max()) == 0
    y_p = torch.full((mp, c), float("nan"), device="cuda")
    L.gemm(a1p[guard:guard + mp], ws, out32=y_p, amap=L.MAP_CONV3X3, p=(wp, c, 2 * c, 0, 0))
    out = torch.empty(rows, c, device="cuda")
    L.layernorm(y_p, g3, b3, 1e-6, out32=out, dt=L.LA_F32, window=-2, H=h, W=w)
    torch.cuda.synchronize()
    assert torch.isfinite(y_p).all()
    assert torch.equal(y_p.view(b, hp, wp, c)[:, 1:-1, 1:-1].reshape(rows, c), y_ref)
    assert torch.equal(out, out_ref)
    # and against the convolution itself in fp64
    z = F.layer_norm(x.double(), (c,), g1.double(), b1.double(), 1e-6).view(b, h, w, c).permute(0, 3, 1, 2)
    ref = F.conv2d(z, wt.double().view(c, 3, 3, c).permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1).reshape(rows, c)
    assert rel_err(y_ref, ref) < 5e-6
