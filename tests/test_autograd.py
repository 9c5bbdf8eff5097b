"""Gradient path of the ops (videoswap_amd/autograd.py) — what the adapter training step needs:
every differentiable op against PyTorch's own autograd on an fp32 restatement of the same op, then the whole UNet:
d loss / d adapter residuals against the oracle's autograd.

`device` tests: on CPU they run on tests/host_emulation.py (the backward passes are built from kernel functions, so
what is checked here is the composition: transposed / flipped weights, zero insertion, re-materialised attention, layout
round trips); on a GPU box the backward kernels of libvsx.so (csrc/train.hip) run."""
import os

import pytest
import torch
import torch.nn.functional as F

from util import DEV, oracle_unet, product_unet_from, rel_l2

pytestmark = pytest.mark.device
H = torch.float16


def ops():
    from videoswap_amd import ops as o
    return o


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale)


def leaf(t):
    return t.to(H).to(DEV).requires_grad_(True)


def ref_leaf(t):
    return t.to(H).float().requires_grad_(True)


def check(got, want, tol=4e-3, what=''):
    e = rel_l2(got.float().cpu(), want)
    assert e < tol, f'{what}: rel-L2 {e:.2e}'


@pytest.mark.parametrize('geglu,res', [(False, False), (False, True), (True, False)])
def test_linear_gradient(geglu, res):
    x0, w0, b0 = rnd(24, 64, seed=1), rnd(96, 64, seed=2, scale=0.2), rnd(96, seed=3)
    n = 48 if geglu else 96
    r0, g0 = rnd(24, n, seed=4), rnd(24, n, seed=5)
    x, r = leaf(x0), leaf(r0) if res else None
    y = ops().linear(x, w0.to(H).to(DEV), b0.to(H).to(DEV), residual=r, geglu=geglu)
    (y.float() * g0.to(DEV)).sum().backward()
    xr, rr = ref_leaf(x0), ref_leaf(r0)
    yr = F.linear(xr, w0.to(H).float(), b0.to(H).float())
    if geglu:
        hh, gg = yr.chunk(2, -1)
        yr = hh * F.gelu(gg)
    if res:
        yr = yr + rr
    (yr * g0).sum().backward()
    check(y.detach(), yr.detach(), what='forward')
    check(x.grad, xr.grad, what='dx')
    if res:
        check(r.grad, rr.grad, what='dresidual')


def test_linear_weight_gradient_for_the_adapter_mlp():
    x0, w0, b0, g0 = rnd(6, 64, seed=6), rnd(32, 64, seed=7, scale=0.2), rnd(32, seed=8), rnd(6, 32, seed=9)
    x, w, b = leaf(x0), leaf(w0), leaf(b0)
    y = ops().linear(x, w, b)
    (y.float() * g0.to(DEV)).sum().backward()
    xr, wr, br = ref_leaf(x0), ref_leaf(w0), ref_leaf(b0)
    (F.linear(xr, wr, br) * g0).sum().backward()
    check(x.grad, xr.grad, what='dx')
    check(w.grad, wr.grad, what='dw')
    check(b.grad, br.grad, what='db')


def test_linear_records_when_only_weight_and_bias_need_a_gradient():
    """The adapter's first Linear: its input (the point embedding) carries NO grad, its weight and bias are the trainable
    parameters (adapter_model.py:70-107).  The call must still go through the gradient path — a raw kernel call returns
    a tensor without history and the whole training step would have nothing to differentiate."""
    x0, w0, b0, g0 = rnd(6, 64, seed=6), rnd(32, 64, seed=7, scale=0.2), rnd(32, seed=8), rnd(6, 32, seed=9)
    x = x0.to(DEV, torch.float16)                     # not a leaf that requires grad
    w, b = leaf(w0), leaf(b0)
    y = ops().linear(x, w, b)
    assert y.requires_grad and y.grad_fn is not None
    (y.float() * g0.to(DEV)).sum().backward()
    wr, br = ref_leaf(w0), ref_leaf(b0)
    (F.linear(x0.half().float(), wr, br) * g0).sum().backward()
    check(w.grad, wr.grad, what='dw')
    check(b.grad, br.grad, what='db')
    y2 = ops().linear(x, w, bias=b)                   # bias by keyword
    assert y2.requires_grad


def conv_ref(x, w, b, stride, x2, ups):
    xin = x if x2 is None else torch.cat([x, x2], -1)
    xin = xin.permute(0, 3, 1, 2)
    if ups:
        xin = F.interpolate(xin, scale_factor=2.0, mode='nearest')
    ks = w.shape[1]
    return F.conv2d(xin, w.permute(0, 3, 1, 2), b, stride=stride, padding=ks // 2).permute(0, 2, 3, 1)


@pytest.mark.parametrize('C1,C2,Cout,ks,stride,ups,res', [
    (64, 0, 64, 3, 1, False, True), (64, 64, 128, 3, 1, False, False), (64, 0, 64, 3, 2, False, False),
    (64, 0, 64, 3, 1, True, False), (128, 64, 64, 1, 1, False, False), (64, 0, 4, 3, 1, False, False)])
def test_conv2d_gradient(C1, C2, Cout, ks, stride, ups, res):
    n, hh, ww = 2, 8, 12
    x0 = rnd(n, hh, ww, C1, seed=10)
    x20 = rnd(n, hh, ww, C2, seed=11) if C2 else None
    w0, b0 = rnd(Cout, ks, ks, C1 + C2, seed=12, scale=(ks * ks * (C1 + C2)) ** -0.5), rnd(Cout, seed=13)
    ho, wo = (2 * hh if ups else hh) // stride, (2 * ww if ups else ww) // stride
    r0, g0 = rnd(n, ho, wo, Cout, seed=14), rnd(n, ho, wo, Cout, seed=15)
    x, x2, r = leaf(x0), leaf(x20) if C2 else None, leaf(r0) if res else None
    y = ops().conv2d(x, w0.to(H).to(DEV), b0.to(H).to(DEV), x2=x2, stride=stride, upsample=ups, residual=r)
    (y.float() * g0.to(DEV)).sum().backward()
    xr, x2r, rr = ref_leaf(x0), ref_leaf(x20) if C2 else None, ref_leaf(r0)
    yr = conv_ref(xr, w0.to(H).float(), b0.to(H).float(), stride, x2r, ups)
    if res:
        yr = yr + rr
    (yr * g0).sum().backward()
    check(y.detach(), yr.detach(), what='forward')
    check(x.grad, xr.grad, what='dx')
    if C2:
        check(x2.grad, x2r.grad, what='dx2')
    if res:
        check(r.grad, rr.grad, what='dresidual')


@pytest.mark.parametrize('nimg,rows,C1,C2,groups,silu', [(2, 96, 64, 0, 32, True), (6, 32, 64, 64, 32, False),
                                                         (2, 50, 128, 64, 32, True)])
def test_group_norm_gradient(nimg, rows, C1, C2, groups, silu):
    x0 = rnd(nimg * rows, C1, seed=20) * 2 + 0.3
    x20 = rnd(nimg * rows, C2, seed=21) if C2 else None
    ga0, be0, g0 = 1 + 0.3 * rnd(C1 + C2, seed=22), 0.1 * rnd(C1 + C2, seed=23), rnd(nimg * rows, C1 + C2, seed=24)
    x, x2 = leaf(x0), leaf(x20) if C2 else None
    y = ops().group_norm(x, ga0.to(H).to(DEV), be0.to(H).to(DEV), groups, 1e-5, nimg, silu=silu, x2=x2)
    (y.float() * g0.to(DEV)).sum().backward()
    xr, x2r = ref_leaf(x0), ref_leaf(x20) if C2 else None
    xin = xr if x2r is None else torch.cat([xr, x2r], -1)
    C = C1 + C2
    yr = F.group_norm(xin.view(nimg, rows, C).transpose(1, 2), groups, ga0.to(H).float(), be0.to(H).float(), 1e-5)
    yr = yr.transpose(1, 2).reshape(nimg * rows, C)
    if silu:
        yr = F.silu(yr)
    (yr * g0).sum().backward()
    check(x.grad, xr.grad, what='dx')
    if C2:
        check(x2.grad, x2r.grad, what='dx2')


def test_layer_norm_silu_axpy_unpack_gradients():
    o = ops()
    x0, ga0, be0, g0 = rnd(40, 64, seed=30) * 2, 1 + 0.2 * rnd(64, seed=31), 0.1 * rnd(64, seed=32), rnd(40, 64, seed=33)
    pe = rnd(4, 64, seed=34).to(H).to(DEV)
    x = leaf(x0)
    y = o.layer_norm(x, ga0.to(H).to(DEV), be0.to(H).to(DEV), 1e-5, pe=pe, rows_per_frame=5, frames=4)
    z = o.axpy(o.silu(y), x)
    (z.float() * g0.to(DEV)).sum().backward()
    xr = ref_leaf(x0)
    yr = F.layer_norm(xr, (64,), ga0.to(H).float(), be0.to(H).float(), 1e-5)
    idx = (torch.arange(40) // 5) % 4
    yr = yr + pe.float().cpu()[idx]
    ((F.silu(yr) + xr) * g0).sum().backward()
    check(x.grad, xr.grad, what='dx')
    # unpack_latents: [B*F, H, W, Cs] -> [B, cout, F, H, W]
    u0, gu = rnd(6, 4, 5, 4, seed=35), rnd(2, 4, 3, 4, 5, seed=36)
    u = leaf(u0)
    (o.unpack_latents(u, 2, 4).float() * gu.to(DEV)).sum().backward()
    want = gu.permute(0, 2, 3, 4, 1).reshape(6, 4, 5, 4)
    check(u.grad, want, what='dunpack')


def attn_ref(q, k, v, heads, scale, kv_div=1):
    nb, nq, c = q.shape
    d = c // heads

    def hd(t):
        return t.view(t.shape[0], t.shape[1], heads, d).permute(0, 2, 1, 3)
    kh, vh = hd(k).repeat_interleave(kv_div, 0), hd(v).repeat_interleave(kv_div, 0)
    p = ((hd(q) @ kh.transpose(-1, -2)) * scale).softmax(-1)
    return (p @ vh).permute(0, 2, 1, 3).reshape(nb, nq, c)


@pytest.mark.parametrize('nb,nq,nk,heads,d,kv_div', [
    (3, 40, 40, 4, 16, 1), (4, 24, 13, 2, 32, 2),
    # head dims of the flash backward (csrc/attention_bwd.hip): ragged query / key tiles (200 = 128 + 72, 3 key tiles + a
    # partial one), the text K / V shared by two images (dQ only, 77 keys), d = 80, one exact tile
    (2, 200, 200, 2, 40, 1), (4, 150, 77, 2, 40, 2), (1, 130, 130, 2, 80, 1), (1, 128, 64, 1, 64, 1)])
def test_attention_gradient(nb, nq, nk, heads, d, kv_div):
    C = heads * d
    q0, k0, v0, g0 = rnd(nb, nq, C, seed=40), rnd(nb // kv_div, nk, C, seed=41), rnd(nb // kv_div, nk, C, seed=42), \
        rnd(nb, nq, C, seed=43)
    self_attn = kv_div == 1
    ld = (nk + 7) // 8 * 8
    vt0 = torch.zeros(nb // kv_div, C, ld)
    vt0[:, :, :nk] = v0.transpose(1, 2)
    q = leaf(q0)
    k = leaf(k0) if self_attn else k0.to(H).to(DEV)
    vt = leaf(vt0) if self_attn else vt0.to(H).to(DEV)
    y = ops().attention(q, k, vt, heads, d ** -0.5, kv_div=kv_div)
    (y.float() * g0.to(DEV)).sum().backward()
    qr, kr, vr = ref_leaf(q0), ref_leaf(k0), ref_leaf(v0)
    yr = attn_ref(qr, kr, vr, heads, d ** -0.5, kv_div)
    (yr * g0).sum().backward()
    check(y.detach(), yr.detach(), what='forward')
    check(q.grad, qr.grad, what='dq')
    if self_attn:
        check(k.grad, kr.grad, what='dk')
        check(vt.grad[:, :, :nk].transpose(1, 2), vr.grad, what='dv')


def test_temporal_attention_gradient():
    B, f, hw, heads, d = 2, 4, 6, 2, 16
    C = heads * d
    qkv0, g0 = rnd(B * f * hw, 3 * C, seed=50), rnd(B * f * hw, C, seed=51)
    qkv = leaf(qkv0)
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    y = ops().temporal_attention(q, k, v, B, f, f, hw, heads, d ** -0.5)
    (y.float() * g0.to(DEV)).sum().backward()
    r = ref_leaf(qkv0)

    def sites(t):
        return t.reshape(B, f, hw, C).permute(0, 2, 1, 3).reshape(B * hw, f, C)
    yr = attn_ref(sites(r[:, :C]), sites(r[:, C:2 * C]), sites(r[:, 2 * C:]), heads, d ** -0.5)
    yr = yr.view(B, hw, f, C).permute(0, 2, 1, 3).reshape(B * f * hw, C)
    (yr * g0).sum().backward()
    check(y.detach(), yr.detach(), what='forward')
    check(qkv.grad, r.grad, what='dqkv')


def test_adapter_scatter_gradient():
    Fr, P, C, h, w = 3, 5, 32, 6, 8
    g = torch.Generator().manual_seed(60)
    tracks = torch.rand(Fr, P, 2, generator=g) * torch.tensor([w * 8.0, h * 8.0])
    tracks[1, 2] = -1.0
    sel = torch.tensor([1, 1, 0, 1, 1], dtype=torch.int32)
    feat0, g0 = rnd(P, C, seed=61), rnd(Fr, h, w, C, seed=62)
    feat = leaf(feat0)
    y = ops().adapter_scatter(tracks.to(DEV), sel.to(DEV), feat, h, w, 8.0, 0.5)
    (y.float() * g0.to(DEV)).sum().backward()
    # the map is linear in feat: d/dfeat[p] = sum of (weight x upstream gradient) over that point's splats; take the
    # weights from the forward itself by scattering one-hot features
    want = torch.zeros(P, C)
    for p in range(P):
        onehot = torch.zeros(P, C)
        onehot[p] = 1.0
        m = ops().adapter_scatter(tracks.to(DEV), sel.to(DEV), onehot.to(H).to(DEV), h, w, 8.0, 0.5).float().cpu()
        want[p] = (m * g0).sum((0, 1, 2))
    check(feat.grad, want, what='dfeat')
    assert float(feat.grad[2].abs().sum()) == 0.0


def test_unet_gradient_wrt_adapter_residuals_matches_oracle_autograd():
    """The quantity the training step needs: d loss / d (the four adapter residual maps) through the frozen UNet."""
    from oracle import unet3d
    cfg = unet3d.tiny_config()
    ora = oracle_unet(cfg)
    prod = product_unet_from(ora, cfg)
    for p in prod.parameters():
        p.requires_grad_(False)
    B, T, hw = 1, 2, 16
    x0, txt0 = rnd(B, 4, T, hw, hw, seed=70), rnd(B, 77, 64, seed=71)
    chans = list(cfg['block_out_channels'])
    res0 = [rnd(B * T, chans[i], hw >> i, hw >> i, seed=72 + i) * 0.3 for i in range(4)]        # reference layout
    g0 = rnd(B, 4, T, hw, hw, seed=80)
    # oracle: fp32 autograd
    res_r = [r.to(H).float().requires_grad_(True) for r in res0]
    for p in ora.parameters():
        p.requires_grad_(False)
    out_r = ora(x0.to(H).float(), torch.tensor(301), txt0.to(H).float(),
                down_block_additional_residuals=[r for r in res_r]).sample
    (out_r * g0).sum().backward()
    # product: channels-last residual leaves (what the adapter hands over)
    res_p = [r.permute(0, 2, 3, 1).contiguous().to(H).to(DEV).requires_grad_(True) for r in res0]
    for r in res_p:
        r.vsx_nhwc = True
    out_p = prod(x0.to(H).to(DEV), 301, txt0.to(H).to(DEV), down_block_additional_residuals=list(res_p)).sample
    (out_p.float() * g0.to(DEV)).sum().backward()
    check(out_p.detach(), out_r.detach(), tol=5e-3, what='forward')
    for i, (gp, rr) in enumerate(zip(res_p, res_r)):
        want = rr.grad.permute(0, 2, 3, 1)
        e = rel_l2(gp.grad.float().cpu(), want)
        print(f'level {i}: d loss / d residual rel-L2 {e:.2e}')
        assert e < 2e-2, (i, e)
