"""Winograd F(4x4,3x3) and F(2x2,3x3) paths (transforms + grouped GEMM) vs torch conv2d."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref(x, w, b, scale_in=None, div=1, relu=True):
    xi = x.repeat_interleave(div, 0) if div > 1 else x
    if scale_in is not None:
        xi = xi * scale_in[:, None, None, :]
    y = F.conv2d(xi.permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=1)
    return (F.relu(y) if relu else y).float()


@pytest.mark.parametrize('m', [4, 2])
@pytest.mark.parametrize('n,h,w,cin,cout,div,scaled', [
    (2, 8, 10, 64, 128, 1, False),
    (1, 7, 9, 32, 8, 1, False),            # odd sizes: partial last tiles; Cout < tile
    (1, 13, 21, 128, 256, 3, True),        # AG-RPN pattern: one map, N guided passes
    (37, 7, 7, 64, 64, 1, False),          # RoI pattern: 7x7 maps, 16 / 4 tiles per RoI
    (1, 50, 84, 256, 128, 3, True),
    (3, 5, 3, 32, 32, 1, False),           # maps smaller than / not a multiple of a tile
])
def test_winograd_matches_direct(n, h, w, cin, cout, div, scaled, m):
    from fgn_amd import ops
    g = torch.Generator().manual_seed(n * h + cin)
    x = torch.randn(n, h, w, cin, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    s = (torch.rand(n * div, cin, generator=g) + 0.5) if scaled else None
    layer = ops.pack_winograd(wt, bias=b, relu=True, m=m).to('cuda')
    assert layer.groups == (m + 2) ** 2
    got = ops.conv3x3_winograd(x.cuda(), layer, in_scale=None if s is None else s.cuda(), a_img_div=div)
    ref = _ref(x, wt, b, s, div)
    assert got.shape == (n * div, h, w, cout)
    d = (got.cpu().permute(0, 3, 1, 2) - ref).abs().max().item()
    # measured: F(2x2) ~5e-7, F(4x4) with the points {0, 1, -1, 1/2, -2} ~3e-6 of the output range
    assert d <= (2e-5 if m == 4 else 5e-6) * max(ref.abs().max().item(), 1.0), d
    # and against the direct implicit-GEMM kernel of this library (same tolerance)
    direct = ops.pack_conv(wt, bias=b, pad=1, relu=True).to('cuda')
    xin = x.cuda() if s is None else ops.scale_channels(x.cuda(), s.cuda(), div)
    if s is None and div > 1:
        xin = x.cuda().repeat_interleave(div, 0)
    dd = (ops.conv2d(xin, direct) - got).abs().max().item()
    assert dd <= 1e-4 * max(ref.abs().max().item(), 1.0), dd
    # deterministic
    again = ops.conv3x3_winograd(x.cuda(), layer, in_scale=None if s is None else s.cuda(), a_img_div=div)
    assert torch.equal(again, got)


@pytest.mark.parametrize('m', [4, 2])
def test_winograd_bn_fold_and_device_count(m):
    from fgn_amd import ops
    g = torch.Generator().manual_seed(5)
    n, cin, cout = 40, 64, 64
    x = torch.randn(n, 7, 7, cin, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.06
    bn = dict(weight=torch.rand(cout, generator=g) + 0.5, bias=torch.randn(cout, generator=g) * 0.1,
              running_mean=torch.randn(cout, generator=g) * 0.1, running_var=torch.rand(cout, generator=g) + 0.5)
    layer = ops.pack_winograd(wt, bn=bn, relu=True, m=m).to('cuda')
    cnt = torch.tensor([23], dtype=torch.int32, device='cuda')
    got = ops.conv3x3_winograd(x.cuda(), layer, n_img_dev=cnt)
    y = F.conv2d(x.permute(0, 3, 1, 2), wt, padding=1)
    ref = F.relu(F.batch_norm(y, bn['running_mean'], bn['running_var'], bn['weight'], bn['bias'], False, 0.0, 1e-5))
    d = (got[:23].cpu().permute(0, 3, 1, 2) - ref[:23]).abs().max().item()
    assert d <= 1e-4 * ref.abs().max().item(), d


def test_full_size_linearity_and_guidance_equivalence():
    """Size-independent properties at the cfg3 AG-RPN size (3 guided maps of 50x84x1024 -> 1024; no CPU reference
    at this size: 238 GFLOP): (1) the layer is linear before its ReLU: conv(x1 + x2) = conv(x1) + conv(x2);
    (2) the guidance multiply fused into the input transform equals scaling the input first; (3) Winograd and the
    direct kernel agree.  Tolerance 1e-4 of the output range (fp32)."""
    from fgn_amd import ops
    g = torch.Generator().manual_seed(11)
    cin = cout = 1024
    x1 = torch.randn(1, 50, 84, cin, generator=g).cuda()
    x2 = torch.randn(1, 50, 84, cin, generator=g).cuda()
    wt = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    s = (torch.rand(3, cin, generator=g) + 0.5).cuda()
    lin = ops.pack_winograd(wt, relu=False).to('cuda')
    y1 = ops.conv3x3_winograd(x1, lin, in_scale=s, a_img_div=3)
    y2 = ops.conv3x3_winograd(x2, lin, in_scale=s, a_img_div=3)
    y12 = ops.conv3x3_winograd(x1 + x2, lin, in_scale=s, a_img_div=3)
    rng = y12.abs().max().item()
    assert y12.shape == (3, 50, 84, cout)
    assert (y12 - (y1 + y2)).abs().max().item() <= 1e-4 * rng
    pre = ops.conv3x3_winograd(ops.scale_channels(x1, s, 3), lin)
    assert (pre - y1).abs().max().item() <= 1e-4 * rng
    direct = ops.conv2d(ops.scale_channels(x1, s, 3), ops.pack_conv(wt, pad=1).to('cuda'))
    assert (direct - y1).abs().max().item() <= 1e-4 * rng


def test_two_tensor_winograd_launches_match_the_single_tensor_form():
    """``ops.conv3x3_winograd_multi``: the query map and the support maps of a backbone layer through ONE input
    transform launch, ONE grouped GEMM and ONE output transform launch (consecutive tile ranges of V / Mo) against
    ``conv3x3_winograd`` on each tensor alone, F(4x4) (two-tensor kernels) and F(2x2) (per-tensor launches): the same
    products summed by whichever GEMM kernel the row count selects (persistent 16x16x4 / one-shot 32x32x2 MFMA: the
    K order differs), so equal to a few ulp, not always bit for bit."""
    from fgn_amd import ops
    g = torch.Generator().manual_seed(9)
    for m in (4, 2):
        for (q_shape, s_shape, cin, cout) in (((1, 50, 84), (9, 16, 16), 256, 256), ((2, 13, 10), (3, 7, 9), 64, 128),
                                              ((1, 100, 167), (9, 32, 32), 128, 128)):
            wt = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
            layer = ops.pack_winograd(wt, bias=torch.randn(cout, generator=g), relu=True, m=m).to('cuda')
            xq = torch.randn(*q_shape, cin, generator=g).cuda()
            xs = torch.randn(*s_shape, cin, generator=g).cuda()
            buf = torch.full((xq[..., 0].numel() + xs[..., 0].numel(), cout), -7.0, device='cuda')
            yq = buf[:xq[..., 0].numel()].view(*q_shape, cout)
            ys = buf[xq[..., 0].numel():].view(*s_shape, cout)
            ops.conv3x3_winograd_multi([xq, xs], layer, [yq, ys])
            for got, x in ((yq, xq), (ys, xs)):
                want = ops.conv3x3_winograd(x, layer)
                assert float((got - want).abs().max()) <= 2e-6 * float(want.abs().max()), (m, tuple(x.shape))
            assert float(buf.min()) >= 0.0                        # every row of the shared buffer was written (ReLU output)
