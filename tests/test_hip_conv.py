"""MFMA implicit-GEMM conv (fgn_conv2d_nhwc_f32) vs a plain PyTorch fp32 reference.
Tolerance: fp32 accumulation-order differences only (rel 2e-5 of the output scale)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref(x_nchw, w, bias, bn, stride, pad, residual, relu, in_scale=None, a_img_div=1):
    x = x_nchw.double()
    if a_img_div > 1:
        x = x.repeat_interleave(a_img_div, dim=0)
    if in_scale is not None:
        x = x * in_scale.double()[:, :, None, None]
    y = F.conv2d(x, w.double(), None if bias is None else bias.double(), stride=stride, padding=pad)
    if bn is not None:
        y = F.batch_norm(y, bn['running_mean'].double(), bn['running_var'].double(), bn['weight'].double(),
                         bn['bias'].double(), False, 0.0, 1e-5)
    if residual is not None:
        y = y + residual.double()
    if relu:
        y = F.relu(y)
    return y.float()


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


CASES = [
    # n, cin, h, w, cout, k, stride, pad, bn, bias, residual, relu
    (2, 64, 17, 23, 64, 1, 1, 0, True, False, False, True),
    (2, 64, 17, 23, 256, 1, 1, 0, True, False, True, True),
    (1, 128, 20, 31, 128, 3, 2, 1, True, False, False, True),
    (3, 256, 7, 7, 128, 3, 1, 1, True, False, False, True),
    (1, 256, 9, 13, 75, 1, 1, 0, False, True, False, False),
    (2, 512, 8, 8, 1024, 1, 2, 0, True, False, False, False),
    (1, 32, 40, 50, 32, 3, 1, 1, False, True, False, True),
]


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('tile', [0, 1, 2, 3, 4, -4])
def test_conv_matches_torch(case, tile):
    from fgn_amd import ops
    n, cin, h, w, cout, k, stride, pad, use_bn, use_bias, use_res, relu = case
    g = torch.Generator().manual_seed(hash(case) % (2 ** 31))
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    bias = torch.randn(cout, generator=g) if use_bias else None
    bn = None
    if use_bn:
        bn = dict(weight=torch.rand(cout, generator=g) + 0.5, bias=torch.randn(cout, generator=g) * 0.1,
                  running_mean=torch.randn(cout, generator=g) * 0.1, running_var=torch.rand(cout, generator=g) + 0.5)
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    res = torch.randn(n, cout, ho, wo, generator=g) if use_res else None
    ref = _ref(x, wt, bias, bn, stride, pad, res, relu)
    layer = ops.pack_conv(wt, bias=bias, bn=bn, stride=stride, pad=pad, relu=relu).to('cuda')
    y = ops.conv2d(_nhwc(x).cuda(), layer, residual=None if res is None else _nhwc(res).cuda(), tile_hint=tile)
    torch.cuda.synchronize()
    got = y.cpu().permute(0, 3, 1, 2)
    scale = ref.abs().max().item() + 1e-6
    assert (got - ref).abs().max().item() <= 2e-5 * scale + 1e-6


def test_stem_conv_cin4():
    from fgn_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, 37, 53, generator=g)
    wt = torch.randn(64, 3, 7, 7, generator=g) / 12.0
    bn = dict(weight=torch.rand(64, generator=g) + 0.5, bias=torch.randn(64, generator=g) * 0.1,
              running_mean=torch.randn(64, generator=g) * 0.1, running_var=torch.rand(64, generator=g) + 0.5)
    ref = _ref(x, wt, None, bn, 2, 3, None, True)
    layer = ops.pack_conv(wt, bn=bn, stride=2, pad=3, relu=True, pad_cin_to=4).to('cuda')
    x4 = ops.nchw3_to_nhwc4(x.cuda())
    assert torch.equal(x4[..., :3].cpu(), _nhwc(x)) and float(x4[..., 3].abs().max()) == 0.0
    for tile in (0, 1, 3):
        y = ops.conv2d(x4, layer, tile_hint=tile).cpu().permute(0, 3, 1, 2)
        assert (y - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()


def test_input_scale_and_image_div_and_device_count():
    """AG-RPN style launch: N guided passes share one input map; per-(image,cin) scale; and a
    device-side image count that cuts the launch short."""
    from fgn_amd import ops
    g = torch.Generator().manual_seed(5)
    b, n_ways, cin, h, w, cout = 2, 3, 64, 9, 11, 96
    x = torch.randn(b, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / 24.0
    bias = torch.randn(cout, generator=g)
    vec = torch.rand(b * n_ways, cin, generator=g) + 0.5
    ref = _ref(x, wt, bias, None, 1, 1, None, True, in_scale=vec, a_img_div=n_ways)
    layer = ops.pack_conv(wt, bias=bias, pad=1, relu=True).to('cuda')
    y = ops.conv2d(_nhwc(x).cuda(), layer, in_scale=vec.cuda(), a_img_div=n_ways)
    got = y.cpu().permute(0, 3, 1, 2)
    assert (got - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    # device count: only the first 4 of 6 output images are produced
    out = torch.full((b * n_ways, h, w, cout), -7.0, device='cuda')
    cnt = torch.tensor([4], dtype=torch.int32, device='cuda')
    ops.conv2d(_nhwc(x).cuda(), layer, in_scale=vec.cuda(), a_img_div=n_ways, n_img_dev=cnt, out=out)
    got = out.cpu().permute(0, 3, 1, 2)
    assert (got[:4] - ref[:4]).abs().max().item() <= 2e-5 * ref.abs().max().item()
    assert float((got[4:] + 7.0).abs().max()) == 0.0


def test_bad_shapes_are_refused():
    from fgn_amd import ops
    from fgn_amd.lib import FgnHipError
    layer = ops.pack_conv(torch.randn(8, 24, 1, 1)).to('cuda')      # Cin=24: not a multiple of 32
    with pytest.raises(FgnHipError):
        ops.conv2d(torch.randn(1, 4, 4, 24, device='cuda'), layer)
    with pytest.raises(FgnHipError):
        ops.conv2d(torch.randn(1, 4, 4, 24), layer)                    # CPU tensor: no fallback


@pytest.mark.parametrize('shape', [(40, 7, 7, 256, 384, 3), (2, 30, 41, 128, 256, 1), (1, 33, 47, 64, 128, 3)])
def test_split_k_is_reproducible_and_honours_the_device_count(shape):
    """Split-K (K split over blockIdx.y + fixed-order reduce): same result as the unsplit kernel, bit-identical run
    to run, device count honoured."""
    from fgn_amd import ops
    n, h, w, cin, cout, k = shape
    g = torch.Generator().manual_seed(n * 7 + k)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    bias = torch.randn(cout, generator=g)
    res = torch.randn(n, cout, h, w, generator=g)
    ref = _ref(x, wt, bias, None, 1, k // 2, res, True)
    layer = ops.pack_conv(wt, bias=bias, pad=k // 2, relu=True).to('cuda')
    xd, rd = _nhwc(x).cuda(), _nhwc(res).cuda()
    y1 = ops.conv2d(xd, layer, residual=rd)
    y2 = ops.conv2d(xd, layer, residual=rd)
    assert torch.equal(y1, y2)                                   # fixed summation order
    got = y1.cpu().permute(0, 3, 1, 2)
    assert (got - ref).abs().max().item() <= 2e-5 * ref.abs().max().item() + 1e-6
    if n > 2:
        out = torch.full_like(y1, -3.0)
        cnt = torch.tensor([n - 3], dtype=torch.int32, device='cuda')
        ops.conv2d(xd, layer, residual=rd, n_img_dev=cnt, out=out)
        got = out.cpu().permute(0, 3, 1, 2)
        assert (got[:n - 3] - ref[:n - 3]).abs().max().item() <= 2e-5 * ref.abs().max().item() + 1e-6
        assert float((got[n - 3:] + 3.0).abs().max()) == 0.0


def test_repeated_runs_are_bit_identical_at_full_occupancy():
    """Regression: a 1x1 conv with 1840 workgroups of the 64x64 kernel (4-5 per CU) gave, about once in
    40 launches, one wave a stale last k-slice of a K-tile: the loop barrier was signalled while that wave's
    LDS reads were still queued and the next tile's LDS-DMA overtook them.  200 launches must agree bitwise,
    and with a torch reference."""
    from fgn_amd import ops
    g = torch.Generator().manual_seed(0)
    x = torch.randn(300, 7, 7, 1024, generator=g)
    wt = torch.randn(512, 1024, 1, 1, generator=g) * 0.05
    b = torch.randn(512, generator=g)
    layer = ops.pack_conv(wt, bias=b, relu=True).to('cuda')
    xc = x.cuda()
    ref = torch.relu(x.reshape(-1, 1024).double() @ wt.reshape(512, 1024).double().T + b.double()).float()
    first = ops.conv2d(xc, layer).clone()
    assert (first.reshape(-1, 512).cpu() - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()
    for i in range(200):
        assert torch.equal(ops.conv2d(xc, layer), first), f'launch {i} differs'


@pytest.mark.parametrize('rows,cin,cout,res,relu', [(49 * 300, 512, 1024, True, True), (103664, 64, 256, True, True),
                                                    (26016, 128, 512, False, True), (49 * 300 + 17, 1024, 1024, False, False),
                                                    (4200 * 5, 96, 260, True, False)])
@pytest.mark.parametrize('math', ['f32', 'x3', 'h2'])
def test_persistent_pointwise_kernel_matches_fp64_and_the_one_tile_kernel(rows, cin, cout, res, relu, math):
    """conv_pw_persist_kernel (the dominant kernel of an episode: 1x1 / stride 1 launches with more 64x64 output tiles
    than its 1024 persistent workgroups) at the episode's own shapes and at ragged ones (a last row tile of 17 rows, a
    channel count that is not a multiple of 64), with residual / BN / ReLU epilogues: against fp64, and against the
    one-tile-per-workgroup kernel of the same tile on pieces of the launch (fewer than 1024 tiles each, split-K off):
    the two differ only in the order of the K sum inside a 16-deep step (16x16x4 vs 32x32x2 MFMA: 2e-6 of the range),
    and every output element must be covered exactly once (a tile walked twice or skipped shows at once).
    math = 'x3': the same launches on conv_pw_x3_kernel (six bf16 MFMA products per f32 product, the build's default) -
    the same bound against fp64, and within 4e-6 of the range of the f32 one-tile kernel's pieces; 'h2': on
    conv_pw_h2_kernel (three f16 products; the default)."""
    from fgn_amd import lib, ops
    g = torch.Generator().manual_seed(rows + cin)
    x = torch.randn(rows, cin, generator=g)
    wt = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
    bn = dict(weight=torch.rand(cout, generator=g) + 0.5, bias=torch.randn(cout, generator=g) * 0.1,
              running_mean=torch.randn(cout, generator=g) * 0.1, running_var=torch.rand(cout, generator=g) + 0.5)
    r = torch.randn(rows, cout, generator=g) if res else None
    with ops.gemm_math(math):
        layer = ops.pack_conv(wt, bn=bn, relu=relu).to('cuda')
        assert (layer.w3 is not None) == ops._x3_ok(cin, cout)         # (Cout 260: 68 % of its 128-column tiles -> f32)
        assert (layer.wh is not None) == ops._h2_ok(cin, cout)
    xc = x.cuda().view(1, rows, 1, cin)
    rc = None if r is None else r.cuda().view(1, rows, 1, cout)
    L = lib.load()
    assert L.fgn_conv2d_kernel_id(1, rows, 1, cin, cout, layer.cout_pad, 1, 1, 1, 0, 1, 0, int(res), 0) % 10 == 4
    y = ops.conv2d(xc, layer, residual=rc)
    sc = bn['weight'].double() / torch.sqrt(bn['running_var'].double() + 1e-5)
    ref = (x.double() @ wt.reshape(cout, cin).double().T) * sc + (bn['bias'].double() - bn['running_mean'].double() * sc)
    if r is not None:
        ref = ref + r.double()
    if relu:
        ref = torch.relu(ref)
    got = y.view(rows, cout).cpu().double()
    assert (got - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    # pieces of <= 60 row tiles x cout/64 channel tiles < 1024 tiles: conv_igemm_dma_kernel<64,64,...,1>
    step = max(64, (1000 // ((cout + 63) // 64)) * 64)
    for m0 in range(0, rows, step):
        m1 = min(rows, m0 + step)
        piece = ops.conv2d(xc[:, m0:m1].contiguous(), layer, residual=None if rc is None else rc[:, m0:m1].contiguous(),
                           tile_hint=-4)
        assert L.fgn_conv2d_kernel_id(1, m1 - m0, 1, cin, cout, layer.cout_pad, 1, 1, 1, 0, 1, 0, int(res), -4) % 10 == 1
        assert (piece - y[:, m0:m1]).abs().max().item() <= (2e-6 if math == 'f32' else 4e-6) * ref.abs().max().item(), (m0, m1)


@pytest.mark.parametrize('cin,cout,k,stride', [(128, 128, 3, 2), (256, 512, 1, 2), (64, 64, 3, 1), (4, 64, 7, 2)])
@pytest.mark.parametrize('math', ['f32', 'h2'])
def test_two_tensor_launch_equals_one_launch_per_tensor(cin, cout, k, stride, math):
    """``ops.conv2d_pair``: the query map and the support maps of a strided backbone layer in ONE launch.  Per tensor it
    is the arithmetic of ``conv2d`` without split-K: identical bytes where the single launch is not split, within
    rounding of the K order where it is (the small support half alone is split to fill the chip).
    math = 'h2' (the default): the 3x3 / strided layers as implicit GEMMs on conv_pw_h2_kernel (fgn_conv2d_pair_h2_nhwc_f32;
    the stem, Cin 4, stays on the f32 kernels): against fp64 like every conv kernel, the two-tensor launch within the
    accumulation's rounding of one launch per tensor (a wave's rows, and so its scale, differ between the two), and the
    query half alone falls back to nothing - the library's own kernel id confirms which kernels ran."""
    from fgn_amd import lib, ops
    g = torch.Generator().manual_seed(17)
    real_cin = 3 if cin == 4 else cin
    wt = torch.randn(cout, real_cin, k, k, generator=g) / (real_cin * k * k) ** 0.5
    bn = dict(weight=torch.rand(cout, generator=g) + 0.5, bias=torch.randn(cout, generator=g) * 0.1,
              running_mean=torch.randn(cout, generator=g) * 0.1, running_var=torch.rand(cout, generator=g) + 0.5)
    with ops.gemm_math(math):
        layer = ops.pack_conv(wt, bn=bn, stride=stride, pad=k // 2, relu=True, **({'pad_cin_to': 4} if cin == 4 else {})).to('cuda')
    h2 = math == 'h2' and cin != 4
    assert (layer.wh is not None) == h2
    xq = torch.randn(1, 201, 335, cin, generator=g).cuda()
    xs = torch.randn(9, 32, 32, cin, generator=g).cuda()
    if cin == 4:
        xq[..., 3] = 0
        xs[..., 3] = 0
    yq, ys = ops.conv2d_pair(xq, xs, layer)
    L = lib.load()
    sc = bn['weight'].double() / torch.sqrt(bn['running_var'].double() + 1e-5)
    for x, y in ((xq, yq), (xs, ys)):
        ref = ops.conv2d(x, layer)
        assert y.shape == ref.shape
        if h2:
            r64 = F.conv2d(x.permute(0, 3, 1, 2).double().cpu(), wt.double(), stride=stride, padding=k // 2)
            r64 = torch.relu(r64 * sc[None, :, None, None] + (bn['bias'].double() - bn['running_mean'].double() * sc)[None, :, None, None])
            r64 = r64.permute(0, 2, 3, 1)
            rng = r64.abs().max().item()
            assert (y.cpu().double() - r64).abs().max().item() <= 2e-6 * rng
            assert (ref.cpu().double() - r64).abs().max().item() <= 2e-6 * rng
            assert (y - ref).abs().max().item() <= 2.5e-6 * rng      # (a small single launch runs on the f32 kernels)
            continue
        split = L.fgn_conv2d_workspace_bytes(x.shape[0], x.shape[1], x.shape[2], cin, cout, k, k, stride, k // 2, 0) > 0
        if split:
            assert (y - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()
        else:
            assert torch.equal(y, ref)
    if h2:      # the launch shapes of this test are large enough for the h2 tile rule (else the pair fell back to the f32 kernel)
        rows = yq.shape[0] * yq.shape[1] * yq.shape[2] + ys.shape[0] * ys.shape[1] * ys.shape[2]
        assert L.fgn_h2_row_tile(rows, cout, k * k * cin, 0, 0) > 0
    # outputs into caller-provided buffers (views of one allocation, as the backbone uses them)
    buf = torch.empty(yq.numel() + ys.numel(), device='cuda')
    oq, os_ = buf[:yq.numel()].view(yq.shape), buf[yq.numel():].view(ys.shape)
    ops.conv2d_pair(xq, xs, layer, oq, os_)
    assert torch.equal(oq, yq) and torch.equal(os_, ys)
    # the two inputs as views of one allocation (how the backbone holds them: the implicit-GEMM form addresses both through
    # one buffer descriptor and leaves inputs more than 2 GiB apart - separate allocations can be - to the f32 kernels), the
    # query map first and the support maps first: identical bytes, and the fp64 bound again
    res = []
    for first in ('q', 's'):
        xb = torch.empty(xq.numel() + xs.numel(), device='cuda')
        if first == 'q':
            xq2, xs2 = xb[:xq.numel()].view(xq.shape), xb[xq.numel():].view(xs.shape)
        else:
            xs2, xq2 = xb[:xs.numel()].view(xs.shape), xb[xs.numel():].view(xq.shape)
        xs2.copy_(xs); xq2.copy_(xq)
        res.append(ops.conv2d_pair(xq2, xs2, layer))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    tol = 2.5e-6 if h2 else 2e-6
    assert (res[0][0] - yq).abs().max().item() <= tol * yq.abs().max().item()
    assert (res[0][1] - ys).abs().max().item() <= tol * ys.abs().max().item()


@pytest.mark.parametrize('rows,cin1,cin2,cout', [(103664, 64, 64, 256), (5000, 32, 96, 132), (777, 128, 32, 64)])
@pytest.mark.parametrize('math', ['f32', 'x3', 'h2'])
def test_dual_operand_pointwise_conv_is_conv3_plus_shortcut(rows, cin1, cin2, cout, math):
    """``conv1x1_dual`` (fgn_conv1x1_dual_nhwc_f32): relu(bn3(conv3(y)) + bn_d(conv_d(x))) of the first Bottleneck of a
    stride-1 stage as ONE K loop over [y | x] with the BatchNorm scales folded into the weights - against fp64, and
    against the two-launch form it replaces (shortcut conv, then conv3 with the residual in its epilogue), at the cfg3
    layer1.0 shape and at ragged ones (fewer tiles than persistent workgroups, Cout not a multiple of 64)."""
    from fgn_amd import ops
    g = torch.Generator().manual_seed(rows)
    y, x = torch.randn(rows, cin1, generator=g), torch.randn(rows, cin2, generator=g)
    w3, wd = torch.randn(cout, cin1, 1, 1, generator=g) / cin1 ** 0.5, torch.randn(cout, cin2, 1, 1, generator=g) / cin2 ** 0.5
    mk = lambda: dict(weight=torch.rand(cout, generator=g) + 0.5, bias=torch.randn(cout, generator=g) * 0.1,
                      running_mean=torch.randn(cout, generator=g) * 0.1, running_var=torch.rand(cout, generator=g) + 0.5)
    bn3, bnd = mk(), mk()

    def affine(v, bn):
        sc = bn['weight'].double() / torch.sqrt(bn['running_var'].double() + 1e-5)
        return v * sc + (bn['bias'].double() - bn['running_mean'].double() * sc)
    ref = torch.relu(affine(y.double() @ w3.reshape(cout, cin1).double().T, bn3) +
                     affine(x.double() @ wd.reshape(cout, cin2).double().T, bnd))
    with ops.gemm_math(math):
        layer = ops.pack_conv_dual(w3, bn3, wd, bnd, relu=True).to('cuda')
        assert (layer.w3 is not None) == ops._x3_ok(cin1 + cin2, cout)
    yc, xc = y.cuda().view(1, rows, 1, cin1), x.cuda().view(1, rows, 1, cin2)
    got = ops.conv1x1_dual(yc, xc, layer)
    assert tuple(got.shape) == (1, rows, 1, cout)
    assert (got.view(rows, cout).cpu().double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    with ops.gemm_math('f32'):
        two = ops.conv2d(yc, ops.pack_conv(w3, bn=bn3, relu=True).to('cuda'),
                         residual=ops.conv2d(xc, ops.pack_conv(wd, bn=bnd).to('cuda')))
    assert (got - two).abs().max().item() <= 4e-6 * ref.abs().max().item()
    again = ops.conv1x1_dual(yc, xc, layer)
    assert torch.equal(got, again)


@pytest.mark.parametrize('math', ['f32', 'x3'])
def test_dual_operand_conv_with_a_strided_shortcut(math):
    """The 1x1 / stride 2 shortcut of layer2.0 / layer3.0 inside conv3's K loop: output row m reads row x2_rows[m] of the
    stage's input (``ops.strided_rows`` over the query map and the support maps lying one behind the other) - against the
    two-launch form (strided shortcut conv, then conv3 with the residual in its epilogue) and fp64."""
    from fgn_amd import ops
    g = torch.Generator().manual_seed(5)
    cin1, cin2, cout = 64, 96, 128
    q, s_ = torch.randn(1, 21, 34, cin2, generator=g), torch.randn(3, 10, 10, cin2, generator=g)
    buf = torch.cat([q.reshape(-1, cin2), s_.reshape(-1, cin2)]).cuda().contiguous()
    rows_tab = ops.strided_rows([(1, 21, 34), (3, 10, 10)], 2, 'cuda')
    rows = rows_tab.numel()
    assert rows == 11 * 17 + 3 * 5 * 5
    y = torch.randn(rows, cin1, generator=g)
    w3, wd = torch.randn(cout, cin1, 1, 1, generator=g) / 8, torch.randn(cout, cin2, 1, 1, generator=g) / 10
    mk = lambda: dict(weight=torch.rand(cout, generator=g) + 0.5, bias=torch.randn(cout, generator=g) * 0.1,
                      running_mean=torch.randn(cout, generator=g) * 0.1, running_var=torch.rand(cout, generator=g) + 0.5)
    bn3, bnd = mk(), mk()
    with ops.gemm_math(math):
        layer = ops.pack_conv_dual(w3, bn3, wd, bnd, relu=True).to('cuda')
    yc = y.cuda().view(1, rows, 1, cin1)
    got = ops.conv1x1_dual(yc, buf.view(1, -1, 1, cin2), layer, x2_rows=rows_tab)
    with ops.gemm_math('f32'):
        down = ops.pack_conv(wd, bn=bnd, stride=2).to('cuda')
        idt = torch.cat([ops.conv2d(q.cuda(), down).reshape(-1, cout), ops.conv2d(s_.cuda(), down).reshape(-1, cout)])
        two = ops.conv2d(yc, ops.pack_conv(w3, bn=bn3, relu=True).to('cuda'), residual=idt.view(1, rows, 1, cout))
    assert (got - two).abs().max().item() <= 4e-6 * two.abs().max().item()
    with pytest.raises(Exception):
        ops.conv1x1_dual(yc, buf.view(1, -1, 1, cin2), layer)                 # rows differ and no table


@pytest.mark.parametrize('groups,grp_rows,valid,K,N', [(1, 14700, 14700, 1024, 1024), (1, 3001, 3001, 64, 76), (1, 130, 130, 96, 260),
                                                       (36, 256, 201, 128, 132), (36, 256, 100, 64, 128), (16, 128, 128, 256, 512), (1, 70000, 70000, 64, 256)])
def test_x3_gemm_is_as_close_to_fp64_as_the_f32_mfma_kernel(groups, grp_rows, valid, K, N):
    """conv_pw_x3_kernel through its direct entry (fgn_gemm_x3_f32): f32 operands, every product as six (nterms 9: nine)
    bf16 MFMA products of exact three-way splits, f32 accumulation.  Against fp64 on post-ReLU activations x random
    weights: within 2e-6 of the range like the f32 MFMA kernels, and no further from fp64 than 1.6x the f32 MFMA kernel on
    the same operands (measured r05: 1.2x, for six terms and for nine alike - the difference is the accumulation order
    inside the matrix pipe, not the dropped terms).  Ragged rows / channels, grouped launches with fewer valid rows than a
    group holds, K of 2 and 3 K-tiles; the 64- and the 128-row tile give the same bits.  With FGN_HIP_LIB = the experiments
    build (tools/micro/build_experiments.sh) the 32x32x16 MFMA instances run too (their own bits, their own weight image);
    the product library refuses them."""
    from fgn_amd import lib, ops
    g = torch.Generator().manual_seed(groups * 1000 + K + N)
    x = torch.randn(groups, grp_rows, K, generator=g).relu_().cuda()
    w = (torch.randn(groups, N, K, generator=g) / K ** 0.5).cuda()
    shift = torch.randn(N, generator=g).cuda()
    ref = torch.einsum('grk,gnk->grn', x[:, :valid].double(), w.double()) + shift.double()
    rng = ref.abs().max().item()
    img16, img32 = ops.pack_x3(w), ops.pack_x3(w, mfma32=True)
    outs = {}
    for bm in (64, 128, 2064, 2128, 2129):          # + 2000: the 32x32x16 MFMA form (experiments build); 129: 8 waves x 3 stages
        img = img32 if bm >= 2000 else img16
        tile = min(bm % 1000, 128)
        if groups > 1 and grp_rows % tile:
            continue
        for nt in (6, 9):
            out = torch.full((groups, grp_rows, N), float('nan'), device='cuda')
            try:
                ops.gemm_x3(x, img, N, shift=shift, groups=groups, grp_valid=valid, bm=bm, nterms=nt, out=out)
            except lib.FgnHipError:
                # the 32x32x16 instances live in the experiments build only; nine terms exist for the 64-row tile
                assert bm >= 2000 or (bm == 128 and nt == 9)
                continue
            outs[(bm, nt)] = out
            err = (out[:, :valid].double() - ref).abs().max().item()
            assert err <= 2e-6 * rng, (bm, nt, err / rng)
            # rows of whole tiles past the last valid one of a group are not written
            last = -(-valid // tile) * tile
            assert torch.isnan(out[:, last:]).all()
    if (128, 6) in outs:                            # one MFMA shape: the same bits whatever the row tile
        assert torch.equal(outs[(64, 6)][:, :valid], outs[(128, 6)][:, :valid])
    for bm in (2128, 2129):
        if (bm, 6) in outs:
            assert torch.equal(outs[(2064, 6)][:, :valid], outs[(bm, 6)][:, :valid])
    # the f32 MFMA kernel on the same operands
    if groups == 1:
        with ops.gemm_math('f32'):
            layer = ops.pack_conv(w[0].reshape(N, K, 1, 1), bias=shift).to('cuda')
        f32 = ops.conv2d(x.view(1, grp_rows, 1, K), layer).view(1, grp_rows, N)
    else:
        L = lib.load()
        cout_pad = (N + 127) // 128 * 128
        u = torch.zeros(groups, cout_pad, K, device='cuda')
        u[:, :N] = w
        f32 = torch.zeros(groups, grp_rows, N, device='cuda')
        lib.check(L.fgn_winograd_gemm_f32(x.data_ptr(), u.data_ptr(), f32.data_ptr(), None, 1, valid, grp_rows, K, N, cout_pad,
                                          groups, torch.cuda.current_stream().cuda_stream), 'wg')
        f32 = f32 + shift
    e32 = (f32[:, :valid].double() - ref).abs()
    e6 = (outs[(64, 6)][:, :valid].double() - ref).abs()
    assert e6.max().item() <= 1.6 * e32.max().item() + 1e-7 * rng
    assert e6.mean().item() <= 1.6 * e32.mean().item() + 1e-8 * rng


def test_x3_gemm_epilogue_and_special_values():
    """Residual + ReLU in the epilogue; operands with zeros, tiny and huge magnitudes (the split is exact from 2^-100 up:
    below that the third plane underflows, far under any activation); rows past M untouched.  (bm = 64: the automatic
    choice leaves a launch this small, half of whose 128 columns are padding, to the f32 kernels.)"""
    from fgn_amd import lib
    assert lib.load().fgn_x3_row_tile(1000, 64, 128, 0, 0) == 0 and lib.load().fgn_x3_row_tile(14700, 1024, 1024, 0, 0) == 128
    from fgn_amd import ops
    g = torch.Generator().manual_seed(3)
    rows, K, N = 1000, 128, 64
    x = torch.randn(rows, K, generator=g)
    x[::7] = 0.0
    x[1::7] *= 1e-20
    x[2::7] *= 1e15
    w = torch.randn(N, K, generator=g) / K ** 0.5
    w[:, ::5] *= 1e-12
    res = torch.randn(rows, N, generator=g)
    ref = torch.relu(x.double() @ w.double().T + res.double())
    out = torch.full((rows + 64, N), -7.0, device='cuda')
    ops.gemm_x3(x.cuda(), ops.pack_x3(w.cuda()), N, residual=res.cuda(), relu=True, bm=64, out=out[:rows])
    got = out[:rows].cpu().double()
    assert torch.isfinite(got).all() and (out[rows:] == -7.0).all()
    # per row: relative to that row's own scale (rows differ by 35 orders of magnitude)
    scale = (x.double().abs() @ w.double().abs().T).max(1, keepdim=True).values + res.double().abs()
    assert ((got - ref).abs() / scale).max().item() <= 2e-6


_H2_SHAPES = [(1, 14700, 14700, 1024, 1024), (1, 3001, 3001, 64, 76), (1, 130, 130, 96, 260), (36, 256, 201, 128, 132),
              (36, 256, 100, 64, 128), (16, 128, 128, 256, 512), (1, 70000, 70000, 64, 256), (1, 40000, 40000, 256, 64),
              (1, 33000, 33000, 576, 52)]
# every shape at magnitude 1; magnitudes below / above the f16 range on four of them
_H2_CASES = [sh + (1.0,) for sh in _H2_SHAPES] + \
    [sh + (mag,) for sh in _H2_SHAPES if (sh[0], sh[3], sh[4]) in ((1, 1024, 1024), (36, 128, 132), (1, 256, 64), (1, 96, 260)) for mag in (3e-5, 2e7)]


@pytest.mark.parametrize('groups,grp_rows,valid,K,N,mag', _H2_CASES)
def test_h2_gemm_is_as_close_to_fp64_as_the_f32_mfma_kernel(groups, grp_rows, valid, K, N, mag):
    """conv_pw_h2_kernel through its direct entry (fgn_gemm_h2_f32): f32 operands, every product as three f16 MFMA products of
    two-way splits of the power-of-two scaled operands (weights per output column at pack time, activations by the scale
    the kernel finds per wave and output tile), f32 accumulation.  Against fp64 on post-ReLU activations x random weights,
    with the activations at magnitudes inside, below and above the f16 range: within 2e-6 of the range like the f32 MFMA
    kernels and conv_pw_x3_kernel, and no further from fp64 than 1.6x the f32 MFMA kernel on the same operands (measured:
    closer than conv_pw_x3_kernel).  Ragged rows / channels, grouped launches with fewer valid rows than a group holds, K of
    2 and 3 K-tiles.  The scales differ per wave and tile, so the tiles may differ where an element's l plane
    reaches the f16 subnormals: within 1e-7 of the range of each other."""
    from fgn_amd import lib, ops
    g = torch.Generator().manual_seed(groups * 1000 + K + N)
    x = (torch.randn(groups, grp_rows, K, generator=g).relu_() * mag).cuda()
    x[:, valid:] = 3e38                                  # what lies behind a group's valid rows must not reach a wave's scale
    w = (torch.randn(groups, N, K, generator=g) / K ** 0.5).cuda()
    w[:, 1::3] *= 40.0                                   # columns of different magnitude: the column scales differ
    shift = (torch.randn(N, generator=g) * mag).cuda()
    ref = torch.einsum('grk,gnk->grn', x[:, :valid].double(), w.double()) + shift.double()
    rng = ref.abs().max().item()
    img = ops.pack_h2(w)
    outs = {}
    assert lib.load().fgn_h2_row_tile(groups * grp_rows, N, K, grp_rows if groups > 1 else 0, valid if groups > 1 else 0) == \
        {1024: 64, 76: 0, 260: 0, 132: 0, 128: 0, 512: 0, 256: 64, 64: 264, 52: 264}[N]
    for bm in (64, 128, 264):                          # 264: 128 rows x 64 columns (the tile of layers with <= 64 channels)
        tile = 64 if bm == 64 else 128
        if groups > 1 and grp_rows % tile:
            continue
        out = torch.full((groups, grp_rows, N), float('nan'), device='cuda')
        ops.gemm_h2(x, img, N, shift=shift, groups=groups, grp_valid=valid, bm=bm, out=out)
        outs[bm] = out
        err = (out[:, :valid].double() - ref).abs().max().item()
        assert err <= 2e-6 * rng, (bm, err / rng)
        last = -(-valid // tile) * tile                  # rows of whole tiles past the last valid one of a group are not written
        assert torch.isnan(out[:, last:]).all()
    for bm in (128, 264):
        if bm in outs:
            assert (outs[64][:, :valid] - outs[bm][:, :valid]).abs().max().item() <= 1e-7 * rng
    if groups == 1:
        with ops.gemm_math('f32'):
            layer = ops.pack_conv(w[0].reshape(N, K, 1, 1), bias=shift).to('cuda')
        f32 = ops.conv2d(x.view(1, grp_rows, 1, K), layer).view(1, grp_rows, N)
    else:
        L = lib.load()
        cout_pad = (N + 127) // 128 * 128
        u = torch.zeros(groups, cout_pad, K, device='cuda')
        u[:, :N] = w
        f32 = torch.zeros(groups, grp_rows, N, device='cuda')
        lib.check(L.fgn_winograd_gemm_f32(x.data_ptr(), u.data_ptr(), f32.data_ptr(), None, 1, valid, grp_rows, K, N, cout_pad,
                                          groups, torch.cuda.current_stream().cuda_stream), 'wg')
        f32 = f32 + shift
    e32 = (f32[:, :valid].double() - ref).abs()
    for bm in (64, 264):
        if bm in outs:
            eh = (outs[bm][:, :valid].double() - ref).abs()
            assert eh.max().item() <= 1.6 * e32.max().item() + 1e-7 * rng
            assert eh.mean().item() <= 1.6 * e32.mean().item() + 1e-8 * rng


def test_h2_gemm_epilogue_and_dynamic_range():
    """Residual + ReLU in the epilogue; rows past M untouched; zero rows and rows of 1e-7 among rows of 1 (a wave's 32 rows
    share one scale: errors stay relative to the largest row); whole tiles of 1e-7 behind tiles of 1 (every wave scales its
    own rows: errors relative to the row itself); a K-tile 3e4 above / 1e4 below the ones before it (the wave picks a new
    scale and its accumulators follow / keeps the old one); an all-zero operand gives the epilogue of zero."""
    from fgn_amd import ops
    g = torch.Generator().manual_seed(4)
    rows, K, N = 1024, 128, 64
    x = torch.randn(rows, K, generator=g)
    x[::7] = 0.0
    x[1::7] *= 1e-7
    w = torch.randn(N, K, generator=g) / K ** 0.5
    w[:, ::5] *= 1e-6
    res = torch.randn(rows, N, generator=g)
    img = ops.pack_h2(w.cuda())
    for grow in (1.0, 3e4, 1e-4):
        x2 = x.clone()
        x2[:, 96:] *= grow
        ref = torch.relu(x2.double() @ w.double().T + res.double())
        out = torch.full((rows + 64, N), -7.0, device='cuda')
        ops.gemm_h2(x2.cuda(), img, N, residual=res.cuda(), relu=True, bm=64, out=out[:rows])
        got = out[:rows].cpu().double()
        assert torch.isfinite(got).all() and (out[rows:] == -7.0).all()
        row_scale = (x2.double().abs() @ w.double().abs().T).max(1, keepdim=True).values
        assert ((got - ref).abs() <= 2e-6 * (row_scale.max() + res.double().abs())).all(), grow
    x3 = x.clone()
    x3[512:] *= 1e-7                                   # rows 512.. : whole 64-row tiles seven orders of magnitude down
    x3[512::7] = x[512::7] * 1e-7 + 1e-9
    got = ops.gemm_h2(x3.cuda(), img, N, bm=64).cpu().double()
    ref = x3.double() @ w.double().T
    row_scale = (x3.double().abs() @ w.double().abs().T).max(1, keepdim=True).values
    blk = row_scale.view(-1, 32, 1).max(1, keepdim=True).values.expand(-1, 32, 1).reshape(rows, 1)     # a wave's 32 rows
    assert ((got - ref).abs() <= 2e-6 * blk).all()
    z = ops.gemm_h2(torch.zeros(256, K, device='cuda'), img, N, shift=res[0].cuda(), bm=64)
    assert torch.equal(z, res[0].cuda().expand(256, N))


def test_winograd_h2_is_as_close_to_fp64_as_x3():
    """F(4x4) layers under GEMM_MATH 'h2' (the default) against the bf16-plane GEMM and fp64: the grouped GEMM on
    conv_pw_h2_kernel adds nothing to F(4x4)'s own f32 error (~2.5e-6 of the range, csrc/winograd.hip); the input-scale
    (guidance) and device-count arguments pass through; two tensors through one launch (conv3x3_winograd_multi) agree with
    separate calls."""
    from fgn_amd import ops
    g = torch.Generator().manual_seed(11)
    cin, cout = 256, 192
    w = torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    x = (torch.randn(6, 29, 45, cin, generator=g).relu_() * 3.0).cuda()
    sc = (torch.rand(6, cin, generator=g) + 0.5).cuda()
    with ops.gemm_math('h2'):
        lh = ops.pack_winograd(w, bias=b, relu=True).to('cuda')
    with ops.gemm_math('x3'):
        l3 = ops.pack_winograd(w, bias=b, relu=True).to('cuda')
    assert lh.uh is not None and lh.u3 is None and l3.uh is None and l3.u3 is not None
    n_dev = torch.tensor([5], dtype=torch.int32, device='cuda')
    conv64 = lambda t: torch.nn.functional.conv2d(t.permute(0, 3, 1, 2).double(), w.cuda().double(), b.cuda().double(),
                                                  padding=1).relu_().permute(0, 2, 3, 1)
    for kw in (dict(), dict(in_scale=sc), dict(n_img_dev=n_dev)):
        yh, y3 = ops.conv3x3_winograd(x, lh, **kw), ops.conv3x3_winograd(x, l3, **kw)
        n = 5 if 'n_img_dev' in kw else 6
        ref = conv64(x[:n] * sc[:n, None, None, :] if 'in_scale' in kw else x[:n])
        rng = ref.abs().max().item()
        eh, e3 = (yh[:n].double() - ref).abs(), (y3[:n].double() - ref).abs()
        assert eh.max().item() <= 1.3 * e3.max().item() + 5e-7 * rng and eh.mean().item() <= 1.2 * e3.mean().item() + 2e-8 * rng
        assert eh.max().item() <= 1e-5 * rng
    x1 = torch.randn(3, 16, 16, cin, generator=g).relu_().cuda()
    o0, o1 = torch.empty(6, 29, 45, cout, device='cuda'), torch.empty(3, 16, 16, cout, device='cuda')
    ops.conv3x3_winograd_multi([x, x1], lh, [o0, o1])
    assert (o0 - ops.conv3x3_winograd(x, lh)).abs().max().item() <= 1e-5 * o0.abs().max().item()
    assert (o1 - ops.conv3x3_winograd(x1, lh)).abs().max().item() <= 1e-5 * o1.abs().max().item()
