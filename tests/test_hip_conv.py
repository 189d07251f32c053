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


@pytest.mark.parametrize('shape', [(1, 50, 84, 1024, 256, 1), (9, 7, 7, 1024, 512, 1), (9, 7, 7, 512, 512, 3),
                                   (9, 16, 16, 1024, 256, 1), (1, 25, 42, 1024, 256, 1)])
def test_split_k_reduce_inside_the_launch_equals_the_two_kernel_form(shape):
    """The workgroup that publishes the last partial tile of an output tile reduces it inside the conv launch
    (write-through slab stores, one ticket per tile, slabs summed in slab order): byte-identical to the separate reduce
    kernel, on every one of 300 back-to-back launches that reuse the same slabs and tickets (layer shapes of the cfg3
    episode that are split: layer3 1x1s on the query / support maps, the 9-RoI support shared head)."""
    from fgn_amd import ops
    n, h, w, cin, cout, k = shape
    g = torch.Generator().manual_seed(cin + cout + k)
    x = torch.randn(n, h, w, cin, generator=g).cuda()
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    res = torch.randn(n, h, w, cout, generator=g).cuda()
    layer = ops.pack_conv(wt, bias=torch.randn(cout, generator=g), pad=k // 2, relu=True).to('cuda')
    L = ops._lib.load()
    assert L.fgn_conv2d_splitk_tickets(n, h, w, cin, cout, k, k, 1, k // 2, 0) > 0, 'shape is not split: test is vacuous'
    keep = ops.SPLITK_IN_LAUNCH
    try:
        ops.SPLITK_IN_LAUNCH = False
        two = ops.conv2d(x, layer, residual=res).clone()
        ops.SPLITK_IN_LAUNCH = True
        for i in range(300):
            got = ops.conv2d(x, layer, residual=res)
            assert torch.equal(got, two), f'launch {i} differs from the two-kernel reduce'
    finally:
        ops.SPLITK_IN_LAUNCH = keep
    ref = torch.relu(torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double().cpu(), wt.double(),
                                                layer.shift.double().cpu(), padding=k // 2)
                     + res.permute(0, 3, 1, 2).double().cpu()).float()
    assert (two.cpu().permute(0, 3, 1, 2) - ref).abs().max().item() <= 2e-5 * ref.abs().max().item() + 1e-6


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


@pytest.mark.parametrize('cin,cout,k,stride', [(128, 128, 3, 2), (256, 512, 1, 2), (64, 64, 3, 1), (4, 64, 7, 2)])
def test_two_tensor_launch_equals_one_launch_per_tensor(cin, cout, k, stride):
    """``ops.conv2d_pair``: the query map and the support maps of a strided backbone layer in ONE launch.  Per tensor it
    is the arithmetic of ``conv2d`` without split-K: identical bytes where the single launch is not split, within
    rounding of the K order where it is (the small support half alone is split to fill the chip)."""
    from fgn_amd import lib, ops
    g = torch.Generator().manual_seed(17)
    real_cin = 3 if cin == 4 else cin
    wt = torch.randn(cout, real_cin, k, k, generator=g) / (real_cin * k * k) ** 0.5
    bn = dict(weight=torch.rand(cout, generator=g) + 0.5, bias=torch.randn(cout, generator=g) * 0.1,
              running_mean=torch.randn(cout, generator=g) * 0.1, running_var=torch.rand(cout, generator=g) + 0.5)
    layer = ops.pack_conv(wt, bn=bn, stride=stride, pad=k // 2, relu=True, **({'pad_cin_to': 4} if cin == 4 else {})).to('cuda')
    xq = torch.randn(1, 101, 167, cin, generator=g).cuda()
    xs = torch.randn(9, 32, 32, cin, generator=g).cuda()
    if cin == 4:
        xq[..., 3] = 0
        xs[..., 3] = 0
    yq, ys = ops.conv2d_pair(xq, xs, layer)
    L = lib.load()
    for x, y in ((xq, yq), (xs, ys)):
        ref = ops.conv2d(x, layer)
        assert y.shape == ref.shape
        split = L.fgn_conv2d_workspace_bytes(x.shape[0], x.shape[1], x.shape[2], cin, cout, k, k, stride, k // 2, 0) > 0
        if split:
            assert (y - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()
        else:
            assert torch.equal(y, ref)
    # outputs into caller-provided buffers (views of one allocation, as the backbone uses them)
    buf = torch.empty(yq.numel() + ys.numel(), device='cuda')
    oq, os_ = buf[:yq.numel()].view(yq.shape), buf[yq.numel():].view(ys.shape)
    ops.conv2d_pair(xq, xs, layer, oq, os_)
    assert torch.equal(oq, yq) and torch.equal(os_, ys)


# ------------------------------------------------------------------------------------------------
# conv_pw_persist2_kernel (round 4): every tile code through the tuning knob, against fp64 and against
# the round-3 kernels on the same operands
# ------------------------------------------------------------------------------------------------
class _pw2:
    """``with _pw2(code):`` forces tile code `code` of conv_pw_persist2_kernel for every eligible launch."""

    def __init__(self, code):
        self.code = code

    def __enter__(self):
        from fgn_amd import lib
        self.prev = lib.load().fgn_conv2d_tune(0, self.code)

    def __exit__(self, *a):
        from fgn_amd import lib
        lib.load().fgn_conv2d_tune(0, self.prev)


@pytest.mark.parametrize('code', [1, 2, 3, 4, 5, 6, 7, 8, 34, 44, 32, 42, 36, 47, 38, 48])   # tile + 10 * LDS stages
@pytest.mark.parametrize('rows,cin,cout,res,relu', [(1000, 64, 72, True, True), (49 * 37, 96, 256, False, True),
                                                     (130, 1024, 512, True, False), (64 * 9 + 1, 32, 4, False, False)])
def test_persist2_pointwise_matches_fp64_and_round3_kernel(code, rows, cin, cout, res, relu):
    """1x1 / stride 1 convolutions with BN epilogue (+ residual, ReLU): row counts that end inside a tile, Cout that
    ends inside a 16-channel MFMA tile and inside a workgroup tile, K of 1..32 K-tiles; a device-side image count."""
    from fgn_amd import ops
    g = torch.Generator().manual_seed(rows + cin + cout)
    x = torch.randn(rows, cin, 1, 1, generator=g)
    wt = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
    bn = dict(weight=torch.rand(cout, generator=g) + 0.5, bias=torch.randn(cout, generator=g) * 0.1,
              running_mean=torch.randn(cout, generator=g) * 0.1, running_var=torch.rand(cout, generator=g) + 0.5)
    r = torch.randn(rows, cout, 1, 1, generator=g) if res else None
    ref = _ref(x, wt, None, bn, 1, 0, r, relu)
    layer = ops.pack_conv(wt, bn=bn, relu=relu).to('cuda')
    xc, rc = _nhwc(x).cuda(), None if r is None else _nhwc(r).cuda()
    with _pw2(0):
        old = ops.conv2d(xc, layer, residual=rc).clone()
    with _pw2(code):
        got = ops.conv2d(xc, layer, residual=rc).clone()
        cnt = torch.tensor([rows - 77], dtype=torch.int32, device='cuda')
        part = torch.full((rows, 1, 1, cout), -7.0, device='cuda')
        ops.conv2d(xc, layer, residual=rc, n_img_dev=cnt, out=part)
    scale = ref.abs().max().item() + 1e-6
    assert (got.cpu().permute(0, 3, 1, 2) - ref).abs().max().item() <= 2e-5 * scale + 1e-6
    # same products in the same k order as the round-3 kernels; the epilogue may contract differently (one ulp)
    assert (got - old).abs().max().item() <= 2e-6 * scale
    assert torch.equal(part[:rows - 77], got[:rows - 77]) and float((part[rows - 77:] + 7.0).abs().max()) == 0.0


@pytest.mark.parametrize('code', [1, 2, 3, 4, 5, 6, 7, 8, 34, 44, 32, 42, 36, 47, 38, 48])   # tile + 10 * LDS stages
@pytest.mark.parametrize('n,tiles,cin,cout', [(3, 273, 64, 128), (5, 4, 512, 64), (100, 4, 32, 256)])
def test_persist2_grouped_gemm_all_tiles_and_the_tile_scheduler(code, n, tiles, cin, cout):
    """The grouped Winograd GEMM (36 groups, per-group weights): t_pad is a multiple of 64, so with 128-row tiles the
    last row tile of every group is cut at the group's end; rows past n * tiles of a group and items past the device
    count are never written.  No epilogue arithmetic -> the sums are the round-3 kernel's bit for bit."""
    from fgn_amd import lib
    L = lib.load()
    g = torch.Generator().manual_seed(n * tiles + cin)
    t_pad = L.fgn_winograd_t_pad(n * tiles)
    V = torch.randn(36, t_pad, cin, generator=g).cuda()
    U = (torch.randn(36, (cout + 127) // 128 * 128, cin, generator=g) / cin ** 0.5).cuda()
    st = torch.cuda.current_stream().cuda_stream

    def run(force, cnt=None):
        Mo = torch.full((36, t_pad, cout), -7.0, device='cuda')
        with _pw2(force):
            rc = L.fgn_winograd_gemm_f32(V.data_ptr(), U.data_ptr(), Mo.data_ptr(), None if cnt is None else cnt.data_ptr(),
                                         n, tiles, t_pad, cin, cout, U.shape[1], 36,
                                         None if sched is None else sched.data_ptr(), None, 0, st)
        assert rc == 0
        torch.cuda.synchronize()
        if sched is not None:
            assert int(sched.abs().max()) == 0            # the scheduler's counters are back at zero after the launch
        return Mo
    sched = None
    old, got, p4 = run(0), run(code), run(4)
    valid = n * tiles
    ref = torch.einsum('gtc,gnc->gtn', V[:, :valid].double(), U[:, :cout].double()).float()
    assert (got[:, :valid] - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    # every tile shape accumulates a sum in the same k order: the tile codes agree bit for bit (and with the round-3
    # persistent kernel; its non-persistent 32x32x2 form pairs the k's differently: last-bit differences)
    assert torch.equal(got[:, :valid], p4[:, :valid])
    assert (got[:, :valid] - old[:, :valid]).abs().max().item() <= 2e-6 * ref.abs().max().item()
    # the tile scheduler (workgroups pull their next tile from a counter) changes who computes a tile, not the tile
    sched = torch.zeros(L.fgn_gemm_sched_words(), dtype=torch.int32, device='cuda')
    for _ in range(3):                                       # back to back on the same counters: they reset themselves
        assert torch.equal(run(code), got)
    sched = None
    # rows of a group past its valid rows: the round-3 kernel writes the padding rows of a started 64-row tile,
    # this one writes none of them
    assert float((got[:, valid:] + 7.0).abs().max()) == 0.0 if valid < t_pad else True
    cnt = torch.tensor([n - 1], dtype=torch.int32, device='cuda')
    sched = torch.zeros(L.fgn_gemm_sched_words(), dtype=torch.int32, device='cuda')     # pulled tiles without valid rows
    part = run(code, cnt)
    v2 = (n - 1) * tiles
    assert torch.equal(part[:, :v2], got[:, :v2]) and float((part[:, v2:] + 7.0).abs().max()) == 0.0


# ------------------------------------------------------------------------------------------------
# conv_pw_streamk_kernel (round 4): the K loop of the last round's tiles shared among all workgroups
# ------------------------------------------------------------------------------------------------
class _streamk:
    """``with _streamk(mode):`` sets the Stream-K mode (fgn_conv2d_tune knob 2; 3 = every eligible launch)."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        from fgn_amd import lib
        L = lib.load()
        self.prev = L.fgn_conv2d_tune(2, self.mode)
        self.prev0 = L.fgn_conv2d_tune(0, 0)

    def __exit__(self, *a):
        from fgn_amd import lib
        L = lib.load()
        L.fgn_conv2d_tune(2, self.prev)
        L.fgn_conv2d_tune(0, self.prev0)


@pytest.mark.parametrize('rows,cin,cout,res,relu', [
    (64 * 400 + 5, 128, 64, True, True),        # 401 tiles x 4 K-tiles: ranges of 4 = whole tiles only
    (6504, 1024, 256, False, True),             # layer3 conv1 of a cfg3 episode: 408 tiles x 32 K-tiles, ranges of 13
    (4900, 512, 1024, True, True),              # 1232 tiles: one whole round + 208 tiles in pieces, residual epilogue
    (130, 1024, 512, True, False),              # 24 tiles: minimum range (4 K-tiles), most workgroups idle
    (64 * 9 + 1, 256, 4, False, False),         # Cout inside one 16-byte vector, 10 tiles
    (14700, 1024, 1024, False, True),           # 3680 tiles = 3 rounds + 608
])
def test_stream_k_pointwise_matches_fp64_and_the_whole_tile_kernels(rows, cin, cout, res, relu):
    from fgn_amd import ops, lib
    L = lib.load()
    g = torch.Generator().manual_seed(rows + cin + cout)
    x = torch.randn(rows, cin, 1, 1, generator=g)
    wt = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
    bn = dict(weight=torch.rand(cout, generator=g) + 0.5, bias=torch.randn(cout, generator=g) * 0.1,
              running_mean=torch.randn(cout, generator=g) * 0.1, running_var=torch.rand(cout, generator=g) + 0.5)
    r = torch.randn(rows, cout, 1, 1, generator=g) if res else None
    ref = _ref(x, wt, None, bn, 1, 0, r, relu)
    layer = ops.pack_conv(wt, bn=bn, relu=relu).to('cuda')
    xc, rc = _nhwc(x).cuda(), None if r is None else _nhwc(r).cuda()
    with _streamk(0):
        old = ops.conv2d(xc, layer, residual=rc).clone()
    sched = torch.zeros(L.fgn_gemm_sched_words(), dtype=torch.int32, device='cuda')
    keep, ops._sched = ops._sched, (lambda dev: sched)
    try:
        with _streamk(3):
            assert L.fgn_conv2d_kernel_id(rows, 1, 1, cin, cout, layer.cout_pad, 1, 1, 1, 0, 1, 0, int(res), 0) % 10 == 6
            got = ops.conv2d(xc, layer, residual=rc).clone()
            torch.cuda.synchronize()
            assert int(sched.abs().max()) == 0            # every ticket is back at zero
            for _ in range(3):                             # the order of arrival changes, the sums do not
                assert torch.equal(ops.conv2d(xc, layer, residual=rc), got)
            cnt = torch.tensor([rows - 77], dtype=torch.int32, device='cuda')
            part = torch.full((rows, 1, 1, cout), -7.0, device='cuda')
            ops.conv2d(xc, layer, residual=rc, n_img_dev=cnt, out=part)
            torch.cuda.synchronize()
            assert int(sched.abs().max()) == 0
    finally:
        ops._sched = keep
    scale = ref.abs().max().item() + 1e-6
    assert (got.cpu().permute(0, 3, 1, 2) - ref).abs().max().item() <= 2e-5 * scale + 1e-6
    assert (got - old).abs().max().item() <= 4e-6 * scale          # pieces are summed after the fact: rounding only
    assert torch.equal(part[:rows - 77], got[:rows - 77]) and float((part[rows - 77:] + 7.0).abs().max()) == 0.0


@pytest.mark.parametrize('n,tiles,cin,cout', [(3, 273, 1024, 128), (5, 4, 512, 64), (100, 4, 256, 256), (300, 4, 128, 512)])
def test_stream_k_grouped_gemm(n, tiles, cin, cout):
    """The grouped Winograd GEMM on conv_pw_streamk_kernel: banded tile order, groups cut at their valid rows, a device-side
    item count that empties some of the remaining tiles."""
    from fgn_amd import lib
    L = lib.load()
    g = torch.Generator().manual_seed(n * tiles + cin)
    t_pad = L.fgn_winograd_t_pad(n * tiles)
    V = torch.randn(36, t_pad, cin, generator=g).cuda()
    U = (torch.randn(36, (cout + 127) // 128 * 128, cin, generator=g) / cin ** 0.5).cuda()
    st = torch.cuda.current_stream().cuda_stream
    sched = torch.zeros(L.fgn_gemm_sched_words(), dtype=torch.int32, device='cuda')

    def run(mode, cnt=None):
        Mo = torch.full((36, t_pad, cout), -7.0, device='cuda')
        with _streamk(mode):
            nb = L.fgn_winograd_gemm_workspace_bytes(t_pad, cin, cout, 36)
            assert (nb > 0) == (mode == 3)
            ws = torch.empty(max(nb, 16), dtype=torch.uint8, device='cuda')
            rc = L.fgn_winograd_gemm_f32(V.data_ptr(), U.data_ptr(), Mo.data_ptr(), None if cnt is None else cnt.data_ptr(),
                                         n, tiles, t_pad, cin, cout, U.shape[1], 36, sched.data_ptr(), ws.data_ptr(), nb, st)
        assert rc == 0
        torch.cuda.synchronize()
        assert int(sched.abs().max()) == 0
        return Mo
    old, got = run(0), run(3)
    valid = n * tiles
    ref = torch.einsum('gtc,gnc->gtn', V[:, :valid].double(), U[:, :cout].double()).float()
    assert (got[:, :valid] - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    assert (got[:, :valid] - old[:, :valid]).abs().max().item() <= 4e-6 * ref.abs().max().item()
    for _ in range(3):
        assert torch.equal(run(3), got)
    cnt = torch.tensor([n - 1], dtype=torch.int32, device='cuda')
    part = run(3, cnt)
    v2 = (n - 1) * tiles
    assert torch.equal(part[:, :v2], got[:, :v2])
    lim = (v2 + 63) // 64 * 64                  # rows of a started 64-row tile are written, nothing beyond
    assert float((part[:, lim:] + 7.0).abs().max()) == 0.0 if lim < t_pad else True
