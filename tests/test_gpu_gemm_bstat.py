"""The activation-stationary GEMM (csrc/gemm_bstat.hip, SF_ALGO_BSTAT) through the C ABI against float64 (-m gpu): every
operand format (fp32 planes, fp16 rows, fp16 k-octets, grouped rows), every output format (fp32 planes, k-octets, both),
every epilogue with both residual formats, ragged M / K / pixel counts, one and two products, small grids (row ranges
split over workgroups) -- and agreement with the tiled kernels on the same problem."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X; torch.cuda.is_available() is False")
    return torch.device("cuda:0")


def _koct(x, dev, ops):
    """[n, K, P] fp32 -> fp16 k-octet Planes (through sf_pack_koct)."""
    from streamflow_amd.ops import Planes
    n, K, P = x.shape
    Ka = (K + 7) // 8 * 8
    Y = Planes(torch.zeros(n * Ka * P // 2 + 8, device=dev), 0, Ka * P, n, K, P, f16=True, koct=True)
    Y.base.view(torch.float16).fill_(1000.0)              # rows K .. Ka - 1 of the last octet: finite garbage (contract: zero weights there)
    ops.pack_koct(Planes.of(x.to(dev).contiguous()), Y)
    return Y


def _f16_rows(x, dev):
    from streamflow_amd.ops import Planes
    n, K, P = x.shape
    h = x.to(dev).half().contiguous()
    base = h.view(torch.float32) if (n * K * P) % 2 == 0 else None
    assert base is not None
    return Planes(base.view(-1), 0, K * P, n, K, P, f16=True, koct=False), h


def _ref(epi, v, R, dw_w, dw_b, ops):
    if epi == ops.EPI_GELU:
        return F.gelu(v)
    if epi == ops.EPI_RELU:
        return torch.relu(v)
    if epi == ops.EPI_RES:
        return R + v
    if epi == ops.EPI_RES_GELU:
        return F.gelu(R + v)
    if epi == ops.EPI_RES_GELU_DW1:
        t = F.gelu(R + v)
        return F.gelu(t + (dw_w[None, :, None] * t + dw_b[None, :, None]))
    return v


def _weights_eff(A, M, K, single):
    hi = A.hi.float().permute(1, 0, 2).reshape(A.lda_h, -1)[:M, :K].double().cpu()
    lo = A.lo.float().permute(1, 0, 2).reshape(A.lda_h, -1)[:M, :K].double().cpu()
    return (hi if single else hi + lo) / A.split_scale


@pytest.mark.parametrize("seed", range(40))
def test_bstat_random(dev, seed):
    from streamflow_amd import _lib, ops
    from streamflow_amd.ops import PackedLinear, Planes
    rng = np.random.default_rng(9000 + seed)
    M = int(rng.choice([6, 31, 64, 65, 126, 128, 192, 200, 324, 486, 640, 960]))
    K = int(rng.choice([65, 100, 128, 192, 256, 324, 384, 486, 576, 640]))
    P = int(rng.choice([32, 96, 128, 130, 1000, 2048, 7040])) & ~1          # (even: fp16 rows are viewed through fp32 storage)
    n = int(rng.integers(1, 4))
    blay = str(rng.choice(["f32", "koct", "rows"]))
    cfmt = int(rng.choice([0, 2, 3]))
    single = bool(rng.integers(0, 2))
    epi = int(rng.choice([ops.EPI_NONE, ops.EPI_GELU, ops.EPI_RELU, ops.EPI_RES, ops.EPI_RES_GELU, ops.EPI_RES_GELU_DW1]))
    needs_r = epi in (ops.EPI_RES, ops.EPI_RES_GELU, ops.EPI_RES_GELU_DW1)
    rk = needs_r and epi == ops.EPI_RES_GELU_DW1 and cfmt != 2 and bool(rng.integers(0, 2))     # k-octet residual (convc1's form)
    if needs_r and K > (512 if rk else 384):
        K = 512 if rk else 384                                              # (register budget of the residual kernels)
    if cfmt == 2 and epi not in (ops.EPI_NONE, ops.EPI_GELU, ops.EPI_RES_GELU):
        epi, needs_r, rk = ops.EPI_GELU, False, False
    alpha = float(rng.choice([1.0, 0.25]))
    g = torch.Generator().manual_seed(seed)
    Wt = torch.randn(M, K, generator=g) / K ** 0.5
    bias = torch.randn(M, generator=g) * 0.1 if rng.random() < 0.8 else None
    X = torch.randn(n, K, P, generator=g)
    R = torch.randn(n, M, P, generator=g)
    dw_w, dw_b = torch.randn(M, generator=g) * 0.5, torch.randn(M, generator=g) * 0.1
    A = PackedLinear(Wt.view(M, K, 1, 1), bias, dev)
    A.single = single
    if blay == "koct":
        Xp = _koct(X, dev, ops)
    elif blay == "rows":
        Xp, _ = _f16_rows(X, dev)
    else:
        Xp = Planes.of(X.to(dev))
    Mo = (M + 7) // 8 * 8
    y32 = torch.full((n, M, P), float("nan"), device=dev)
    if cfmt == 2:
        # (NaN-filled: every row of the last octet must come out finite -- the next layer multiplies the rows past M by zero weights)
        Y = Planes(torch.full((n * Mo * P // 2 + 8,), float("nan"), device=dev), 0, Mo * P, n, M, P, f16=True, koct=True)
    else:
        Y = Planes.of(y32)
        if cfmt == 3:
            sh = ops.new_shadow(Y, dev)
            sh.base.view(torch.float16).fill_(7.0)                             # rows >= M of a last octet must keep this
            Y = Planes(Y.base, Y.off, Y.img_stride, Y.n_img, Y.rows, Y.P, shadow=sh)
    Rp = None
    if needs_r:
        Rp = _koct(R, dev, ops) if rk else Planes.of(R.to(dev))
    prev = ops.set_precision("f16x2")
    try:
        ops.gemm(A, Xp, Y, epi, R=Rp, dw_w=dw_w.to(dev), dw_b=dw_b.to(dev), alpha=alpha, algo=_lib.ALGO_BSTAT)
        torch.cuda.synchronize()
    finally:
        ops.set_precision(prev)
    Xh = X.half().double()
    Rr = (R.half().double() if rk else R.double())
    v = alpha * (torch.einsum("mk,zkp->zmp", _weights_eff(A, M, K, single), Xh) + (bias.double()[None, :, None] if bias is not None else 0.0))
    ref = _ref(epi, v, Rr, dw_w.double(), dw_b.double(), ops)
    tag = (M, K, P, n, blay, cfmt, single, epi, rk)
    scale = max(1.0, ref.abs().max().item())
    if cfmt != 2:
        err = (y32.double().cpu() - ref).abs().max().item()
        assert err < 3e-5 * scale, tag + (err,)
    if cfmt >= 2:
        if cfmt == 2:
            whole = Y.base.view(torch.float16)[: n * Mo * P].view(n, Mo // 8, P, 8)
            assert bool(torch.isfinite(whole.float()).all()), tag
        got = (Y if cfmt == 2 else Y.shadow).tensor().double().cpu()
        # fp16 storage: one rounding (2^-11) (+ the polynomial GELU of the k-octet-only output: 5.2e-5)
        err = (got - ref).abs()
        assert bool((err <= 2.0 ** -11 * ref.abs() * 1.01 + 8e-5).all()), tag + (err.max().item(),)
        if cfmt == 3 and M % 8:
            oc = Y.shadow.base.view(torch.float16).view(n, Mo // 8, P, 8)[:, -1, :, M % 8:]
            assert bool((oc == 7.0).all()), tag                                   # someone else's rows of the last octet


@pytest.mark.parametrize("M,K", [(960, 640), (640, 640), (486, 324), (384, 256), (256, 384), (128, 128), (192, 128)])
@pytest.mark.parametrize("single", [False, True])
def test_bstat_matches_tiled(dev, M, K, single):
    """Same problem on both kernel families (the update block's layer shapes, k-octet hand-over in and out, GELU): equal up
    to the summation order of the k-steps and one fp16 ulp of the stored result."""
    from streamflow_amd import _lib, ops
    from streamflow_amd.ops import PackedLinear, Planes
    if not ops.uses_dma_tile(M):
        pytest.skip("the tiled family takes a k-octet operand on its 128-row tile only")
    n, P = 3, 7040
    g = torch.Generator().manual_seed(M + K)
    A = PackedLinear(torch.randn(M, K, 1, 1, generator=g) / K ** 0.5, torch.randn(M, generator=g) * 0.1, dev)
    A.single = single
    Xp = _koct(torch.randn(n, K, P, generator=g), dev, ops)
    outs = []
    prev = ops.set_precision("f16x2")
    try:
        for algo in (_lib.ALGO_TILED, _lib.ALGO_BSTAT):
            Mo = (M + 7) // 8 * 8
            Y = Planes(torch.zeros(n * Mo * P // 2 + 8, device=dev), 0, Mo * P, n, M, P, f16=True, koct=True)
            ops.gemm(A, Xp, Y, ops.EPI_GELU, algo=algo)
            outs.append(Y.tensor().float().cpu())
    finally:
        ops.set_precision(prev)
    d = (outs[0] - outs[1]).abs()
    assert bool((d <= 2.0 ** -10 * outs[0].abs() + 2e-5).all()), d.max().item()


def test_bstat_grouped_rows_and_small_grid(dev):
    """The flow head's operand: '(B T) C -> B (T C)' grouped rows (b_group = 128) in k-octet and fp32 form, residual through the
    same grouped view, on a grid small enough that the rows are cut into ranges (msplit > 1)."""
    from streamflow_amd import _lib, ops
    from streamflow_amd.ops import PackedLinear, Planes
    Bc, Pn, C, P, M = 2, 3, 128, 352, 384
    g = torch.Generator().manual_seed(5)
    nets = torch.randn(Bc * Pn, C, P, generator=g)
    Wt = torch.randn(M, Pn * C, generator=g) / (Pn * C) ** 0.5
    bias = torch.randn(M, generator=g) * 0.1
    A = PackedLinear(Wt.view(M, Pn * C, 1, 1), bias, dev)
    nd = nets.to(dev).contiguous()
    flat = nd.view(-1)
    Xg = Planes(flat, 0, Pn * C * P, Bc, Pn * C, P, group=C, group_stride=C * P)
    sh = _koct(nets, dev, ops)
    Xk = Planes(sh.base, sh.off, Pn * sh.img_stride, Bc, Pn * C, P, f16=True, koct=True, group=C, group_stride=sh.img_stride)
    ref_in = nets.view(Bc, Pn * C, P).half().double()
    W_eff = _weights_eff(A, M, Pn * C, False)
    v = torch.einsum("mk,zkp->zmp", W_eff, ref_in) + bias.double()[None, :, None]
    prev = ops.set_precision("f16x2")
    try:
        for Xp in (Xg, Xk):
            y = torch.full((Bc, M, P), float("nan"), device=dev)
            ops.gemm(A, Xp, Planes.of(y), ops.EPI_GELU, algo=_lib.ALGO_BSTAT)
            assert (y.double().cpu() - F.gelu(v)).abs().max().item() < 3e-5
        # residual through the grouped fp32 view (M = K = 384: flow_head.ffn1_2's addressing)
        W2 = torch.randn(Pn * C, M, generator=g) / M ** 0.5
        A2 = PackedLinear(W2.view(Pn * C, M, 1, 1), None, dev)
        hid = torch.randn(Bc, M, P, generator=g)
        y = torch.full((Bc, Pn * C, P), float("nan"), device=dev)
        ops.gemm(A2, Planes.of(hid.to(dev)), Planes.of(y), ops.EPI_RES, R=Xg, algo=_lib.ALGO_BSTAT)
        ref = nets.view(Bc, Pn * C, P).double() + torch.einsum("mk,zkp->zmp", _weights_eff(A2, Pn * C, M, False), hid.half().double())
        assert (y.double().cpu() - ref).abs().max().item() < 3e-5
    finally:
        ops.set_precision(prev)


def test_bstat_refuses_what_it_cannot_run(dev):
    from streamflow_amd import _lib, ops
    from streamflow_amd.ops import PackedLinear, Planes
    A = PackedLinear(torch.randn(64, 960, 1, 1), None, dev)
    x, y = torch.randn(1, 960, 64, device=dev), torch.empty(1, 64, 64, device=dev)
    prev = ops.set_precision("f16x2")
    try:
        with pytest.raises(RuntimeError, match="SF_ALGO_BSTAT"):
            ops.gemm(A, Planes.of(x), Planes.of(y), algo=_lib.ALGO_BSTAT)
        ops.gemm(A, Planes.of(x), Planes.of(y))                                 # (auto: the tiled kernel takes it)
    finally:
        ops.set_precision(prev)


@pytest.mark.parametrize("M,K,P", [(192, 128, 7040), (640, 96, 2048), (324, 192, 7040)])
@pytest.mark.parametrize("epi_name", ["none", "res_gelu"])
def test_bstat_general_kernel_short_k_many_msteps(dev, M, K, P, epi_name):
    """gemm_bstat_kernel (the non-pipelined form) with k-octet-only output, >= 3 m-steps and a short K: the first two stages of
    every m-step wait with a count that includes the previous epilogue's stores (BsArgs.e_ops, ADVICE r4: the field was left
    uninitialised once) -- a wrong count reads a weight stage before its DMA pieces have landed.  Repeated launches, exact
    reference."""
    from streamflow_amd import _lib, ops
    from streamflow_amd.ops import PackedLinear, Planes
    epi = ops.EPI_NONE if epi_name == "none" else ops.EPI_RES_GELU
    n = 3
    g = torch.Generator().manual_seed(M * 7 + K)
    Wt = torch.randn(M, K, generator=g) / K ** 0.5
    bias = torch.randn(M, generator=g) * 0.1
    X = torch.randn(n, K, P, generator=g)
    R = torch.randn(n, M, P, generator=g)
    A = PackedLinear(Wt.view(M, K, 1, 1), bias, dev)
    Xp = _koct(X, dev, ops)
    Rp = Planes.of(R.to(dev)) if epi == ops.EPI_RES_GELU else None
    Mo = (M + 7) // 8 * 8
    v = torch.einsum("mk,zkp->zmp", _weights_eff(A, M, K, False), X.half().double()) + bias.double()[None, :, None]
    ref = v if epi == ops.EPI_NONE else F.gelu(R.double() + v)
    prev = ops.set_precision("f16x2")
    try:
        for rep in range(6):
            Y = Planes(torch.full((n * Mo * P // 2 + 8,), float("nan"), device=dev), 0, Mo * P, n, M, P, f16=True, koct=True)
            ops.gemm(A, Xp, Y, epi, R=Rp, algo=_lib.ALGO_BSTAT)
            got = Y.tensor().double().cpu()
            err = (got - ref).abs()
            assert bool((err <= 2.0 ** -11 * ref.abs() * 1.01 + 8e-5).all()), (M, K, P, epi_name, rep, err.max().item())
    finally:
        ops.set_precision(prev)
