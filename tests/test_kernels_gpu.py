"""Numerics of every HIP kernel against a plain PyTorch fp32 reference (GPU only)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _lib_loaded():
    from distributed_tensorflow_amd.ops import _lib
    _lib.kernels()
    with open("/proc/self/maps") as f:
        return "libdtf_kernels.so" in f.read()


def test_native_library_is_loaded():
    assert _lib_loaded()


@pytest.mark.parametrize("n", [1, 7, 64, 1000, 407050, 1 << 20 | 3])
@pytest.mark.parametrize("gdtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_adam_matches_reference(n, gdtype, mode):
    from distributed_tensorflow_amd.ops import optim as K
    torch.manual_seed(n + mode)
    npad = (n + 63) // 64 * 64
    p = torch.randn(npad, device=DEV); g = torch.randn(npad, device=DEV).to(gdtype)
    m = torch.zeros(npad, device=DEV); v = torch.zeros(npad, device=DEV)
    sh = torch.zeros(npad, device=DEV, dtype=torch.bfloat16)
    pc, gc, mc, vc = p.cpu().clone(), g.cpu().clone(), m.cpu().clone(), v.cpu().clone()
    shc = sh.cpu().clone()
    hg = K.OptHyper(DEV, lr=1e-2, weight_decay=0.01, grad_scale=0.5)
    hc = K.OptHyper("cpu", lr=1e-2, weight_decay=0.01, grad_scale=0.5)
    for _ in range(3):
        hg.tick(); hc.tick()
        K.adam_apply(p, g, m, v, hg, shadow=sh, mode=mode, n=n)
        K.adam_apply(pc, gc, mc, vc, hc, shadow=shc, mode=mode, n=n)
    torch.testing.assert_close(hg.buf.cpu(), hc.buf, rtol=1e-6, atol=1e-7)
    torch.testing.assert_close(p.cpu()[:n], pc[:n], rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(m.cpu()[:n], mc[:n], rtol=2e-5, atol=1e-7)
    torch.testing.assert_close(v.cpu()[:n], vc[:n], rtol=2e-5, atol=1e-7)
    torch.testing.assert_close(sh.cpu()[:n].float(), shc[:n].float(), rtol=1e-2, atol=1e-2)
    # untouched tail
    assert torch.equal(p.cpu()[n:], pc[n:])


@pytest.mark.parametrize("n", [5, 4096, 100003])
@pytest.mark.parametrize("gdtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_sgdm_matches_reference(n, gdtype, mode):
    from distributed_tensorflow_amd.ops import optim as K
    torch.manual_seed(n)
    npad = (n + 63) // 64 * 64
    p = torch.randn(npad, device=DEV); g = torch.randn(npad, device=DEV).to(gdtype)
    vel = torch.zeros(npad, device=DEV)
    pc, gc, vc = p.cpu().clone(), g.cpu().clone(), vel.cpu().clone()
    hg = K.OptHyper(DEV, lr=0.1, momentum=0.9, weight_decay=1e-4)
    hc = K.OptHyper("cpu", lr=0.1, momentum=0.9, weight_decay=1e-4)
    for _ in range(3):
        K.sgdm_apply(p, g, vel, hg, mode=mode, n=n)
        K.sgdm_apply(pc, gc, vc, hc, mode=mode, n=n)
    torch.testing.assert_close(p.cpu()[:n], pc[:n], rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(vel.cpu()[:n], vc[:n], rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("tdtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("fdtype", [torch.float32, torch.bfloat16])
def test_bucket_flatten_unflatten(tdtype, fdtype):
    from distributed_tensorflow_amd.ops import bucket as B
    torch.manual_seed(0)
    shapes = [(3,), (64, 3, 7, 7), (1000, 2048), (17,), (512,), (10, 512), (1,)] + [(33, 5)] * 60
    ts = [torch.randn(s, device=DEV).to(tdtype) for s in shapes]
    offs, o = [], 0
    for t in ts:
        offs.append(o); o += (t.numel() + 63) // 64 * 64
    flat = torch.full((o,), 7.0, device=DEV, dtype=fdtype)
    B.flatten(ts, flat, offs, scale=0.25)
    for t, off in zip(ts, offs):
        ref = (t.float() * 0.25).to(fdtype).reshape(-1)
        assert torch.equal(flat[off:off + t.numel()], ref)
    outs = [torch.zeros_like(t) for t in ts]
    B.unflatten(flat, outs, offs, scale=4.0)
    for t, u, off in zip(ts, outs, offs):
        ref = (flat[off:off + t.numel()].float() * 4.0).to(tdtype).reshape(t.shape)
        assert torch.equal(u, ref)
    # unaligned source pointer takes the scalar path
    base = torch.randn(1001, device=DEV).to(tdtype)
    flat2 = torch.zeros(1024, device=DEV, dtype=fdtype)
    B.flatten([base[1:]], flat2, [0])
    assert torch.equal(flat2[:1000], base[1:].float().to(fdtype))


def test_scale_cast_nary_sumsq():
    from distributed_tensorflow_amd.ops import bucket as B
    torch.manual_seed(1)
    n = 100003 + 5
    n = n // 8 * 8 + 3
    a = torch.randn(n, device=DEV)
    out = torch.empty(n, device=DEV, dtype=torch.bfloat16)
    s = torch.tensor([0.5], device=DEV)
    B.scale_cast(a, out, 2.0, s)
    assert torch.equal(out, a.to(torch.bfloat16))
    for dt in (torch.float32, torch.bfloat16):
        dst = torch.randn(n, device=DEV).to(dt)
        srcs = [torch.randn(n, device=DEV).to(dt) for _ in range(7)]
        ws = [0.125 * (i + 1) for i in range(7)]
        ref = dst.float() * 0.5
        for s_, w in zip(srcs, ws):
            ref = torch.addcmul(ref, s_.float(), torch.tensor(w, device=DEV))
        B.nary_axpy(dst, srcs, ws, w_dst=0.5)
        torch.testing.assert_close(dst.float(), ref.to(dt).float(), rtol=1e-2 if dt == torch.bfloat16 else 1e-5,
                                   atol=1e-2 if dt == torch.bfloat16 else 1e-5)
    x = torch.randn(n, device=DEV)
    torch.testing.assert_close(B.sumsq(x)[0], (x.double() ** 2).sum().float(), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("B_,C", [(100, 10), (7, 3), (64, 1000), (33, 1001), (16, 30522), (5, 65), (300, 64)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("ldt", [torch.int64, torch.int32])
def test_softmax_ce_fwd_bwd(B_, C, dtype, ldt):
    from distributed_tensorflow_amd.ops.softmax_ce import softmax_ce_eval, softmax_cross_entropy
    torch.manual_seed(B_ * C)
    x = (torch.randn(B_, C, device=DEV) * 3).to(dtype).requires_grad_(True)
    y = torch.randint(0, C, (B_,), device=DEV).to(ldt)
    y[0] = -100
    g = torch.rand(B_, device=DEV) + 0.5
    loss = softmax_cross_entropy(x, y)
    loss.backward(g)
    xr = x.detach().float().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(xr, y.long(), reduction="none", ignore_index=-100)
    ref.backward(g)
    tol = dict(rtol=1e-4, atol=1e-4) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(loss, ref, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(x.grad.float(), xr.grad, **tol)
    l2, corr = softmax_ce_eval(x.detach(), y)
    torch.testing.assert_close(l2, ref.detach(), rtol=1e-4, atol=1e-4)
    exp_corr = (x.detach().float().argmax(1) == y.long())
    assert torch.equal(corr.bool(), exp_corr)


def _asym(M, N, K, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    a = torch.randn(M, K, generator=g)
    b = torch.randn(N, K, generator=g)
    return a.to(DEV).to(torch.bfloat16), b.to(DEV).to(torch.bfloat16)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (100, 512, 784), (256, 512, 8), (300, 264, 200), (1024, 1024, 1024),
                                   (50, 8, 4096), (4096, 64, 256)])
@pytest.mark.parametrize("splits", [1, 3])
def test_gemm_layouts(M, N, K, splits):
    from distributed_tensorflow_amd.ops import gemm
    a, b = _asym(M, N, K, M + N + K)
    af, bf = a.float(), b.float()
    tol = dict(rtol=2e-2, atol=2e-2 * math.sqrt(K))
    ref = af @ bf.t()
    # NT with identity check of orientation: asymmetric operands
    out = gemm.gemm_nt(a, b, splits=splits)
    torch.testing.assert_close(out.float(), ref, **tol)
    bias = torch.randn(N, device=DEV)
    out = gemm.gemm_nt(a, b, bias=bias, activation="relu", splits=splits, out_dtype=torch.float32)
    torch.testing.assert_close(out, torch.relu(ref + bias), rtol=1e-2, atol=1e-2 * math.sqrt(K))
    out = gemm.gemm_nt(a, b, bias=bias.to(torch.bfloat16), activation="gelu", splits=splits)
    torch.testing.assert_close(out.float(), torch.nn.functional.gelu(ref + bias.to(torch.bfloat16).float()), **tol)
    if M % 8 == 0:
        # NN: [M,K] @ [K,N]  with b^T stored as [K,N]
        bt = b.t().contiguous()          # [K, N]
        out = gemm.gemm_nn(a, bt, splits=splits)
        torch.testing.assert_close(out.float(), ref, **tol)
        # TN: a^T stored [K,M], b^T stored [K,N]
        at = a.t().contiguous()
        out = gemm.gemm_tn(at, bt, splits=splits)
        torch.testing.assert_close(out.float(), ref, **tol)


def test_gemm_identity_orientation():
    from distributed_tensorflow_amd.ops import gemm
    n = 128
    eye = torch.eye(n, device=DEV, dtype=torch.bfloat16)
    b = (torch.arange(n * n, device=DEV).reshape(n, n) % 251).to(torch.bfloat16)  # asymmetric, exact in bf16
    out = gemm.gemm_nt(eye, b)       # I @ b^T
    assert torch.equal(out, b.t().contiguous())
    out = gemm.gemm_nn(eye, b)       # I @ b
    assert torch.equal(out, b)
    out = gemm.gemm_tn(eye, b)       # I^T @ b
    assert torch.equal(out, b)


def test_dense_autograd_matches_torch():
    from distributed_tensorflow_amd.ops.dense import linear
    torch.manual_seed(0)
    for act in (None, "relu", "gelu"):
        x = torch.randn(96, 784, device=DEV).to(torch.bfloat16).requires_grad_(True)
        w = (torch.randn(512, 784, device=DEV) * 0.05).to(torch.bfloat16).requires_grad_(True)
        b = torch.randn(512, device=DEV).to(torch.bfloat16).requires_grad_(True)
        y = linear(x, w, b, act)
        gy = torch.randn_like(y)
        y.backward(gy)
        xr, wr, br = (t.detach().float().requires_grad_(True) for t in (x, w, b))
        yr = torch.nn.functional.linear(xr, wr, br)
        yr = torch.relu(yr) if act == "relu" else torch.nn.functional.gelu(yr) if act == "gelu" else yr
        yr.backward(gy.float())
        torch.testing.assert_close(y.float(), yr, rtol=2e-2, atol=5e-2)
        torch.testing.assert_close(x.grad.float(), xr.grad, rtol=3e-2, atol=8e-2)
        torch.testing.assert_close(w.grad.float(), wr.grad, rtol=3e-2, atol=3e-1)
        torch.testing.assert_close(b.grad.float(), br.grad, rtol=3e-2, atol=3e-1)


@pytest.mark.parametrize("N,C,H,W", [(4, 64, 14, 14), (2, 16, 32, 32), (3, 2048, 7, 7), (8, 256, 5, 3), (2, 24, 9, 9)])
@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("use_res", [False, True])
def test_batch_norm_act_matches_torch(N, C, H, W, relu, use_res):
    from distributed_tensorflow_amd.ops.norm import batch_norm_act
    torch.manual_seed(C + H)
    x = (torch.randn(N, C, H, W, device=DEV) * 2 + 0.5).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    res = torch.randn(N, C, H, W, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    g = (torch.rand(C, device=DEV) + 0.5); b = torch.randn(C, device=DEV)
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    x1 = x.clone().requires_grad_(True); r1 = res.clone().requires_grad_(True)
    g1 = g.clone().requires_grad_(True); b1 = b.clone().requires_grad_(True)
    y = batch_norm_act(x1, g1, b1, rm, rv, True, 0.1, 1e-5, relu, r1 if use_res else None)
    assert y.is_contiguous(memory_format=torch.channels_last)
    dy = torch.randn_like(y)
    y.backward(dy)
    # fp32 reference
    xr = x.float().requires_grad_(True); rr = res.float().requires_grad_(True)
    gr = g.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    rmr, rvr = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    yr = torch.nn.functional.batch_norm(xr, rmr, rvr, gr, br, True, 0.1, 1e-5)
    if use_res:
        yr = yr + rr
    if relu:
        yr = torch.relu(yr)
    yr.backward(dy.float())
    torch.testing.assert_close(y.float(), yr, rtol=2e-2, atol=3e-2)
    torch.testing.assert_close(rm, rmr, rtol=1e-3, atol=1e-3)
    torch.testing.assert_close(rv, rvr, rtol=1e-3, atol=1e-3)
    # gradients: the relu mask is taken from the bf16 output, so compare where the reference is not at the kink
    torch.testing.assert_close(x1.grad.float(), xr.grad, rtol=5e-2, atol=5e-2)
    scale = max(1.0, float(gr.grad.abs().max()))
    torch.testing.assert_close(g1.grad / scale, gr.grad / scale, rtol=3e-2, atol=3e-2)
    torch.testing.assert_close(b1.grad / scale, br.grad / scale, rtol=3e-2, atol=3e-2)
    if use_res:
        torch.testing.assert_close(r1.grad.float(), rr.grad, rtol=2e-2, atol=2e-2)
    # inference path
    ye = batch_norm_act(x, g, b, rm, rv, False, 0.1, 1e-5, relu, res if use_res else None)
    yer = torch.nn.functional.batch_norm(x.float(), rm, rv, g, b, False, 0.1, 1e-5)
    yer = torch.relu(yer + (res.float() if use_res else 0)) if relu else yer + (res.float() if use_res else 0)
    torch.testing.assert_close(ye.float(), yer, rtol=2e-2, atol=3e-2)


@pytest.mark.parametrize("cfg", [
    # (N, C, H, W, K, k, stride, pad)
    (4, 64, 14, 14, 256, 1, 1, 0), (4, 64, 14, 14, 256, 1, 2, 0), (3, 64, 15, 13, 64, 3, 1, 1),
    (2, 128, 28, 28, 128, 3, 2, 1), (2, 8, 33, 31, 64, 7, 2, 3), (5, 256, 7, 7, 512, 3, 1, 1),
    (2, 16, 32, 32, 32, 3, 2, 1), (1, 512, 9, 9, 24, 3, 1, 0),
    # 128 x 64 output tile (at most 64 columns in some pass), odd edges
    (3, 256, 20, 20, 64, 1, 1, 0), (2, 64, 30, 34, 64, 3, 2, 1), (3, 40, 11, 9, 48, 3, 1, 1), (2, 64, 12, 12, 200, 1, 1, 0),
    # 1x1 stride 2: data gradient as one product stored at every other pixel
    (3, 256, 28, 28, 512, 1, 2, 0), (2, 72, 10, 6, 40, 1, 2, 0),
    # first layers: pixel-pair form (<= 4 channels, stride 2, even width) and 8-channel padding
    (3, 3, 64, 64, 64, 7, 2, 3), (2, 3, 38, 30, 32, 5, 2, 2), (2, 1, 28, 28, 32, 3, 2, 1), (2, 4, 16, 20, 16, 2, 2, 0),
    (4, 3, 32, 32, 16, 3, 1, 1), (5, 1, 28, 28, 32, 3, 1, 0), (2, 3, 17, 15, 24, 3, 1, 1), (2, 12, 17, 15, 24, 3, 1, 1),
])
def test_implicit_gemm_conv_matches_fp32_autograd(cfg):
    """y, dx and dw of the framework's convolution against fp32 torch autograd on the same bf16-rounded inputs."""
    from distributed_tensorflow_amd.ops import conv
    N, C, H, W, K, k, stride, pad = cfg
    torch.manual_seed(sum(cfg))
    x = torch.randn(N, C, H, W, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(K, C, k, k, device=DEV) * (1.0 / math.sqrt(C * k * k))).to(torch.bfloat16)
    w = w.contiguous(memory_format=torch.channels_last)
    xr = x.float().requires_grad_(True); wr = w.float().requires_grad_(True)
    yr = torch.nn.functional.conv2d(xr, wr, None, stride, pad)
    dy = torch.randn(yr.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1)).to(torch.bfloat16)
    yr.backward(dy.float())
    ref = (yr.detach(), xr.grad, wr.grad)
    for be in ("own",):
        x1 = x.clone().requires_grad_(True); w1 = w.clone().requires_grad_(True)
        y = conv.conv2d(x1, w1, stride, pad)
        assert y.shape == yr.shape and y.dtype == torch.bfloat16
        y.backward(dy)
        for a, b, name in zip((y.float(), x1.grad.float(), w1.grad.float()), ref, ("y", "dx", "dw")):
            scale = max(1.0, float(b.abs().max()))
            torch.testing.assert_close(a / scale, b / scale, rtol=2e-2, atol=1e-2,
                                       msg=lambda m: be + " " + name + ": " + m)


def test_first_layer_conv_hands_over_batch_norm_statistics():
    """The few-channel first layer fills the partial-statistics rows like any other convolution."""
    from distributed_tensorflow_amd.ops import conv
    torch.manual_seed(5)
    x = torch.randn(6, 3, 32, 32, device=DEV).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(64, 3, 7, 7, device=DEV) * 0.1).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    y, st = conv.conv2d(x, w, 2, 3, bn_stats=True)
    assert st is not None
    yf = y.float().permute(0, 2, 3, 1).reshape(-1, 64)
    torch.testing.assert_close(st[:, 0].sum(0), yf.sum(0), rtol=1e-3, atol=1e-2)
    torch.testing.assert_close(st[:, 1].sum(0), (yf * yf).sum(0), rtol=1e-3, atol=1e-2)


def test_gemm_epilogue_accumulates_onto_existing_tensor():
    from distributed_tensorflow_amd.ops import gemm
    torch.manual_seed(0)
    a = torch.randn(300, 64, device=DEV).to(torch.bfloat16)
    b = torch.randn(64, 136, device=DEV).to(torch.bfloat16)       # [K, N]
    base = torch.randn(300, 136, device=DEV).to(torch.bfloat16)
    out = torch.empty_like(base)
    gemm.gemm_raw(gemm.NN, a, b, out, 300, 136, 64, 64, 136, 136, add=base)
    ref = ((a.float() @ b.float()).to(torch.bfloat16).float() + base.float()).to(torch.bfloat16)
    torch.testing.assert_close(out.float(), ref.float(), rtol=2e-2, atol=6e-2)


def test_resnet_train_step_decreases_loss():
    from distributed_tensorflow_amd import bench_lib
    from distributed_tensorflow_amd.comm.backend import Comm
    st = bench_lib.make_resnet50_strategy(Comm(0, 1, torch.device(DEV)), classes=16, lr=0.05)
    x, y = bench_lib.synthetic_imagenet_batch(16, DEV, size=64, classes=16)
    losses = [float(st.train_step(x, y)) for _ in range(12)]
    assert all(l == l for l in losses)
    assert losses[-1] < losses[0]


@pytest.mark.parametrize("rows,H", [(7, 768), (300, 128), (64, 1024), (33, 2048), (5, 40)])
@pytest.mark.parametrize("pdt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("use_res", [False, True])
def test_layer_norm_matches_torch(rows, H, pdt, use_res):
    from distributed_tensorflow_amd.ops.layernorm import layer_norm
    torch.manual_seed(H)
    x = torch.randn(rows, H, device=DEV).to(torch.bfloat16)
    r = torch.randn(rows, H, device=DEV).to(torch.bfloat16)
    g = (torch.rand(H, device=DEV) + 0.5).to(pdt); b = torch.randn(H, device=DEV).to(pdt)
    x1, r1, g1, b1 = (t.clone().requires_grad_(True) for t in (x, r, g, b))
    if use_res:
        y, s = layer_norm(x1, g1, b1, 1e-5, residual=r1, return_sum=True)
        dy, dsum = torch.randn_like(y), torch.randn_like(y)
        (y.float() * dy.float()).sum().add((s.float() * dsum.float()).sum()).backward()
    else:
        y = layer_norm(x1, g1, b1, 1e-5)
        dy = torch.randn_like(y)
        y.backward(dy)
    xr, rr, gr, br = (t.float().clone().requires_grad_(True) for t in (x, r, g, b))
    sr = (xr + rr) + ((xr + rr).to(torch.bfloat16).float() - (xr + rr)).detach() if use_res else xr
    yr = torch.nn.functional.layer_norm(sr, (H,), gr, br, 1e-5)
    if use_res:
        ((yr * dy.float()).sum() + (sr * dsum.float()).sum()).backward()
        torch.testing.assert_close(s.float(), sr.detach(), rtol=1e-2, atol=1e-2)
    else:
        yr.backward(dy.float())
    torch.testing.assert_close(y.float(), yr.detach(), rtol=2e-2, atol=3e-2)
    torch.testing.assert_close(x1.grad.float(), xr.grad, rtol=3e-2, atol=6e-2)
    if use_res:
        torch.testing.assert_close(r1.grad.float(), rr.grad, rtol=3e-2, atol=6e-2)
    sc = max(1.0, float(gr.grad.abs().max()))
    torch.testing.assert_close(g1.grad.float() / sc, gr.grad / sc, rtol=3e-2, atol=3e-2)
    torch.testing.assert_close(b1.grad.float() / sc, br.grad / sc, rtol=3e-2, atol=3e-2)


@pytest.mark.parametrize("rows,H,N", [(300, 256, 768), (4096, 3072, 768), (65, 128, 40)])
def test_layer_norm_backward_hands_the_bias_gradient_to_the_linear_layer(rows, H, N):
    """``LN(x W^T + b + res)``: the norm's backward kernel sums the columns of the gradient it writes, the linear
    layer takes them as its bias gradient instead of a pass of its own. Same result as without the hand-over and as
    fp32 autograd; a gradient that is not the norm's untouched output is never matched."""
    from distributed_tensorflow_amd.ops import dense, layernorm
    torch.manual_seed(rows + H)
    x = torch.randn(rows, H, device=DEV).bfloat16()
    w = (torch.randn(N, H, device=DEV) / H ** 0.5).bfloat16()
    b = torch.randn(N, device=DEV).bfloat16()
    res = torch.randn(rows, N, device=DEV).bfloat16()
    gamma, beta = torch.rand(N, device=DEV) + 0.5, torch.randn(N, device=DEV)
    wts = torch.rand(rows, N, device=DEV) + 0.5

    def run(hand_over, second_use=False):
        layernorm._hand_over = hand_over
        layernorm._handed.clear()
        bi, wi = b.clone().requires_grad_(True), w.clone().requires_grad_(True)
        y = dense.linear(x, wi, bi)
        out = layernorm.layer_norm(y, gamma, beta, 1e-12, residual=res)
        loss = (out.float() * wts).sum()
        if second_use:
            loss = loss + (y.float() * 0.5).sum()        # autograd then sums two gradients of y: no match
        loss.backward()
        taken = hand_over and not layernorm._handed
        return bi.grad.float(), wi.grad.float(), taken
    try:
        gb1, gw1, taken = run(True)
        assert taken, "the linear layer did not pick the column sums up"
        gb0, gw0, _ = run(False)
        gb2, _, taken2 = run(True, second_use=True)
        assert not taken2
    finally:
        layernorm._hand_over = True
        layernorm._handed.clear()
    wf, bf = w.float().requires_grad_(True), b.float().requires_grad_(True)
    yf = x.float() @ wf.t() + bf
    (torch.nn.functional.layer_norm(yf + res.float(), (N,), gamma, beta, 1e-12) * wts).sum().backward()
    tol = 2e-2 * float(bf.grad.abs().max()) + 1e-2
    assert torch.allclose(gb1, gb0, atol=tol, rtol=2e-2)
    assert torch.allclose(gb1, bf.grad, atol=3 * tol, rtol=5e-2)
    assert torch.equal(gw1, gw0)
    assert torch.allclose(gb2, bf.grad + 0.5 * rows, atol=3 * tol + 0.02 * rows, rtol=5e-2)


@pytest.mark.parametrize("B,S,nh,hd", [(2, 128, 4, 64), (3, 512, 2, 64), (1, 64, 3, 32), (2, 72, 2, 64)])
@pytest.mark.parametrize("masked", [False, True])
def test_attention_matches_reference(B, S, nh, hd, masked):
    from distributed_tensorflow_amd.ops.attention import self_attention
    torch.manual_seed(S + nh)
    qkv = (torch.randn(B, S, 3, nh, hd, device=DEV) * 0.7).to(torch.bfloat16)
    bias = None
    if masked:
        bias = torch.zeros(B, S, device=DEV)
        bias[:, S - 9:] = float("-inf")
    q1 = qkv.clone().requires_grad_(True)
    out = self_attention(q1, bias)
    do = torch.randn_like(out)
    out.backward(do)
    qr = qkv.float().requires_grad_(True)
    q, k, v = (qr[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    s = q @ k.transpose(-1, -2) / math.sqrt(hd)
    if masked:
        s = s + bias[:, None, None, :]
    ref = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3)
    ref.backward(do.float())
    torch.testing.assert_close(out.float(), ref.detach(), rtol=3e-2, atol=3e-2)
    torch.testing.assert_close(q1.grad.float(), qr.grad, rtol=5e-2, atol=5e-2)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_embedding_gather_scatter_and_sparse_path(dt):
    from distributed_tensorflow_amd.ops import embedding as E
    torch.manual_seed(5)
    V, D, B, S = 1000, 64, 6, 16
    table = torch.randn(V, D, device=DEV).to(dt).requires_grad_(True)
    pos = torch.randn(32, D, device=DEV).to(dt).requires_grad_(True)
    typ = torch.randn(2, D, device=DEV).to(dt).requires_grad_(True)
    ids = torch.randint(0, V, (B, S), device=DEV)
    ids[0, :4] = 7                                  # repeated ids exercise accumulation
    tt = torch.randint(0, 2, (B, S), device=DEV)
    out = E.embedding(table, ids, pos, typ, tt)
    ref = table.detach().float()[ids] + pos.detach().float()[:S][None] + typ.detach().float()[tt]
    torch.testing.assert_close(out.float(), ref, rtol=1e-2, atol=2e-2)
    g = torch.randn_like(out)
    out.backward(g)
    dref = torch.zeros(V, D, device=DEV).index_add_(0, ids.reshape(-1), g.float().reshape(-1, D))
    torch.testing.assert_close(table.grad.float(), dref, rtol=2e-2, atol=3e-2)
    torch.testing.assert_close(pos.grad.float()[:S], g.float().sum(0), rtol=2e-2, atol=5e-2)
    assert float(pos.grad.float()[S:].abs().sum()) == 0
    dtyp = torch.zeros(2, D, device=DEV).index_add_(0, tt.reshape(-1), g.float().reshape(-1, D))
    torch.testing.assert_close(typ.grad.float(), dtyp, rtol=2e-2, atol=2e-1)
    # sparse path: unique ids + segment sum == dense scatter-add restricted to the touched rows
    uniq, perm, seg = E.unique_segments(ids)
    gs = E.segment_sum(g.reshape(-1, D), perm, seg)
    torch.testing.assert_close(gs, dref[uniq], rtol=1e-3, atol=1e-3)
    # shard-local gather: ids outside the shard give zeros
    half = E._gather(table.detach()[500:], ids, row_base=500)
    exp = torch.where((ids >= 500)[..., None], table.detach().float()[ids], torch.zeros((), device=DEV)).reshape(-1, D)
    torch.testing.assert_close(half.float(), exp, rtol=1e-2, atol=1e-2)
    # sparse optimizer kernels vs the CPU formulas
    for kind in (E.SGD, E.ADAGRAD, E.ADAM):
        tab = torch.randn(V, D, device=DEV); m = torch.zeros_like(tab); v = torch.zeros_like(tab)
        tc, mc, vc = tab.cpu().clone(), m.cpu().clone(), v.cpu().clone()
        hyper = torch.tensor([0.1, 0.1, 0.001, 1e-7, 0.9, 0.999, 0.5, 0.0], device=DEV)
        E.sparse_apply(tab, m, v, gs, uniq, hyper, kind)
        E.sparse_apply(tc, mc, vc, gs.cpu(), uniq.cpu(), hyper.cpu(), kind)
        torch.testing.assert_close(tab.cpu(), tc, rtol=1e-4, atol=1e-5)


def test_bert_tiny_trains_on_gpu():
    from distributed_tensorflow_amd.engine.optimizer import Adam, FlatOptimizer
    from distributed_tensorflow_amd.engine.replica import Replica, TrainConfig
    from distributed_tensorflow_amd.models.bert import bert_tiny, synthetic_mlm_batch
    torch.manual_seed(0)
    rep = Replica(bert_tiny(), DEV, TrainConfig(param_dtype=torch.bfloat16))
    opt = FlatOptimizer(rep.flat, Adam(learning_rate=1e-3))
    x, y = synthetic_mlm_batch(8, 64, 1024, DEV, n_pred=8)
    losses = []
    for _ in range(8):
        losses.append(float(rep.forward_backward(x, y))); rep.finish_backward(); opt.step()
    assert all(l == l for l in losses) and losses[-1] < losses[0] - 0.3


# ------------------------------------------------------------------ fused Philox dropout
@pytest.mark.gpu
@pytest.mark.parametrize("activation", [None, "relu", "gelu"])
def test_dense_fused_dropout_matches_philox_reference(activation):
    """The GEMM-epilogue dropout mask equals the NumPy Philox reference, forward and backward."""
    import numpy as np
    from distributed_tensorflow_amd.ops import dense, rng

    dev = torch.device("cuda:0")
    torch.manual_seed(1234)
    M, K, N, p = 200, 64, 136, 0.3
    x = torch.randn(M, K, device=dev).bfloat16().requires_grad_(True)
    w = (torch.randn(N, K, device=dev) * 0.2).bfloat16().requires_grad_(True)
    b = torch.randn(N, device=dev).bfloat16().requires_grad_(True)
    st = rng.state_for(dev)
    off0 = int(st.offset.item())
    y = dense.linear(x, w, b, activation=activation, dropout_p=p, training=True)
    gy = torch.randn_like(y)
    y.backward(gy)
    assert int(st.offset.item()) == off0 + (M * N + 3) // 4          # the device counter advanced
    keep = torch.from_numpy(rng.dropout_keep_mask_reference(st.seed, off0, M * N, p)).view(M, N).to(dev)
    assert abs(float(keep.float().mean()) - (1 - p)) < 0.02
    # reference with the same mask in fp32
    xr, wr, br = (t.detach().float().requires_grad_(True) for t in (x, w, b))
    z = xr @ wr.t() + br
    a = torch.relu(z) if activation == "relu" else (torch.nn.functional.gelu(z) if activation == "gelu" else z)
    yr = a * keep / (1 - p)
    yr.backward(gy.float())
    assert torch.equal((y != 0), (yr.bfloat16() != 0)) or float(((y != 0) != (yr != 0)).float().mean()) < 2e-3
    assert torch.allclose(y.float(), yr.detach(), atol=0.06, rtol=0.03)
    assert torch.allclose(x.grad.float(), xr.grad, atol=0.15, rtol=0.05)
    assert torch.allclose(w.grad.float(), wr.grad, atol=0.5, rtol=0.05)
    assert torch.allclose(b.grad.float(), br.grad, atol=0.3, rtol=0.05)
    # eval mode: no dropout, counter untouched
    y2 = dense.linear(x, w, b, activation=activation, dropout_p=p, training=False)
    assert int(st.offset.item()) == off0 + (M * N + 3) // 4
    assert float((y2 == 0).float().mean()) < (0.6 if activation == "relu" else 0.01)


@pytest.mark.gpu
def test_sequential_fuses_dropout_and_graph_replays_draw_fresh_masks():
    from distributed_tensorflow_amd.api.layers import Dense, Dropout, Flatten
    from distributed_tensorflow_amd.api.model import Sequential

    dev = torch.device("cuda:0")
    torch.manual_seed(7)
    m = Sequential([Flatten(input_shape=(16, 16)), Dense(256, activation="relu"), Dropout(0.5), Dense(16)]).to(dev)
    for prm in m.parameters():
        prm.data = prm.data.bfloat16()
    x = torch.rand(64, 16, 16, device=dev).bfloat16()
    m.train()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.no_grad():
        m(x)
    torch.cuda.current_stream().wait_stream(s)
    with torch.no_grad(), torch.cuda.graph(g):
        out = m(x)
    g.replay(); a = out.clone()
    g.replay(); b = out.clone()
    assert not torch.equal(a, b)                 # the device-side offset advanced between replays
    m.eval()
    with torch.no_grad():
        assert torch.equal(m(x), m(x))


@pytest.mark.gpu
@pytest.mark.parametrize("stages", [1, 2])
def test_gemm_single_and_double_buffered_loops_agree(stages):
    from distributed_tensorflow_amd.ops import _lib, gemm

    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    lib = _lib.kernels()
    lib.dtf_gemm_set_stages(stages)
    try:
        for (M, N, K) in [(304, 136, 64), (1000, 256, 200), (520, 72, 1032)]:
            a = torch.randn(M, K, device=dev).bfloat16()
            b = torch.randn(N, K, device=dev).bfloat16()
            ref = a.float() @ b.float().t()
            for out in (gemm.gemm_nt(a, b), gemm.gemm_nn(a, b.t().contiguous()),
                        gemm.gemm_tn(a.t().contiguous(), b.t().contiguous())):
                assert torch.allclose(out.float(), ref, atol=0.5, rtol=2e-2), (stages, M, N, K)
    finally:
        lib.dtf_gemm_set_stages(0)


@pytest.mark.gpu
def test_conv_fork_accumulates_branch_gradient_in_epilogue():
    from distributed_tensorflow_amd.ops import conv
    torch.manual_seed(5)
    x = torch.randn(4, 64, 14, 14, device=DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(32, 64, 1, 1, device=DEV) * 0.1).bfloat16().contiguous(memory_format=torch.channels_last)
    res = {}
    for fork in (True, False):
        x1, w1 = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
        if fork:
            y, xi = conv.conv2d_fork(x1, w1, 1, 0)
        else:
            y, xi = conv.conv2d(x1, w1, 1, 0), x1
        out = y.float().square().sum() + (xi.float() * 3.0).sin().sum()
        out.backward()
        res[fork] = (x1.grad.float(), w1.grad.float())
    ref_scale = float(res[False][0].abs().max())
    assert torch.allclose(res[True][0], res[False][0], atol=0.02 * ref_scale, rtol=2e-2)
    assert torch.allclose(res[True][1], res[False][1], atol=0.02 * float(res[False][1].abs().max()), rtol=2e-2)
    # unused convolution branch: only the alias gradient flows
    x1 = x.clone().requires_grad_(True)
    _, xi = conv.conv2d_fork(x1, w, 1, 0)
    xi.float().sum().backward()
    assert torch.equal(x1.grad, torch.ones_like(x1))


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [(2, 64, 112, 112, 3, 2, 1), (3, 16, 28, 28, 2, 2, 0), (2, 8, 13, 11, 3, 1, 1),
                                 (1, 24, 9, 10, 3, 2, 0)])
def test_max_pool_matches_torch(cfg):
    from distributed_tensorflow_amd.ops import pool
    N, C, H, W, k, s, p = cfg
    torch.manual_seed(sum(cfg))
    # relu-like data: many exact ties (zeros), where the first-maximum convention decides the gradient
    x = torch.relu(torch.randn(N, C, H, W, device=DEV)).bfloat16().contiguous(memory_format=torch.channels_last)
    x1, x2 = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    y = pool.max_pool2d(x1, k, s, p)
    yr = torch.nn.functional.max_pool2d(x2, k, s, p)
    assert y.shape == yr.shape and torch.equal(y, yr)
    gy = torch.randn_like(yr)
    y.backward(gy)
    yr.backward(gy)
    assert torch.allclose(x1.grad.float(), x2.grad.float(), atol=2e-2, rtol=2e-2)
    # windows that overlap (k > s) add gradients in fp32 in the own kernel: total mass is preserved
    assert abs(float(x1.grad.float().sum()) - float(gy.float().sum())) < 0.02 * float(gy.float().abs().sum()) ** 0.5 + 1.0


@pytest.mark.gpu
def test_lds_dma_main_loop_matches_the_register_staged_one():
    """The LDS-DMA main loop of the 128x128 kernel (dtf_gemm_set_stages(3)): every layout, ragged edges, and the
    convolution gathers (padding and tile edges arrive as zeros written by out-of-range lanes)."""
    from distributed_tensorflow_amd.ops import _lib, conv
    from distributed_tensorflow_amd.ops import gemm as G
    lib = _lib.kernels()
    torch.manual_seed(21)
    M, N, K = 1000, 264, 328                  # nothing a multiple of the tile
    a = torch.randn(M, K, device=DEV).bfloat16()
    w_nt = (torch.randn(N, K, device=DEV) * 0.1).bfloat16()
    w_nn = w_nt.t().contiguous()
    gy = torch.randn(M, N, device=DEV).bfloat16()
    x = torch.randn(3, 136, 13, 13, device=DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(144, 136, 3, 3, device=DEV) * 0.05).bfloat16().contiguous(memory_format=torch.channels_last)

    def run_all():
        out = {}
        out["nt"] = G.gemm_raw(G.NT, a, w_nt, torch.empty(M, N, device=DEV, dtype=torch.bfloat16), M, N, K, K, K, N)
        out["nn"] = G.gemm_raw(G.NN, a, w_nn, torch.empty(M, N, device=DEV, dtype=torch.bfloat16), M, N, K, K, N, N)
        out["tn"] = G.gemm_raw(G.TN, gy, a, torch.empty(N, K, device=DEV, dtype=torch.bfloat16), N, K, M, N, K, K,
                               splits=1)
        x1 = x.clone().requires_grad_(True)
        y = conv.conv2d(x1, w, 1, 1)
        y.backward(torch.ones_like(y))
        out["conv"], out["conv_dx"] = y.detach(), x1.grad
        return out
    try:
        lib.dtf_gemm_set_stages(1)
        ref = run_all()
        lib.dtf_gemm_set_stages(3)
        got = run_all()
    finally:
        lib.dtf_gemm_set_stages(0)
    for k in ref:
        assert torch.equal(ref[k], got[k]), k           # same products in the same order: bit-identical
    assert torch.allclose(got["nt"].float(), a.float() @ w_nt.float().t(), atol=0.15, rtol=3e-2)


@pytest.mark.gpu
def test_stem_norm_pool_takes_the_statistics_of_the_first_convolution():
    """conv(7x7 / 2, 3 channels) -> norm + ReLU + max pool: the convolution's store phase reduces the norm's statistics;
    the result must match the path where the norm reduces them itself (and fp32 torch)."""
    from distributed_tensorflow_amd.ops import conv, norm
    torch.manual_seed(5)
    N, H = 6, 64
    x = torch.randn(N, 3, H, H, device=DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(64, 3, 7, 7, device=DEV) * 0.1).bfloat16().contiguous(memory_format=torch.channels_last)
    gamma, beta = torch.rand(64, device=DEV) + 0.5, torch.randn(64, device=DEV) * 0.1
    outs = []
    for use in (True, False):
        rm, rv = torch.zeros(64, device=DEV), torch.ones(64, device=DEV)
        z, st = conv.conv2d(x, w, 2, 3, bn_stats=True)
        assert st is not None
        y = norm.batch_norm_relu_max_pool(z, gamma, beta, rm, rv, True, 0.1, 1e-5, 3, 2, 1, st if use else None)
        outs.append((y.float(), rm.clone(), rv.clone()))
    (ya, ma, va), (yb, mb, vb) = outs
    assert torch.allclose(ma, mb, atol=1e-4, rtol=1e-4) and torch.allclose(va, vb, atol=1e-4, rtol=1e-3)
    assert float((ya - yb).abs().max()) <= 0.04            # a bf16 step at the activation's magnitude
    zr = torch.nn.functional.conv2d(x.float(), w.float(), None, 2, 3)
    ref = torch.nn.functional.max_pool2d(torch.relu(torch.nn.functional.batch_norm(
        zr, None, None, gamma, beta, True, 0.1, 1e-5)), 3, 2, 1)
    assert torch.allclose(ya, ref, atol=8e-2, rtol=5e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [(300, 64, 1002), (1000, 768, 3054), (64, 128, 67), (2048, 768, 30522)])
def test_linear_unaligned_width_runs_on_the_bf16_kernel_with_padded_rows(cfg):
    """A bf16 layer whose width is not a multiple of 8 (BERT's vocabulary projection) followed by the fused loss:
    logits and their gradient stay in row-padded buffers; values and all gradients against an fp32 evaluation."""
    from distributed_tensorflow_amd.ops import dense
    from distributed_tensorflow_amd.ops.softmax_ce import softmax_cross_entropy
    M, K, N = cfg
    torch.manual_seed(M + N)
    x = (torch.randn(M, K, device=DEV) * 0.5).bfloat16().requires_grad_(True)
    w = (torch.randn(N, K, device=DEV) * 0.05).bfloat16().requires_grad_(True)
    b = (torch.randn(N, device=DEV) * 0.1).requires_grad_(True)
    y = torch.randint(0, N, (M,), device=DEV)
    logits = dense.linear(x, w, b)
    assert logits.shape == (M, N) and logits.stride(0) % 8 == 0 and logits.stride(0) >= N     # padded rows
    loss = softmax_cross_entropy(logits, y, reduction="mean")
    loss.backward()
    xr, wr, br = (t.detach().float().requires_grad_(True) for t in (x, w, b))
    lr = torch.nn.functional.cross_entropy(xr @ wr.t() + br, y)
    lr.backward()
    assert torch.allclose(logits.float(), (xr @ wr.t() + br).detach(), atol=3e-2, rtol=2e-2)
    assert abs(float(loss.detach()) - float(lr.detach())) < 2e-2
    for got, ref in ((x.grad, xr.grad), (w.grad, wr.grad), (b.grad, br.grad)):
        scale = float(ref.abs().max()) + 1e-12
        assert float((got.float() - ref).abs().max()) < 3e-2 * scale
    # a gradient that does not come in the padded layout (here: a plain sum) is copied into it
    x2 = x.detach().clone().requires_grad_(True)
    dense.linear(x2, w.detach(), None).float().sum().backward()
    ref = w.detach().float().sum(0)[None, :].expand(M, K)
    assert torch.allclose(x2.grad.float(), ref, atol=0.05 * float(ref.abs().max()), rtol=3e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [(5, 2048, 7, 7), (3, 64, 8, 8), (2, 24, 1, 1), (130, 8, 3, 5)])
def test_global_avg_pool_matches_fp32(cfg):
    from distributed_tensorflow_amd.ops import pool
    N, C, H, W = cfg
    torch.manual_seed(sum(cfg))
    x = torch.randn(N, C, H, W, device=DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    x1 = x.clone().requires_grad_(True)
    y = pool.global_avg_pool(x1)
    ref = x.float().mean(dim=(2, 3))
    assert y.shape == (N, C) and y.dtype == torch.bfloat16
    assert torch.allclose(y.float(), ref, atol=4e-3, rtol=8e-3)
    gy = torch.randn(N, C, device=DEV).bfloat16()
    y.backward(gy)
    gref = (gy.float() / (H * W))[:, :, None, None].expand(N, C, H, W)
    assert x1.grad.shape == x.shape and x1.grad.is_contiguous(memory_format=torch.channels_last)
    assert torch.allclose(x1.grad.float(), gref, atol=1e-6, rtol=8e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float32])
def test_embedding_table_grad_sorted_runs(dt):
    """Long runs of one id (cut into pieces, combined with atomics) and singletons (plain stores)."""
    from distributed_tensorflow_amd.ops import embedding as E
    torch.manual_seed(11)
    n, D, V = 5000, 64, 300
    ids = torch.randint(0, V, (n,), device=DEV)
    ids[:2000] = 3                        # a padding-like id: one run of > 2000 rows
    ids[2000:2300] = V - 1                # the last row of the table, run crosses a cut
    ids = ids[torch.randperm(n, device=DEV)]
    g = torch.randn(n, D, device=DEV).to(dt)
    out = E._table_grad(g, ids, V)
    ref = torch.zeros(V, D, device=DEV, dtype=torch.float64).index_add_(0, ids, g.double())
    torch.testing.assert_close(out.double(), ref, rtol=1e-3, atol=2e-2 if dt == torch.bfloat16 else 1e-3)
    # shard-local: only rows [100, 200) are kept
    out2 = E._table_grad(g, ids, 100, row_base=100)
    torch.testing.assert_close(out2.double(), ref[100:200], rtol=1e-3, atol=2e-2 if dt == torch.bfloat16 else 1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(256, 256, 64), (512, 768, 768), (1000, 520, 128), (300, 264, 192),
                                   (2048, 3072, 768), (4096, 256, 1024), (257, 8, 64)])
@pytest.mark.parametrize("act", [None, "gelu"])
def test_gemm256_deep_pipeline_matches_reference(shape, act):
    """LDS-DMA pipeline with counted waits: every K-tile parity, edge tiles, bias + activation."""
    from distributed_tensorflow_amd.ops import gemm
    M, N, K = shape
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K, device=DEV).bfloat16()
    b = torch.randn(N, K, device=DEV).bfloat16()
    bias = torch.randn(N, device=DEV)
    out = gemm.gemm256_nt(a, b, bias, act)
    ref = a.float() @ b.float().t() + bias
    if act == "gelu":
        ref = torch.nn.functional.gelu(ref)
    scale = max(1.0, float(ref.abs().max()))
    assert torch.allclose(out.float() / scale, ref / scale, atol=1e-2, rtol=2e-2), float((out.float() - ref).abs().max())
    # identical to itself across repeated launches (a race in the pipeline shows up as flicker)
    for _ in range(3):
        assert torch.equal(out, gemm.gemm256_nt(a, b, bias, act))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(16640, 1280, 256), (20008, 1032, 192), (70000, 264, 128), (70000, 256, 64),
                                   (33000, 768, 768)])
def test_gemm256_several_tiles_per_workgroup(shape):
    """More output tiles than CUs: a workgroup walks several tiles, the next tile's operands are requested before the
    current tile's stores and those stores stay in flight under the next main loop (full tiles) or are drained (edge
    tiles). Every epilogue, both operand orientations, even / odd / one / two K-tiles; repeated launches must agree
    bit for bit (a wait that counts wrongly shows up as flicker)."""
    from distributed_tensorflow_amd.ops import gemm
    M, N, K = shape
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K, device=DEV).bfloat16()
    b = torch.randn(N, K, device=DEV).bfloat16()
    bt = b.t().contiguous()
    bias = torch.randn(N, device=DEV)
    cin = torch.randn(M, N, device=DEV).bfloat16()
    prod = a.float() @ b.float().t()
    scale = max(1.0, float(prod.abs().max()))

    def close(out, ref):
        assert torch.allclose(out.float() / scale, ref / scale, atol=1e-2, rtol=2e-2), float((out.float() - ref).abs().max())

    out = gemm.gemm256_nt(a, b, bias, "relu")
    close(out, torch.relu(prod + bias))
    for _ in range(2):
        assert torch.equal(out, gemm.gemm256_nt(a, b, bias, "relu"))
    out = gemm.gemm256(gemm.NN, a, bt, M, N, K)
    close(out, prod)
    assert torch.equal(out, gemm.gemm256(gemm.NN, a, bt, M, N, K))
    out = gemm.gemm256(gemm.NN, a, bt, M, N, K, add=cin)
    close(out, prod.bfloat16().float() + cin.float())
    assert torch.equal(out, gemm.gemm256(gemm.NN, a, bt, M, N, K, add=cin))
    out = gemm.gemm256(gemm.NN, a, bt, M, N, K, mul=cin)
    close(out, prod.bfloat16().float() * cin.float())
    assert torch.equal(out, gemm.gemm256(gemm.NN, a, bt, M, N, K, mul=cin))
    der = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    out = gemm.gemm256(gemm.NT, a, b, M, N, K, bias, "gelu", grad_out=der)
    pre = (prod + bias).requires_grad_(True)
    ref = torch.nn.functional.gelu(pre)
    close(out, ref.detach())
    (dref,) = torch.autograd.grad(ref.sum(), pre)
    assert torch.allclose(der.float(), dref, atol=2e-2, rtol=2e-2)
    der2 = torch.empty_like(der)
    assert torch.equal(out, gemm.gemm256(gemm.NT, a, b, M, N, K, bias, "gelu", grad_out=der2)) and torch.equal(der, der2)


@pytest.mark.gpu
def test_gemm256_operand_orientation():
    from distributed_tensorflow_amd.ops import gemm
    M = N = K = 256
    a = torch.eye(M, K, device=DEV).bfloat16()
    b = torch.randn(N, K, device=DEV).bfloat16()          # asymmetric: A = I gives C = B^T
    out = gemm.gemm256_nt(a, b)
    assert torch.equal(out, b.t().contiguous())


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(256, 256, 64), (512, 768, 768), (1000, 520, 128), (304, 264, 192), (2048, 3072, 768)])
def test_gemm256_nn_and_tn_layouts(shape):
    from distributed_tensorflow_amd.ops import gemm
    M, N, K = shape
    torch.manual_seed(M * 3 + N + K)
    a = torch.randn(M, K, device=DEV).bfloat16()
    bt = torch.randn(K, N, device=DEV).bfloat16()            # k-major B
    ref = a.float() @ bt.float()
    scale = max(1.0, float(ref.abs().max()))
    out = gemm.gemm256(gemm.NN, a, bt, M, N, K)
    assert torch.allclose(out.float() / scale, ref / scale, atol=1e-2, rtol=2e-2), float((out.float() - ref).abs().max())
    at = a.t().contiguous()                                   # k-major A: [K, M]
    for splits in (1, 3):
        out = gemm.gemm256(gemm.TN, at, bt, M, N, K, splits=splits)
        assert torch.allclose(out.float() / scale, ref / scale, atol=1e-2, rtol=2e-2), (splits, float((out.float() - ref).abs().max()))
        assert torch.equal(out, gemm.gemm256(gemm.TN, at, bt, M, N, K, splits=splits))


@pytest.mark.gpu
def test_gemm256_weight_gradient_shape_with_split_k():
    from distributed_tensorflow_amd.ops import gemm
    torch.manual_seed(9)
    T, N, K = 8192, 768, 512                                   # dW[N, K] = dY[T, N]^T @ X[T, K]
    dy = torch.randn(T, N, device=DEV).bfloat16()
    x = torch.randn(T, K, device=DEV).bfloat16()
    out = gemm.gemm_tn(dy, x)
    ref = dy.float().t() @ x.float()
    assert torch.allclose(out.float(), ref, atol=0.02 * float(ref.abs().max()), rtol=2e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("B,S,nh,masked", [(2, 128, 2, False), (2, 256, 3, True), (1, 512, 12, True)])
def test_flash_attention_matches_fp32_reference(B, S, nh, masked):
    from distributed_tensorflow_amd.ops import attention as A
    hd = 64
    torch.manual_seed(B * 1000 + S + nh)
    qkv = (torch.randn(B, S, 3, nh, hd, device=DEV) * 0.7).bfloat16()
    bias = None
    if masked:
        bias = torch.zeros(B, S, device=DEV)
        bias[:, S - 37:] = float("-inf")          # padded tail
        bias[0, 5] = float("-inf")
    dout = torch.randn(B, S, nh, hd, device=DEV).bfloat16()

    def run(flash):
        A.set_flash(flash)
        x = qkv.clone().requires_grad_(True)
        o = A.self_attention(x, bias)
        o.backward(dout)
        return o.float(), x.grad.float()

    try:
        of, gf = run(True)
        ou, gu = run(False)
    finally:
        A.set_flash(True)
    # fp32 reference
    x = qkv.float().requires_grad_(True)
    q, k, v = (x[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    s = q @ k.transpose(-1, -2) / math.sqrt(hd)
    if bias is not None:
        s = s + bias[:, None, None, :]
    ref = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3)
    ref.backward(dout.float())
    assert torch.allclose(of, ref, atol=2e-2, rtol=2e-2), float((of - ref).abs().max())
    gs = float(x.grad.abs().max())
    assert float((gf - x.grad).abs().max()) < 0.03 * gs + 2e-2, (float((gf - x.grad).abs().max()), gs)
    # at least as close to fp32 as the unfused bf16 path is, within a factor
    assert float((gf - x.grad).abs().mean()) <= 2.0 * float((gu - x.grad).abs().mean()) + 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("M", [96, 4096])        # 128x128 kernel / 256x256 kernel data-gradient paths
def test_linear_fork_accumulates_residual_gradient(M):
    from distributed_tensorflow_amd.ops import dense
    torch.manual_seed(M)
    K, N = 512, 768
    x = torch.randn(M, K, device=DEV).bfloat16()
    w = (torch.randn(N, K, device=DEV) * 0.05).bfloat16()
    b = torch.randn(N, device=DEV).bfloat16()
    res = {}
    for fork in (True, False):
        x1, w1, b1 = (t.clone().requires_grad_(True) for t in (x, w, b))
        if fork:
            y, xi = dense.linear_fork(x1, w1, b1, "gelu")
        else:
            y, xi = dense.linear(x1, w1, b1, "gelu"), x1
        (y.float().square().sum() * 1e-2 + (xi.float() * 2.0).cos().sum()).backward()
        res[fork] = [t.grad.float() for t in (x1, w1, b1)]
    for a, r in zip(res[True], res[False]):
        assert torch.allclose(a, r, atol=0.03 * float(r.abs().max()) + 1e-3, rtol=3e-2)


@pytest.mark.parametrize("M", [4096, 8192])
def test_ffn_fork_fused_gelu_epilogues_match_fp32_reference(M):
    """The transformer feed-forward block with GELU and its derivative inside the 256x256 GEMM epilogues."""
    from distributed_tensorflow_amd.ops import dense, gemm
    torch.manual_seed(M + 1)
    H, Fd = 512, 4096
    assert gemm.big_ok(M, Fd, H)
    x = torch.randn(M, H, device=DEV).bfloat16()
    w1 = (torch.randn(Fd, H, device=DEV) * 0.05).bfloat16()
    b1 = (torch.randn(Fd, device=DEV) * 0.1).bfloat16()
    w2 = (torch.randn(H, Fd, device=DEV) * 0.03).bfloat16()
    b2 = (torch.randn(H, device=DEV) * 0.1).bfloat16()
    ts = [t.clone().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    y, xi = dense.ffn_fork(*ts)
    assert type(y.grad_fn).__name__ == "_FfnForkFnBackward"
    gy = torch.randn_like(y)
    ((y.float() * gy.float()).sum() + (xi.float() * 2.0).cos().sum()).backward()
    rs = [t.float().clone().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    pre = rs[0] @ rs[1].t() + rs[2]
    yr = torch.nn.functional.gelu(pre) @ rs[3].t() + rs[4]
    ((yr * gy.float()).sum() + (rs[0] * 2.0).cos().sum()).backward()
    assert torch.allclose(y.float(), yr.detach(), atol=0.02 * float(yr.detach().abs().max()), rtol=3e-2)
    for a, r in zip(ts, rs):
        assert torch.allclose(a.grad.float(), r.grad, atol=0.03 * float(r.grad.abs().max()) + 1e-3, rtol=5e-2), \
            float((a.grad.float() - r.grad).abs().max())



@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [(4, 64, 56, 256, 1, 1, 0), (3, 128, 28, 128, 3, 1, 1), (5, 64, 30, 72, 3, 2, 1),
                                 (2, 256, 14, 1024, 1, 1, 0)])
def test_conv_hands_batch_norm_its_partial_statistics(cfg):
    """The GEMM store phase accumulates sum(y), sum(y^2) per channel; the batch norm that consumes them must give
    the result of the two-pass norm (same values, different summation order)."""
    from distributed_tensorflow_amd.ops import conv, norm
    N, C, H, K, k, stride, pad = cfg
    torch.manual_seed(N * C + K)
    x = torch.randn(N, C, H, H, device=DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(K, C, k, k, device=DEV) * (1.0 / (C * k * k) ** 0.5)).bfloat16().contiguous(
        memory_format=torch.channels_last)
    y, st = conv.conv2d(x, w, stride, pad, bn_stats=True)
    assert st is not None and st.shape[1:] == (2, K)
    yf = y.float().permute(0, 2, 3, 1).reshape(-1, K)
    tot = st.sum(0)
    assert torch.allclose(tot[0], yf.sum(0), rtol=1e-4, atol=1e-2 * float(yf.abs().max()))
    assert torch.allclose(tot[1], (yf * yf).sum(0), rtol=1e-4, atol=1e-2)
    gamma, beta = torch.rand(K, device=DEV) + 0.5, torch.randn(K, device=DEV)
    res = torch.randn_like(y)
    outs = []
    for partials in (None, st):
        rm, rv = torch.zeros(K, device=DEV), torch.ones(K, device=DEV)
        o = norm.batch_norm_act(y, gamma, beta, rm, rv, True, 0.1, 1e-5, True, res, partials=partials)
        outs.append((o.float(), rm, rv))
    assert torch.allclose(outs[0][0], outs[1][0], atol=2e-2, rtol=2e-2)
    assert torch.allclose(outs[0][1], outs[1][1], atol=1e-5, rtol=1e-4)
    assert torch.allclose(outs[0][2], outs[1][2], atol=1e-5, rtol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("block", ["bottleneck", "basic"])
def test_norm_backward_sums_from_the_consuming_convolution(block):
    """The data-gradient GEMM of the convolution after a batch norm reduces that norm's backward sums while it
    stores the gradient; every parameter gradient must match the path where the norm reduces on its own."""
    from distributed_tensorflow_amd.models import resnet as R
    from distributed_tensorflow_amd.ops import conv
    torch.manual_seed(11)
    if block == "bottleneck":
        net = torch.nn.Sequential(R.Bottleneck(64, 32, 1), R.Bottleneck(128, 32, 1), R.Bottleneck(128, 64, 2))
        x = torch.randn(6, 64, 28, 28, device=DEV)
    else:
        net = torch.nn.Sequential(R.BasicBlock(128, 128, 1), R.BasicBlock(128, 128, 1), R.BasicBlock(128, 256, 2))
        x = torch.randn(6, 128, 20, 20, device=DEV)
    net = net.to(DEV)
    for m in net.modules():
        if isinstance(m, R.Conv):
            m.weight.data = m.weight.data.bfloat16().contiguous(memory_format=torch.channels_last)
        if isinstance(m, R.BNAct):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.3)
    net.train()
    x = x.bfloat16().contiguous(memory_format=torch.channels_last)
    grads = {}
    with torch.no_grad():
        n_out = net(x).numel()
    # positive loss weights: the parameter gradients are sums without wholesale cancellation, so bf16 rounding of
    # the activations shows up as a relative error of about a percent instead of ten
    wts = torch.rand(n_out, device=DEV, generator=torch.Generator(device=DEV).manual_seed(3)) + 0.5
    try:
        for tag, fwd, grad in (("none", False, False), ("fwd", True, False), ("both", True, True)):
            conv.set_bn_stats(fwd, grad)
            for p in net.parameters():
                p.grad = None
            xi = x.clone().requires_grad_(True)
            out = net(xi)
            (out.float() * wts.reshape(out.shape)).sum().backward()
            grads[tag] = {n: p.grad.float().clone() for n, p in net.named_parameters()}
            grads[tag]["input"] = xi.grad.float().clone()
    finally:
        conv.set_bn_stats(True)

    # fp32 reference: the same blocks from torch's fp32 convolution / batch norm on the bf16-rounded weights
    def bn(m, t, res=None):
        y = torch.nn.functional.batch_norm(t, None, None, m.weight.float(), m.bias.float(), True, 0.1, m.eps)
        y = y.to(t.dtype)
        if res is not None:
            y = y + res
        return torch.relu(y) if m.relu else y

    def cv(m, t):
        return torch.nn.functional.conv2d(t, m.weight.to(t.dtype), None, m.stride, m.padding)

    def ref_forward(t):
        for blk in net:
            idt = t if blk.down is None else bn(blk.down_bn, cv(blk.down, t))
            y = bn(blk.bn1, cv(blk.conv1, t))
            if block == "bottleneck":
                y = bn(blk.bn2, cv(blk.conv2, y))
                t = bn(blk.bn3, cv(blk.conv3, y), idt)
            else:
                t = bn(blk.bn2, cv(blk.conv2, y), idt)
        return t

    def torch_grads(dtype):
        for p in net.parameters():
            p.grad = None
        xr = x.to(dtype).requires_grad_(True)
        outr = ref_forward(xr)
        (outr.float() * wts.reshape(outr.shape)).sum().backward()
        g = {n: p.grad.float().clone() for n, p in net.named_parameters()}
        g["input"] = xr.grad.float().clone()
        return g

    ref = torch_grads(torch.float32)
    stock = torch_grads(torch.bfloat16)      # torch's own bf16 evaluation: the yardstick for bf16 rounding

    def rel(a, r):
        return float((a - r).norm() / (r.norm() + 1e-12))
    print("worst relative L2 error against fp32:",
          {tag: max(rel(g[n], ref[n]) for n in ref) for tag, g in list(grads.items()) + [("torch-bf16", stock)]})
    for n in ref:
        # same forward, the norm's backward sums taken in the GEMM or by the norm: only the summation order differs
        assert rel(grads["both"][n], grads["fwd"][n]) < 5e-3, (n, rel(grads["both"][n], grads["fwd"][n]))
        # every path is a bf16 evaluation of the same function (gradients through batch norms cancel heavily, so
        # bf16 activations cost 10-20 % on the early layers, for torch's bf16 kernels as for these): no further from
        # the fp32 gradients than 1.6 x torch's bf16 evaluation of the same blocks plus two percent -- the two are
        # independent roundings, so their ratio scatters; a wrong partial sum is off by the size of the gradient
        for tag in grads:
            assert rel(grads[tag][n], ref[n]) < 1.6 * rel(stock[n], ref[n]) + 2e-2, \
                (tag, n, rel(grads[tag][n], ref[n]), rel(stock[n], ref[n]))


@pytest.mark.gpu
def test_fused_fold_and_finalize_matches_the_separate_kernels():
    """More than 1024 partial rows (M / 128 with M = 48 * 56 * 56): the single-launch fold + finalize (last block
    per channel group finishes the statistics) against the separate fold and finalize kernels, for the forward
    statistics handed over by the producing convolution and the backward sums handed over by the consuming one."""
    from distributed_tensorflow_amd.models import resnet as R
    from distributed_tensorflow_amd.ops import norm
    torch.manual_seed(5)
    net = torch.nn.Sequential(R.Bottleneck(64, 16, 1), R.Bottleneck(64, 16, 1)).to(DEV)
    for m in net.modules():
        if isinstance(m, R.Conv):
            m.weight.data = m.weight.data.bfloat16().contiguous(memory_format=torch.channels_last)
        if isinstance(m, R.BNAct):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.3)
    net.train()
    x = torch.randn(48, 64, 56, 56, device=DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    assert x.shape[0] * 56 * 56 // 128 > 1024
    wts = torch.rand(48 * 64 * 56 * 56, device=DEV) + 0.5
    res = {}
    saved = norm._fused_reduce
    try:
        for fused in (False, True):
            norm._fused_reduce = fused
            for p in net.parameters():
                p.grad = None
            for m in net.modules():
                if isinstance(m, R.BNAct):
                    m.running_mean.zero_()
                    m.running_var.fill_(1.0)
            xi = x.clone().requires_grad_(True)
            out = net(xi)
            (out.float() * wts.reshape(out.shape)).sum().backward()
            torch.cuda.synchronize()
            r = {n: p.grad.float().clone() for n, p in net.named_parameters()}
            r["input"], r["out"] = xi.grad.float().clone(), out.float().clone()
            for n, b in net.named_buffers():
                if b.is_floating_point():
                    r["buf." + n] = b.float().clone()
            res[fused] = r
    finally:
        norm._fused_reduce = saved
    assert norm._counters           # the fused kernel did run; its ring of counters is back at zero after every launch
    for buf, _ in norm._counters.values():
        assert int(buf.abs().sum()) == 0
    for n, ref in res[False].items():
        got = res[True][n]
        err = float((got - ref).norm() / (ref.norm() + 1e-12))
        # the same sums in another order: the statistics agree to fp32 rounding, parameter gradients (long sums)
        # closely; the input gradient has passed six norms' backward passes, each rounding to bf16, where a last-bit
        # change of a coefficient flips roundings (bf16 epsilon is 0.8 %)
        tol = 1e-5 if n.startswith("buf.") else 2e-2 if n in ("input", "out") else 5e-3
        assert err < tol, (n, err)
    # element by element: all but the few activations whose bf16 rounding (or ReLU, next to zero) flipped on the way
    close = torch.isclose(res[True]["out"], res[False]["out"], atol=3e-2, rtol=3e-2)
    assert float(close.float().mean()) > 0.999


@pytest.mark.gpu
def test_masked_residual_gradient_hand_over_and_its_guards():
    """norm-add-ReLU with a fork alias as residual: the branch gradient travels unmasked plus mask bits and the
    fork's data-gradient epilogue applies them; a second consumer of the alias must fail loudly."""
    from distributed_tensorflow_amd.ops import conv, norm
    torch.manual_seed(21)
    N, C, H = 4, 64, 16
    x = torch.randn(N, C, H, H, device=DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(C, C, 1, 1, device=DEV) * 0.1).bfloat16().contiguous(memory_format=torch.channels_last)
    gamma, beta = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV) * 0.2

    def run(hand_over, extra_use=False):
        conv.set_bn_stats(True, hand_over)
        xi = x.clone().requires_grad_(True)
        wi = w.clone().requires_grad_(True)
        y, alias = conv.conv2d_fork(xi, wi, 1, 0)
        rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
        out = norm.batch_norm_act(y, gamma, beta, rm, rv, True, 0.1, 1e-5, True, alias)
        loss = (out.float() * torch.linspace(0.5, 1.5, out.numel(), device=DEV).reshape(out.shape)).sum()
        if extra_use:
            loss = loss + alias.float().sum()
        loss.backward()
        return xi.grad.float(), wi.grad.float()
    try:
        a, b = run(True), run(False)
        for p, q in zip(a, b):
            assert torch.allclose(p, q, atol=0.01 * float(q.abs().max()) + 1e-3, rtol=2e-2)
        with pytest.raises(RuntimeError, match="only use"):
            run(True, extra_use=True)
    finally:
        conv.set_bn_stats(True)
    # the slow path: mask bits applied by a torch composition
    g = torch.randn(N, C, H, H, device=DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    keep = torch.rand(N, H, H, C, device=DEV) > 0.5
    bits = (keep.reshape(-1, 8).to(torch.uint8) << torch.arange(8, device=DEV, dtype=torch.uint8)).sum(1).to(torch.uint8)
    out = conv._apply_mask_bits(g, bits)
    assert torch.equal(out.float(), g.float() * keep.permute(0, 3, 1, 2))


@pytest.mark.gpu
def test_plain_norm_reads_the_residual_gradient_with_the_mask_bits():
    """Down-sampling branch: ``relu(bn_b(y) + bn_a(z))`` with ``bn_a`` a plain norm. With the hand-over ``bn_b``
    writes no masked copy of its incoming gradient; ``bn_a`` reads that gradient together with the mask bits. The
    gradients must match the path without the hand-over and fp32 autograd; a second use of ``bn_a``'s output fails
    loudly."""
    from distributed_tensorflow_amd.ops import conv, norm
    torch.manual_seed(33)
    N, C, H = 6, 64, 12
    y = torch.randn(N, C, H, H, device=DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    z = torch.randn(N, C, H, H, device=DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    ga, ba = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV) * 0.2
    gb, bb = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV) * 0.2
    wts = torch.linspace(0.5, 1.5, y.numel(), device=DEV).reshape(y.shape)

    def run(hand_over, extra_use=False):
        conv.set_bn_stats(True, hand_over)
        yi, zi = y.clone().requires_grad_(True), z.clone().requires_grad_(True)
        gai, gbi = ga.clone().requires_grad_(True), gb.clone().requires_grad_(True)
        st = lambda: (torch.zeros(C, device=DEV), torch.ones(C, device=DEV))
        idt = norm.batch_norm_act(zi, gai, ba, *st(), True, 0.1, 1e-5, False)
        assert (getattr(idt, norm._TOKEN_ATTR, None) is not None) == hand_over
        out = norm.batch_norm_act(yi, gbi, bb, *st(), True, 0.1, 1e-5, True, idt)
        loss = (out.float() * wts).sum()
        if extra_use:
            loss = loss + idt.float().sum()
        loss.backward()
        return [t.grad.float() for t in (yi, zi, gai, gbi)]

    def reference():
        yi, zi = y.float().requires_grad_(True), z.float().requires_grad_(True)
        gai, gbi = ga.clone().requires_grad_(True), gb.clone().requires_grad_(True)
        idt = torch.nn.functional.batch_norm(zi, None, None, gai, ba, True, 0.1, 1e-5)
        out = torch.relu(torch.nn.functional.batch_norm(yi, None, None, gbi, bb, True, 0.1, 1e-5) + idt)
        (out * wts).sum().backward()
        return [t.grad for t in (yi, zi, gai, gbi)]
    try:
        a, b, r = run(True), run(False), reference()
        for p, q, f in zip(a, b, r):
            assert torch.allclose(p, q, atol=0.01 * float(q.abs().max()) + 1e-3, rtol=2e-2)
            # against fp32: bf16 activations (a ReLU mask bit flips where the pre-activation rounds across zero)
            assert float((p - f).norm() / f.norm()) < 6e-2
        with pytest.raises(RuntimeError, match="only use"):
            run(True, extra_use=True)
    finally:
        conv.set_bn_stats(True)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [(4, 64, 56, 3, 2, 1), (3, 16, 17, 3, 2, 1), (2, 32, 12, 2, 2, 0)])
def test_norm_relu_max_pool_in_one_pass_matches_the_separate_kernels(cfg):
    from distributed_tensorflow_amd.ops import norm, pool
    N, C, H, k, s, p = cfg
    torch.manual_seed(N + C + H)
    x = torch.randn(N, C, H, H, device=DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    gamma = (torch.rand(C, device=DEV) + 0.5) * torch.where(torch.arange(C, device=DEV) % 5 == 0, -1.0, 1.0)
    beta = torch.randn(C, device=DEV) * 0.3
    res = {}
    for fused in (True, False):
        xi, gi, bi = (t.clone().requires_grad_(True) for t in (x, gamma, beta))
        rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
        if fused:
            y = norm.batch_norm_relu_max_pool(xi, gi, bi, rm, rv, True, 0.1, 1e-5, k, s, p)
            assert type(y.grad_fn).__name__ == "_BNReluPoolFnBackward" or y.grad_fn.next_functions
        else:
            y = pool.max_pool2d(norm.batch_norm_act(xi, gi, bi, rm, rv, True, 0.1, 1e-5, True, None), k, s, p)
        w = torch.linspace(0.5, 1.5, y.numel(), device=DEV).reshape(y.shape)
        (y.float() * w).sum().backward()
        res[fused] = (y.detach().float(), xi.grad.float(), gi.grad, bi.grad, rm, rv)
    a, b = res[True], res[False]
    assert torch.equal(a[0], b[0])                       # taps are compared after bf16 rounding, as unfused
    assert torch.allclose(a[1], b[1], atol=0.02 * float(b[1].abs().max()) + 1e-4, rtol=2e-2)
    for i in (2, 3, 4, 5):
        assert torch.allclose(a[i], b[i], atol=1e-3 * float(b[i].abs().max()) + 1e-5, rtol=1e-3), i


@pytest.mark.gpu
@pytest.mark.parametrize("second", ["conv", "add"])
def test_norm_output_with_two_consumers_falls_back_to_its_own_reduction(second):
    """A batch-norm output that feeds a convolution AND something else (a second convolution, a plain add): the
    gradient the norm receives is the sum over its consumers, so sums gathered by one convolution's data-gradient
    GEMM must not be used. Gradients are checked against fp32 autograd of the same graph."""
    from distributed_tensorflow_amd.ops import conv, norm
    torch.manual_seed(31)
    N, C, H = 4, 64, 12
    x = torch.randn(N, C, H, H, device=DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    g = (torch.rand(C, device=DEV) + 0.5)
    b = torch.randn(C, device=DEV) * 0.2
    w1 = (torch.randn(64, C, 3, 3, device=DEV) * 0.05).bfloat16().contiguous(memory_format=torch.channels_last)
    w2 = (torch.randn(64, C, 1, 1, device=DEV) * 0.1).bfloat16().contiguous(memory_format=torch.channels_last)
    wts = torch.rand(N, 64, H, H, device=DEV) + 0.5

    def run(fp32):
        xi = (x.float() if fp32 else x.clone()).requires_grad_(True)
        gi, bi = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
        a1 = (w1.float() if fp32 else w1.clone()).requires_grad_(True)
        a2 = (w2.float() if fp32 else w2.clone()).requires_grad_(True)
        if fp32:
            y = torch.relu(torch.nn.functional.batch_norm(xi, None, None, gi, bi, True, 0.1, 1e-5))
            o = torch.nn.functional.conv2d(y, a1, None, 1, 1)
            o = o + (torch.nn.functional.conv2d(y, a2) if second == "conv" else y)
        else:
            y = norm.batch_norm_act(xi, gi, bi, torch.zeros(C, device=DEV), torch.ones(C, device=DEV), True, 0.1, 1e-5,
                                    True)
            o = conv.conv2d(y, a1, 1, 1)
            o = o + (conv.conv2d(y, a2) if second == "conv" else y)
        (o.float() * wts).sum().backward()
        return [t.grad.float() for t in (xi, gi, bi, a1, a2 if second == "conv" else a1)]

    ref, got = run(True), run(False)
    for r, a, name in zip(ref, got, ("dx", "dgamma", "dbeta", "dw1", "dw2")):
        err = float((a - r).norm() / r.norm())
        assert err < 3e-2, (name, err)


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(100, 512, 784), (100, 10, 512), (7, 3, 5), (64, 64, 16), (257, 130, 1000), (33, 1000, 2048)])
@pytest.mark.parametrize("dts", [("f", "f"), ("b", "b"), ("f", "b")])
def test_general_gemm_matches_fp32_reference_in_every_layout(M, N, K, dts):
    """NT / NN / TN products through strided views (nothing is copied), fp32 and bf16 operands, unaligned sizes."""
    from distributed_tensorflow_amd.ops import gemm_any as G
    torch.manual_seed(M + N + K)
    dt = {"f": torch.float32, "b": torch.bfloat16}
    a = torch.randn(M, K, device=DEV).to(dt[dts[0]])
    w = (torch.randn(N, K, device=DEV) / math.sqrt(K)).to(dt[dts[1]])
    bias = torch.randn(N, device=DEV)
    ref = a.float() @ w.float().t()
    tol = dict(rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(G.matmul(a, w.t(), out_dtype=torch.float32), ref, **tol)                  # NT
    torch.testing.assert_close(G.matmul(a, w.t(), bias=bias, activation="relu", out_dtype=torch.float32),
                               torch.relu(ref + bias), **tol)
    g = torch.randn(M, N, device=DEV).to(dt[dts[0]])
    torch.testing.assert_close(G.matmul(g, w, out_dtype=torch.float32), g.float() @ w.float(), **tol)    # NN
    torch.testing.assert_close(G.matmul(g.t(), a, out_dtype=torch.float32), g.float().t() @ a.float(), **tol)   # TN
    if K >= 1000:   # the split-K path (slabs + ordered reduction) must agree with the single pass
        torch.testing.assert_close(G.matmul(a, w.t(), out_dtype=torch.float32, splits=5), ref, **tol)
    out_b = G.matmul(a, w.t(), bias=bias, out_dtype=torch.bfloat16)
    torch.testing.assert_close(out_b.float(), (ref + bias).bfloat16().float(), rtol=1e-2, atol=1e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_reference_mlp_layers_run_on_own_kernels_and_match_torch(dtype):
    """The reference model's two dense layers (784 -> 512 relu, dropout, 512 -> 10) in fp32 -- the launcher's default
    -- and in bf16 (where the 10-column layer is not 16-byte aligned): forward and gradients against fp32 torch."""
    from distributed_tensorflow_amd.ops import dense
    torch.manual_seed(3)
    x = torch.rand(100, 784, device=DEV).to(dtype)
    w1 = (torch.randn(512, 784, device=DEV) * 0.05).to(dtype).requires_grad_(True)
    b1 = torch.zeros(512, device=DEV, requires_grad=True)
    w2 = (torch.randn(10, 512, device=DEV) * 0.05).to(dtype).requires_grad_(True)
    b2 = torch.zeros(10, device=DEV, requires_grad=True)
    h = dense.linear(x, w1, b1, activation="relu")
    y = dense.linear(h, w2, b2)
    gy = torch.randn(100, 10, device=DEV)
    y.backward(gy.to(y.dtype))
    got = [y.float(), w1.grad.float(), b1.grad.float(), w2.grad.float(), b2.grad.float()]
    w1r, b1r = w1.detach().float().requires_grad_(True), b1.detach().clone().requires_grad_(True)
    w2r, b2r = w2.detach().float().requires_grad_(True), b2.detach().clone().requires_grad_(True)
    hr = torch.relu(x.float() @ w1r.t() + b1r)
    yr = hr @ w2r.t() + b2r
    yr.backward(gy.to(y.dtype).float())
    ref = [yr.detach(), w1r.grad, b1r.grad, w2r.grad, b2r.grad]
    tol = dict(rtol=1e-4, atol=1e-4) if dtype == torch.float32 else dict(rtol=3e-2, atol=3e-2)
    for a, r in zip(got, ref):
        scale = max(1.0, float(r.abs().max()))
        torch.testing.assert_close(a / scale, r / scale, **tol)


@pytest.mark.gpu
def test_fp32_dropout_in_dense_layer_regenerates_its_mask_in_backward():
    from distributed_tensorflow_amd.ops import dense, rng
    torch.manual_seed(9)
    x = torch.randn(64, 96, device=DEV)
    w = (torch.randn(48, 96, device=DEV) * 0.1).requires_grad_(True)
    st = rng.state_for(torch.device(DEV))
    off0 = int(st.offset.item())
    y = dense.linear(x, w, None, activation="relu", dropout_p=0.25, training=True)
    keep = torch.from_numpy(rng.dropout_keep_mask_reference(st.seed, off0, y.numel(), 0.25)).to(DEV).reshape(y.shape)
    ref = torch.relu(x @ w.detach().t()) * keep / 0.75
    torch.testing.assert_close(y, ref, rtol=1e-5, atol=1e-5)
    y.backward(torch.ones_like(y))
    gref = ((keep / 0.75) * (ref > 0)).t() @ x
    torch.testing.assert_close(w.grad, gref, rtol=1e-4, atol=1e-4)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_keras_style_cnn_layers_match_torch_reference(dtype):
    """Conv2D ('valid' and 'same', stride 1 and 2, bias + ReLU) -> MaxPooling2D -> Flatten -> Dense layers of the
    Keras-style API, forward and every parameter gradient against the same network written with fp32 torch ops."""
    import torch.nn.functional as F
    from distributed_tensorflow_amd.api.layers import Conv2D, Dense, Flatten, MaxPooling2D
    from distributed_tensorflow_amd.api.model import Sequential
    torch.manual_seed(17)
    net = Sequential([Conv2D(32, 3, padding="same", activation="relu", input_shape=(28, 28, 1)), MaxPooling2D(2),
                      Conv2D(64, 3, strides=2, padding="same", activation="relu"), Conv2D(16, 3, padding="same"),
                      Flatten(), Dense(40, activation="relu"), Dense(10, activation="softmax")]).to(DEV)
    for p in net.parameters():
        if p.dim() > 1:
            p.data = p.data.to(dtype)
            if p.dim() == 4:
                p.data = p.data.contiguous(memory_format=torch.channels_last)
    for p in net.parameters():
        if p.dim() == 1:
            p.data.normal_(0, 0.1)
    x = torch.rand(12, 28, 28, 1, device=DEV)
    out = net(x, logits=True)
    assert out.shape == (12, 10)
    gy = torch.randn(12, 10, device=DEV)
    out.float().backward(gy)
    got = [out.float()] + [p.grad.float() for p in net.parameters()]

    ps = [p.detach().float().requires_grad_(True) for p in net.parameters()]
    w1, b1, w2, b2, w3, b3, w4, b4, w5, b5 = ps
    h = x.to(dtype).float().permute(0, 3, 1, 2)
    h = F.max_pool2d(F.relu(F.conv2d(h, w1, b1, padding=1)), 2)           # 28 -> 14
    h = F.relu(F.conv2d(F.pad(h, (0, 1, 0, 1)), w2, b2, stride=2))         # 'same', stride 2 on 14: pad 0 / 1 -> 7
    h = F.conv2d(h, w3, b3, padding=1)
    h = h.permute(0, 2, 3, 1).reshape(12, -1)                              # Keras flattens in NHWC order
    h = F.relu(h @ w4.t() + b4)
    ref_out = h @ w5.t() + b5
    ref_out.backward(gy)
    ref = [ref_out.detach()] + [p.grad for p in ps]
    # bf16: five layers of bf16-rounded activations against an fp32 evaluation (each layer has its own tight test)
    tol = dict(rtol=2e-4, atol=2e-4) if dtype == torch.float32 else dict(rtol=1e-1, atol=1e-1)
    for a, r in zip(got, ref):
        scale = max(1.0, float(r.abs().max()))
        torch.testing.assert_close(a / scale, r / scale, **tol)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_standalone_dropout_layer_uses_the_philox_state(dtype):
    from distributed_tensorflow_amd.ops import dense, rng
    torch.manual_seed(4)
    x = torch.randn(6, 16, 5, 7, device=DEV).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    st = rng.state_for(torch.device(DEV))
    off0 = int(st.offset.item())
    y = dense.dropout(x, 0.3, True)
    assert y.shape == x.shape and y.is_contiguous(memory_format=torch.channels_last)
    keep = torch.from_numpy(rng.dropout_keep_mask_reference(st.seed, off0, x.numel(), 0.3)).to(DEV)
    keep = keep.reshape(6, 5, 7, 16).permute(0, 3, 1, 2)                   # the mask follows memory (NHWC) order
    ref = (x.detach().float() * keep / 0.7).to(dtype)
    torch.testing.assert_close(y.float(), ref.float(), rtol=1e-2, atol=1e-3)
    y.float().sum().backward()
    torch.testing.assert_close(x.grad.float(), (keep / 0.7).float().to(dtype).float(), rtol=1e-2, atol=1e-3)
    assert int(st.offset.item()) == off0 + (x.numel() + 3) // 4


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [
    # (N, C, H, W, K, k, stride, pad)
    (3, 128, 14, 14, 128, 3, 1, 1), (2, 128, 28, 28, 128, 3, 2, 1), (4, 64, 14, 14, 256, 1, 1, 0),
    (3, 64, 18, 18, 64, 3, 1, 1), (2, 64, 20, 20, 64, 3, 2, 1), (5, 256, 7, 7, 64, 1, 1, 0), (2, 72, 9, 11, 200, 3, 1, 1),
])
def test_deferred_norm_is_applied_by_the_consuming_convolution(cfg):
    """conv(DeferredNorm) == conv(relu(bn(x))): the operand transform computes the very bf16 values the norm's apply
    kernel would have written, so the output and the weight gradient agree bit for bit; the data gradient and the
    norm's own gradients agree up to the order of the norm's backward sums."""
    from distributed_tensorflow_amd.ops import conv, norm
    N, C, H, W, K, k, stride, pad = cfg
    torch.manual_seed(sum(cfg))
    x0 = (torch.randn(N, C, H, W, device=DEV) * 1.5 + 0.3).bfloat16().contiguous(memory_format=torch.channels_last)
    wp = (torch.randn(C, C, 1, 1, device=DEV) * 0.1).bfloat16().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(K, C, k, k, device=DEV) / math.sqrt(C * k * k)).bfloat16().contiguous(memory_format=torch.channels_last)
    gamma0, beta0 = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV) * 0.3
    res = {}
    norm.set_deferred(True)
    for mode in ("deferred", "materialized", "classic"):
        x = x0.clone().requires_grad_(True)
        w1 = w.clone().requires_grad_(True)
        g, b = gamma0.clone().requires_grad_(True), beta0.clone().requires_grad_(True)
        rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
        z, st = conv.conv2d(x, wp, 1, 0, bn_stats=True)            # producer: hands its statistics over
        if mode == "classic":
            a = norm.batch_norm_act(z, g, b, rm, rv, True, 0.1, 1e-5, True, None, st)
        else:
            a = norm.batch_norm_deferred(z, g, b, rm, rv, 0.1, 1e-5, True, st)
            assert isinstance(a, norm.DeferredNorm)
            if mode == "materialized":
                a = a.materialize()
        y = conv.conv2d(a, w1, stride, pad)
        gy = torch.randn(y.shape, device=DEV, generator=torch.Generator(device=DEV).manual_seed(2)).bfloat16()
        y.backward(gy)
        res[mode] = [y.detach().float(), x.grad.float(), w1.grad.float(), g.grad.float(), b.grad.float(), rm, rv]
    norm.set_deferred(False)
    for name, a, m_, c in zip(("y", "dx", "dw", "dgamma", "dbeta", "run_mean", "run_var"), *res.values()):
        if name in ("y", "dw", "run_mean", "run_var"):     # same bf16 operand values, same accumulation order
            assert torch.equal(a, m_), (name, "deferred vs materialized", float((a - m_).abs().max()))
        # the norm's backward sums come from the convolution's epilogue in one path and from the norm's own
        # reduction in the others: same value up to the summation order
        for other in (m_, c):
            scale = max(1.0, float(other.abs().max()))
            torch.testing.assert_close(a / scale, other / scale, rtol=1e-2, atol=1e-2, msg=lambda m: name + ": " + m)


# ------------------------------------------------------------------ general (any dtype / width) norm and pooling kernels
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(6, 3, 9, 7), (4, 10, 5, 5), (3, 64, 8, 8), (2, 100, 6, 4), (33, 20), (5000, 7, 3, 3)])
@pytest.mark.parametrize("relu,res", [(False, False), (True, False), (True, True)])
def test_general_batch_norm_matches_fp32_autograd(dtype, shape, relu, res):
    """Training-mode batch norm (+ residual, + ReLU) of fp32 / bf16 activations of any width on generic_nn.hip: output,
    all gradients and the running statistics against torch's fp32 batch norm on the same rounded inputs."""
    from distributed_tensorflow_amd.ops import generic_nn as G
    torch.manual_seed(sum(shape))
    C = shape[1]
    x = (torch.randn(shape, device=DEV) * 2 + 0.5).to(dtype)
    if len(shape) == 4:
        x = x.contiguous(memory_format=torch.channels_last)
    r = torch.randn(shape, device=DEV).to(dtype) if res else None
    gamma = (torch.rand(C, device=DEV) + 0.5); beta = torch.randn(C, device=DEV) * 0.3
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    rm2, rv2 = rm.clone(), rv.clone()
    x1 = x.clone().requires_grad_(True); g1 = gamma.clone().requires_grad_(True); b1 = beta.clone().requires_grad_(True)
    r1 = r.clone().requires_grad_(True) if res else None
    y = G.batch_norm(x1, g1, b1, rm, rv, True, 0.1, 1e-5, relu, r1)
    dy = torch.randn(shape, device=DEV).to(dtype)
    y.backward(dy)
    xr = x.detach().clone().float().requires_grad_(True); gr = gamma.clone().requires_grad_(True); br = beta.clone().requires_grad_(True)
    rr = r.detach().clone().float().requires_grad_(True) if res else None
    yr = torch.nn.functional.batch_norm(xr, rm2, rv2, gr, br, True, 0.1, 1e-5)
    if res:
        yr = yr + rr
    if relu:
        yr = torch.relu(yr)
    yr.backward(dy.float())
    tol = dict(rtol=1e-4, atol=1e-4) if dtype == torch.float32 else dict(rtol=3e-2, atol=3e-2)
    torch.testing.assert_close(y.float(), yr.detach(), **tol)
    torch.testing.assert_close(x1.grad.float(), xr.grad, **tol)
    n = x.numel() // C
    ptol = dict(rtol=2e-3, atol=2e-3 * max(1.0, n ** 0.5 / 8)) if dtype == torch.float32 else dict(rtol=5e-2, atol=0.05 * n ** 0.5)
    torch.testing.assert_close(g1.grad, gr.grad, **ptol)
    torch.testing.assert_close(b1.grad, br.grad, **ptol)
    if res:
        torch.testing.assert_close(r1.grad.float(), rr.grad, **tol)
    torch.testing.assert_close(rm, rm2, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(rv, rv2, rtol=1e-3, atol=1e-3)
    # inference mode: the running statistics
    with torch.no_grad():
        ye = G.batch_norm(x, gamma, beta, rm, rv, False, 0.1, 1e-5, relu, r)
        yre = torch.nn.functional.batch_norm(x.float(), rm2, rv2, gamma, beta, False, 0.1, 1e-5)
        if res:
            yre = yre + r.float()
        if relu:
            yre = torch.relu(yre)
    torch.testing.assert_close(ye.float(), yre, **tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", [((4, 3, 12, 10), 2, 2, 0), ((2, 16, 9, 9), 3, 2, 1), ((3, 10, 8, 11), (2, 3), (2, 1), (0, 1)),
                                 ((2, 64, 7, 7), 3, 1, 1), ((1, 5, 6, 6), (3, 2), (1, 2), 0)])
def test_general_pooling_matches_torch(dtype, cfg):
    from distributed_tensorflow_amd.ops import generic_nn as G
    shape, k, s, p = cfg
    torch.manual_seed(shape[1])
    x = torch.randn(shape, device=DEV).to(dtype).contiguous(memory_format=torch.channels_last)
    for mine, ref in ((G.max_pool2d, torch.nn.functional.max_pool2d), (G.avg_pool2d, torch.nn.functional.avg_pool2d)):
        x1 = x.clone().requires_grad_(True)
        y = mine(x1, k, s, p)
        # reference on the CPU in fp32 (the vendor library's average-pool gradient for padded, overlapping windows of
        # channels_last tensors disagrees with torch's own CPU kernel on this platform)
        xr = x.detach().cpu().float().contiguous().requires_grad_(True)
        yr = ref(xr, k, s, p)
        assert y.shape == yr.shape
        dy = torch.randn(yr.shape).to(dtype)
        y.backward(dy.to(DEV))
        yr.backward(dy.float())
        tol = dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)
        torch.testing.assert_close(y.float().cpu(), yr.detach(), **tol)
        torch.testing.assert_close(x1.grad.float().cpu(), xr.grad, **tol)
    x1 = x.clone().requires_grad_(True)
    g = G.global_avg_pool(x1)
    xr = x.detach().clone().float().requires_grad_(True)
    gr = xr.mean(dim=(2, 3))
    dg = torch.randn(gr.shape, device=DEV).to(dtype)
    g.backward(dg); gr.backward(dg.float())
    tol = dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    torch.testing.assert_close(g.float(), gr.detach(), **tol)
    torch.testing.assert_close(x1.grad.float(), xr.grad, **tol)


def test_general_max_pool_propagates_nan_and_picks_the_first_maximum():
    from distributed_tensorflow_amd.ops import generic_nn as G
    x = torch.zeros(1, 1, 4, 4, device=DEV).contiguous(memory_format=torch.channels_last)
    x[0, 0, 0, 1] = float("nan")
    x[0, 0, 2, 2] = 1.0; x[0, 0, 3, 3] = 1.0           # two equal maxima in one window
    x.requires_grad_(True)
    y = G.max_pool2d(x, 2, 2, 0)
    assert torch.isnan(y[0, 0, 0, 0]) and float(y[0, 0, 1, 1]) == 1.0
    y.backward(torch.ones_like(y))
    assert float(x.grad[0, 0, 2, 2]) == 1.0 and float(x.grad[0, 0, 3, 3]) == 0.0


def test_fp32_keras_layers_run_on_own_kernels():
    """A fp32 Keras-style CNN with BatchNormalization, both poolings, an Embedding and a strided convolution: forward and
    gradients against the same model evaluated with torch's own operators."""
    import torch.nn.functional as F
    from distributed_tensorflow_amd.api import layers as L
    from distributed_tensorflow_amd.api.model import Sequential
    torch.manual_seed(0)
    m = Sequential([L.Conv2D(12, 3, strides=2, padding="same", activation="relu", input_shape=(17, 15, 3)),
                    L.BatchNormalization(), L.MaxPooling2D((2, 3), (2, 1)), L.Conv2D(10, 1), L.AveragePooling2D(2),
                    L.GlobalAveragePooling2D(), L.Dense(5)]).to(DEV)
    m.train()
    x = torch.randn(6, 17, 15, 3, device=DEV)
    out = m(x, logits=True) if getattr(m, "supports_logits", False) else m(x)
    out.float().square().sum().backward()
    assert out.shape == (6, 5) and all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
    # Embedding: gather + sorted-run gradient kernels
    e = L.Embedding(50, 8).to(DEV)
    e.build((4,))
    ids = torch.randint(0, 50, (7, 4), device=DEV)
    y = e(ids)
    y.sum().backward()
    ref = torch.nn.functional.embedding(ids, e.emb.weight.detach())
    torch.testing.assert_close(y.detach(), ref)
    want = torch.zeros(50, 8, device=DEV).index_add_(0, ids.reshape(-1), torch.ones(28, 8, device=DEV))
    torch.testing.assert_close(e.emb.weight.grad, want)


def test_fp32_convolution_has_no_library_route():
    """conv2d of fp32 tensors (and of bf16 geometries the implicit-GEMM kernel does not take) runs on the general GEMM."""
    from distributed_tensorflow_amd.ops import conv
    assert not hasattr(conv, "set_conv1x1_backend") and not hasattr(conv, "set_convkxk_backend")
    torch.manual_seed(1)
    for dtype, shape, k, s, p in ((torch.float32, (3, 16, 12, 12), 3, 1, 1), (torch.float32, (2, 3, 9, 9), 3, 2, 1),
                                  (torch.bfloat16, (2, 12, 9, 7), 3, 1, 1)):
        N, C, H, W = shape
        x = torch.randn(shape, device=DEV).to(dtype).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(20, C, k, k, device=DEV) * 0.2).to(dtype).contiguous(memory_format=torch.channels_last)
        x1 = x.clone().requires_grad_(True); w1 = w.clone().requires_grad_(True)
        y = conv.conv2d(x1, w1, s, p)
        xr = x.detach().clone().float().requires_grad_(True); wr = w.detach().clone().float().requires_grad_(True)
        yr = torch.nn.functional.conv2d(xr, wr, None, s, p)
        dy = torch.randn_like(yr)
        y.backward(dy.to(dtype)); yr.backward(dy)
        tol = dict(rtol=1e-4, atol=1e-4) if dtype == torch.float32 else dict(rtol=3e-2, atol=3e-2)
        torch.testing.assert_close(y.float(), yr.detach(), **tol)
        torch.testing.assert_close(x1.grad.float(), xr.grad, **tol)
        torch.testing.assert_close(w1.grad.float(), wr.grad, **(tol if dtype == torch.float32 else dict(rtol=5e-2, atol=0.15)))


def test_general_convolution_with_more_than_65535_row_tiles():
    """The general (fallback) convolution folds its row tiles into grid.x: 4.46 million output pixels are 69,696 tiles of
    64 rows, which a grid.y could not hold (a 1536-image ResNet-50 stem reached this and failed to launch)."""
    from distributed_tensorflow_amd.ops import conv
    torch.manual_seed(2)
    x = torch.randn(36, 3, 352, 352, device=DEV).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(8, 3, 3, 3, device=DEV) * 0.2).contiguous(memory_format=torch.channels_last)
    y = conv.conv2d(x, w, 1, 1)
    ref = torch.nn.functional.conv2d(x, w, None, 1, 1)
    torch.testing.assert_close(y, ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("shape", [(33000, 768, 256), (16640, 1280, 192), (2000, 520, 128)])
def test_gemm256_multiply_epilogue_with_column_sums(shape):
    """The multiply epilogue also returns the column sums of what it stores (full tiles, edge tiles, several tiles per
    workgroup): equal to a sum over the stored bf16 result, and the result itself unchanged by the option."""
    from distributed_tensorflow_amd.ops import gemm
    M, N, K = shape
    torch.manual_seed(M + N)
    a = torch.randn(M, K, device=DEV).bfloat16()
    bt = torch.randn(K, N, device=DEV).bfloat16()
    cin = torch.randn(M, N, device=DEV).bfloat16()
    plain = gemm.gemm256(gemm.NN, a, bt, M, N, K, mul=cin)
    out, sums = gemm.gemm256(gemm.NN, a, bt, M, N, K, mul=cin, col_sum=True)
    assert torch.equal(out, plain)
    ref = out.double().sum(0)
    assert torch.allclose(sums.double(), ref, rtol=1e-4, atol=1e-2 * float(ref.abs().max())), float((sums.double() - ref).abs().max())
    out2, sums2 = gemm.gemm256(gemm.NN, a, bt, M, N, K, mul=cin, col_sum=True)
    assert torch.equal(out, out2) and torch.equal(sums, sums2)


def test_flash_backward_offers_the_qkv_bias_gradient():
    """The fused attention's backward kernels sum the columns of the dqkv they store; the QKV projection's backward
    takes those sums as its bias gradient instead of a pass over dqkv."""
    from distributed_tensorflow_amd.ops import attention as A, dense, gemm, layernorm as LN
    torch.manual_seed(5)
    B, S, nh, hd, H = 3, 256, 4, 64, 256
    x = torch.randn(B, S, H, device=DEV).bfloat16()
    w = (torch.randn(3 * nh * hd, H, device=DEV) * 0.05).bfloat16().requires_grad_(True)
    bias = torch.zeros(3 * nh * hd, device=DEV).bfloat16().requires_grad_(True)
    dout = torch.randn(B, S, nh, hd, device=DEV).bfloat16()
    grads, passes = {}, {}
    real = gemm.col_sum
    for on in (True, False):
        calls = []
        gemm.col_sum = lambda g, dtype=torch.float32: (calls.append(tuple(g.shape)), real(g, dtype))[1]
        LN._hand_over = on
        try:
            w.grad = bias.grad = None
            qkv = dense.linear(x, w, bias).view(B, S, 3, nh, hd)
            A.self_attention(qkv, None).backward(dout)
        finally:
            gemm.col_sum = real
            LN._hand_over = True
        grads[on], passes[on] = bias.grad.float().clone(), len(calls)
    assert passes[True] == 0 and passes[False] == 1
    scale = float(grads[False].abs().max())
    assert torch.allclose(grads[True], grads[False], rtol=2e-2, atol=2e-2 * scale), float((grads[True] - grads[False]).abs().max())


@pytest.mark.parametrize("block", ["bottleneck", "basic"])
def test_lazy_down_sampling_norm_matches_the_written_one(block):
    """The down-sampling branch's norm evaluated inside the closing norm-add-ReLU (its normalised tensor never written)
    against the same block with that tensor written: outputs differ only by the bf16 rounding of the branch tensor, all
    gradients agree, and the running statistics are identical."""
    from distributed_tensorflow_amd.models import resnet as R
    from distributed_tensorflow_amd.ops import norm
    torch.manual_seed(21)
    blk = (R.Bottleneck(64, 32, 2) if block == "bottleneck" else R.BasicBlock(64, 128, 2)).to(DEV)
    for m in blk.modules():
        if isinstance(m, R.Conv):
            m.weight.data = m.weight.data.bfloat16().contiguous(memory_format=torch.channels_last)
        if isinstance(m, R.BNAct):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.3)
    blk.train()
    x = torch.randn(8, 64, 16, 16, device=DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    res = {}
    try:
        for lazy in (True, False):
            norm._lazy_residual = lazy
            for m in blk.modules():
                if isinstance(m, R.BNAct):
                    m.running_mean.zero_(); m.running_var.fill_(1.0)
            for p in blk.parameters():
                p.grad = None
            xi = x.clone().requires_grad_(True)
            out = blk(xi)
            (out.float() * torch.linspace(0.5, 1.5, out.numel(), device=DEV).reshape(out.shape)).sum().backward()
            res[lazy] = (out.float().detach(), xi.grad.float(), {n: p.grad.float().clone() for n, p in blk.named_parameters()},
                         blk.down_bn.running_mean.clone(), blk.down_bn.running_var.clone())
    finally:
        norm._lazy_residual = True
    a, b = res[True], res[False]
    assert torch.allclose(a[0], b[0], atol=6e-2, rtol=2e-2)
    assert torch.equal(a[3], b[3]) and torch.equal(a[4], b[4])

    def rel(u, v):
        return float((u - v).norm() / (v.norm() + 1e-12))
    # the two differ by one bf16 rounding of the branch tensor: a few ReLU decisions next to zero flip, and gradients
    # through batch norms cancel heavily, so a few percent of relative L2 distance is rounding, a wrong sum is not
    assert rel(a[1], b[1]) < 8e-2
    for n in a[2]:
        assert rel(a[2][n], b[2][n]) < 8e-2, (n, rel(a[2][n], b[2][n]))


@pytest.mark.parametrize("lazy", [True, False], ids=["lazy", "written"])
def test_down_sampling_block_backward_in_one_apply_pass(lazy):
    """The backward of a down-sampling branch's norm run inside the closing norm-add-ReLU's backward (one apply pass
    writes both input gradients) against the two separate passes: the same sums in the same order and the same fused
    multiply-adds per element, so every gradient is bit-identical. The block is followed by a second one, whose first
    convolution hands the closing norm its reduced gradient sums: only then does the one-pass form apply."""
    from distributed_tensorflow_amd.models import resnet as R
    from distributed_tensorflow_amd.ops import norm
    torch.manual_seed(22)
    net = torch.nn.Sequential(R.Bottleneck(64, 32, 2), R.Bottleneck(128, 32, 1)).to(DEV)
    for m in net.modules():
        if isinstance(m, R.Conv):
            m.weight.data = m.weight.data.bfloat16().contiguous(memory_format=torch.channels_last)
        if isinstance(m, R.BNAct):
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.3)
    net.train()
    x = torch.randn(8, 64, 16, 16, device=DEV).bfloat16().contiguous(memory_format=torch.channels_last)
    res, launches = {}, {}
    try:
        norm._lazy_residual = lazy
        for dual in (True, False):
            norm._dual_bwd = dual
            before = norm.dual_launches
            for p in net.parameters():
                p.grad = None
            xi = x.clone().requires_grad_(True)
            out = net(xi)
            (out.float() * torch.linspace(0.5, 1.5, out.numel(), device=DEV).reshape(out.shape)).sum().backward()
            res[dual] = (xi.grad.clone(), {n: p.grad.clone() for n, p in net.named_parameters()})
            launches[dual] = norm.dual_launches - before
    finally:
        norm._lazy_residual = True
        norm._dual_bwd = True
    assert launches == {True: 1, False: 0}
    assert torch.equal(res[True][0], res[False][0])
    for n in res[True][1]:
        assert torch.equal(res[True][1][n], res[False][1][n]), n
    g = res[True][1]["0.down_bn.weight"]
    assert torch.isfinite(g).all() and float(g.abs().sum()) > 0


def test_vram_pretouch_child_process_touches_free_memory():
    """utils.vram.pretouch: a child process allocates and writes a fraction of the device's free memory and reports
    how much; this process's own allocations are untouched and usable afterwards."""
    from distributed_tensorflow_amd.utils.vram import pretouch
    keep = torch.arange(1024, device=DEV)
    free, _ = torch.cuda.mem_get_info(DEV)
    r = pretouch(0, fraction=0.25)
    assert r["ok"], r
    assert 0.1 * free / 2 ** 30 <= r["gib"] <= 0.26 * free / 2 ** 30 + 1.0
    assert int(keep.sum()) == 1023 * 1024 // 2
