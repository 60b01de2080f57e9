"""Runs each hot kernel a few times so that `rocprofv3 --pmc ...` can attach hardware counters to it.

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES ... --output-format csv -d gpurun_out/pmc_sq -- \
        python3 tools/pmc_kernels.py

The shapes are those of the ResNet-50 (batch 256) and BERT-base (batch 64 x 512) steps.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributed_tensorflow_amd.ops import bucket as B  # noqa: E402
from distributed_tensorflow_amd.ops import conv as C  # noqa: E402
from distributed_tensorflow_amd.ops import gemm  # noqa: E402
from distributed_tensorflow_amd.ops import layernorm as LN  # noqa: E402
from distributed_tensorflow_amd.ops import norm as N  # noqa: E402
from distributed_tensorflow_amd.ops import optim as K  # noqa: E402
from distributed_tensorflow_amd.ops.softmax_ce import softmax_cross_entropy  # noqa: E402

DEV = torch.device("cuda:0")
ITERS = int(os.environ.get("PMC_ITERS", "3"))


def rep(fn):
    for _ in range(ITERS):
        fn()
    torch.cuda.synchronize()


def main():
    bf = torch.bfloat16
    # GEMM: square, BERT FFN, ResNet 1x1 (as GEMM), wgrad shape
    for (M, Nn, Kd) in [(4096, 4096, 4096), (32768, 3072, 768), (802816, 256, 64), (50176, 1024, 256)]:
        A = torch.randn(M, Kd, device=DEV).to(bf)
        W = torch.randn(Nn, Kd, device=DEV).to(bf)
        G = torch.randn(M, Nn, device=DEV).to(bf)
        rep(lambda: gemm.gemm_nt(A, W))
        rep(lambda: gemm.gemm_nn(G, W))
        rep(lambda: gemm.gemm_tn(G, A))
        del A, W, G
    # fused attention, BERT-base shape (64 x 512 tokens, 12 heads of 64)
    from distributed_tensorflow_amd.ops import attention as A
    qkv = (torch.randn(64, 512, 3, 12, 64, device=DEV) * 0.5).to(bf).requires_grad_(True)
    rep(lambda: A.self_attention(qkv).sum().backward())
    del qkv
    # implicit-GEMM 3x3 convolution, fwd + bwd
    x = torch.randn(256, 128, 28, 28, device=DEV).to(bf).to(memory_format=torch.channels_last).requires_grad_(True)
    w = torch.randn(128, 128, 3, 3, device=DEV).to(bf).to(memory_format=torch.channels_last).requires_grad_(True)
    rep(lambda: C.conv2d(x, w, stride=1, padding=1).sum().backward())
    # ResNet-50 stage-1 bottleneck, training mode: 1x1 / 3x3 convolutions that gather the batch-norm statistics in
    # their store phase, norm + residual + ReLU, masked residual-gradient hand-over
    from distributed_tensorflow_amd.models import resnet as R
    blk = R.Bottleneck(256, 64, 1).to(DEV)
    for m_ in blk.modules():
        if isinstance(m_, R.Conv):
            m_.weight.data = m_.weight.data.to(bf).contiguous(memory_format=torch.channels_last)
    blk.train()
    xr = torch.randn(256, 256, 56, 56, device=DEV).to(bf).to(memory_format=torch.channels_last).requires_grad_(True)
    rep(lambda: blk(blk(xr)).sum().backward())
    del xr, blk
    # transformer feed-forward block with GELU and its derivative in the GEMM epilogues
    from distributed_tensorflow_amd.ops import dense as D
    xf = torch.randn(32768, 768, device=DEV).to(bf).requires_grad_(True)
    w1 = (torch.randn(3072, 768, device=DEV) * 0.03).to(bf).requires_grad_(True)
    b1 = torch.zeros(3072, device=DEV).to(bf).requires_grad_(True)
    w2 = (torch.randn(768, 3072, device=DEV) * 0.03).to(bf).requires_grad_(True)
    b2 = torch.zeros(768, device=DEV).to(bf).requires_grad_(True)
    rep(lambda: D.ffn_fork(xf, w1, b1, w2, b2)[0].sum().backward())
    del xf, w1, w2
    # batch norm + ReLU fwd/bwd on the largest ResNet-50 activation
    xb = torch.randn(256, 256, 56, 56, device=DEV).to(bf).to(memory_format=torch.channels_last).requires_grad_(True)
    g = torch.ones(256, device=DEV, requires_grad=True)
    b = torch.zeros(256, device=DEV, requires_grad=True)
    rm, rv = torch.zeros(256, device=DEV), torch.ones(256, device=DEV)
    rep(lambda: N.batch_norm_act(xb, g, b, rm, rv, True, 0.1, 1e-5, relu=True).sum().backward())
    del xb
    # ResNet stem: batch norm + ReLU + 3x3 / 2 max pool in one forward pass, backward on 2x2 input patches
    xs_ = torch.randn(256, 64, 112, 112, device=DEV).to(bf).to(memory_format=torch.channels_last).requires_grad_(True)
    gs = torch.ones(64, device=DEV, requires_grad=True)
    bs = torch.zeros(64, device=DEV, requires_grad=True)
    rms, rvs = torch.zeros(64, device=DEV), torch.ones(64, device=DEV)
    rep(lambda: N.batch_norm_relu_max_pool(xs_, gs, bs, rms, rvs, True, 0.1, 1e-5, 3, 2, 1).sum().backward())
    del xs_
    # layer norm
    xl = torch.randn(32768, 768, device=DEV).to(bf).requires_grad_(True)
    gl = torch.ones(768, device=DEV, requires_grad=True)
    bl = torch.zeros(768, device=DEV, requires_grad=True)
    rep(lambda: LN.layer_norm(xl, gl, bl, 1e-12).sum().backward())
    # softmax cross-entropy (ResNet head, BERT MLM head)
    for (Bn, Cn) in [(256, 1000), (8192, 30522)]:
        xs = torch.randn(Bn, Cn, device=DEV).to(bf).requires_grad_(True)
        ys = torch.randint(0, Cn, (Bn,), device=DEV)
        rep(lambda: softmax_cross_entropy(xs, ys, "mean").backward())
        del xs
    # optimizer + bucket kernels on a ResNet-50-sized arena
    n = 25_600_000
    p = torch.randn(n, device=DEV); gr = torch.randn(n, device=DEV).to(bf)
    m = torch.zeros(n, device=DEV); v = torch.zeros(n, device=DEV)
    sh = torch.zeros(n, device=DEV, dtype=bf)
    h = K.OptHyper(DEV)
    rep(lambda: K.adam_apply(p, gr, m, v, h, shadow=sh))
    rep(lambda: K.sgdm_apply(p, gr, m, h, shadow=sh))
    ts = [torch.randn(s, device=DEV, dtype=bf) for s in [2048 * 512, 512 * 512 * 9, 512 * 2048] * 6]
    offs, o = [], 0
    for t_ in ts:
        offs.append(o); o += (t_.numel() + 63) // 64 * 64
    flat = torch.empty(o, device=DEV, dtype=bf)
    rep(lambda: B.flatten(ts, flat, offs, scale=0.125))
    # peer-pointer reduce (8 local "peers")
    from distributed_tensorflow_amd.comm.ipc import LocalPeers
    bufs = [torch.randn(n, device=DEV).to(bf) for _ in range(8)]
    lp = LocalPeers(bufs)
    rep(lambda: lp.all_reduce("direct", scale=0.125))
    print("pmc workload done")


if __name__ == "__main__":
    main()
