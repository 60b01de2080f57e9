"""How much does exchanging gradients in bf16 cost at 8 ranks?  (CPU study, no GPU needed)

Eight replicas of ResNet-20 (fp32, CPU) each compute the gradient of their own batch. The exact mean gradient (fp32
sum) is compared with what the default exchange produces: every rank scales by 1/8 and rounds to bf16 (the bucket
pack kernel), a ring all-reduce adds the eight contributions one hop at a time and rounds each partial sum to bf16
(seven roundings per element), and the optimizer reads the bf16 result. Reported per parameter tensor group and
overall: relative L2 error, cosine similarity, and for scale the relative L2 distance between the mean gradient and a
single rank's gradient (the sampling noise data parallelism averages over).

usage: python tools/assess_bf16_allreduce.py [ranks] [batch]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from distributed_tensorflow_amd.models.resnet import resnet20
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    torch.manual_seed(0)
    model = resnet20(10)
    params = [p for p in model.parameters() if p.requires_grad]
    grads = []
    g = torch.Generator().manual_seed(1)
    for r in range(world):
        x = torch.randn(batch, 32, 32, 3, generator=g)
        y = torch.randint(0, 10, (batch,), generator=g)
        model.zero_grad(set_to_none=True)
        loss = torch.nn.functional.cross_entropy(model(x, logits=True).float(), y)
        loss.backward()
        grads.append(torch.cat([p.grad.reshape(-1) for p in params]).double())
    exact = torch.stack(grads).mean(0)
    # the default exchange: scale and round (pack), ring order with a rounding per hop
    acc = (grads[0] / world).float().bfloat16()
    for r in range(1, world):
        acc = (acc.float() + (grads[r] / world).float().bfloat16().float()).bfloat16()
    got = acc.double()
    # the same with the exchange in fp32 (--grad-dtype fp32): only fp32 rounding
    acc32 = (grads[0] / world).float()
    for r in range(1, world):
        acc32 = acc32 + (grads[r] / world).float()
    got32 = acc32.double()

    def rel(a, b):
        return float((a - b).norm() / b.norm())

    def cos(a, b):
        return float(torch.dot(a, b) / (a.norm() * b.norm()))

    lines = [f"ResNet-20, {world} ranks x batch {batch}, {exact.numel()} gradient elements",
             f"bf16 exchange (scale 1/{world}, round, ring sum with a rounding per hop): relative L2 error {rel(got, exact):.3e}, "
             f"cosine {cos(got, exact):.8f}",
             f"fp32 exchange:                                                        relative L2 error {rel(got32, exact):.3e}, "
             f"cosine {cos(got32, exact):.8f}",
             f"for scale, one rank's own gradient against the mean (sampling noise):  relative L2 distance "
             f"{rel(grads[0], exact):.3e}, cosine {cos(grads[0], exact):.6f}"]
    off = 0
    worst = (0.0, "")
    for name, p in model.named_parameters():
        n = p.numel()
        e = rel(got[off:off + n], exact[off:off + n]) if float(exact[off:off + n].norm()) > 0 else 0.0
        if e > worst[0]:
            worst = (e, name)
        off += n
    lines.append(f"worst single tensor under the bf16 exchange: {worst[1]} relative L2 error {worst[0]:.3e}")
    out = "\n".join(lines)
    print(out)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/bf16_allreduce_error.txt", "w") as f:
        f.write(out + "\n")


if __name__ == "__main__":
    main()
