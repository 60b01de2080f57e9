"""What does staleness alone cost? One process, the reference MLP, Adam, batch 64, three epochs on the noisy-teacher
task: every gradient is computed on the parameters of ``delay`` applies ago (the situation of a worker of an
asynchronous parameter server with ``delay + 1`` workers in step), applied at the full learning rate or at
``lr / (1 + delay)``. The numbers are the yardstick for the asynchronous rows of tools/convergence_teacher.sh: a parameter
server that loses or doubles updates would fall below them, one that matches them implements delayed-gradient Adam.

    DTF_SYNTH_MNIST_KIND=teacher python tools/simulate_stale_adam.py
"""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DTF_SYNTH_MNIST_KIND", "teacher")
from distributed_tensorflow_amd.data.sources import load_mnist  # noqa: E402


def run(X, Y, XT, yte, delay, lr, scale):
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Flatten(), torch.nn.Linear(784, 512), torch.nn.ReLU(), torch.nn.Dropout(0.2),
                            torch.nn.Linear(512, 10))
    opt = torch.optim.Adam(m.parameters(), lr, eps=1e-7)
    hist = [copy.deepcopy(m.state_dict())]
    for _ in range(3):
        for i in range(0, X.shape[0], 64):
            cur = copy.deepcopy(m.state_dict())
            m.load_state_dict(hist[max(0, len(hist) - 1 - delay)])
            loss = torch.nn.functional.cross_entropy(m(X[i:i + 64]), Y[i:i + 64])
            opt.zero_grad()
            loss.backward()
            g = [p.grad.clone() for p in m.parameters()]
            m.load_state_dict(cur)
            for p, gg in zip(m.parameters(), g):
                p.grad = gg
            for grp in opt.param_groups:
                grp["lr"] = lr / (1 + delay) if scale else lr
            opt.step()
            hist = (hist + [copy.deepcopy(m.state_dict())])[-(delay + 2):]
    m.eval()
    with torch.no_grad():
        return float((m(XT).argmax(1).numpy() == yte).mean())


def main():
    (xtr, ytr), (xte, yte), src = load_mnist()
    X, Y, XT = torch.tensor(xtr).float() / 255, torch.tensor(ytr).long(), torch.tensor(xte).float() / 255
    print(f"data: {src}")
    for delay in (0, 3, 7):
        for scale in ((False,) if delay == 0 else (False, True)):
            acc = run(X, Y, XT, yte, delay, 1e-3, scale)
            print(f"delay {delay}  lr {'/(1+delay)' if scale else 'full':10s}  test accuracy {acc:.4f}", flush=True)


if __name__ == "__main__":
    main()
