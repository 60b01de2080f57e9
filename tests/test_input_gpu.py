"""The host input pipeline of bench.py on the GPU: loader threads -> pinned staging -> copy stream -> normalise."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_host_input_pipeline_uploads_pool_rows_ahead_of_their_step():
    from distributed_tensorflow_amd.bench_lib import HostInputPipeline
    dev = torch.device("cuda:0")
    feed = HostInputPipeline(16, dev, size=24, classes=10, pool=64, seed=5)
    assert feed.loader.pinned
    labels = []
    busy = torch.zeros(1 << 22, device=dev)
    for step in range(6):
        x, y = feed.next()
        assert x.is_cuda and x.dtype == torch.bfloat16 and x.shape == (16, 24, 24, 3)
        # keep the main stream busy with work that reads the batch, as a training step would, so that the next
        # upload really has to wait for the buffer set it reuses
        for _ in range(4):
            busy.add_(x.float().sum())
        raw = (x.float() * 64.0 + 127.5).round().clamp(0, 255).to(torch.uint8).cpu().numpy()
        yy = y.cpu().numpy()
        for i in range(16):
            match = [j for j in range(64) if int(feed._y[j]) == int(yy[i])
                     and np.abs(raw[i].astype(int) - feed._x[j].astype(int)).max() <= 1]
            assert match, f"step {step}: delivered image {i} is not a pool row"
        labels.append(yy.copy())
    assert any(not np.array_equal(labels[0], l) for l in labels[1:])
    feed.close()
