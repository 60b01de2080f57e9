"""CPU unit tests: flat arenas, bucket planner, optimizer formulas, Keras weight format, dataset pipeline."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from distributed_tensorflow_amd.api import keras_weights as kw  # noqa: E402
from distributed_tensorflow_amd.api.layers import BatchNormalization, Conv2D, Dense, Flatten  # noqa: E402
from distributed_tensorflow_amd.api.model import Sequential, as_model  # noqa: E402
from distributed_tensorflow_amd.data.dataset import Dataset  # noqa: E402
from distributed_tensorflow_amd.engine.flat import ALIGN, BucketPlanner, FlatParams  # noqa: E402
from distributed_tensorflow_amd.engine.optimizer import SGD, Adam, FlatOptimizer  # noqa: E402
from distributed_tensorflow_amd.engine.replica import Replica, TrainConfig  # noqa: E402
from distributed_tensorflow_amd.models.mlp import mnist_cnn, reference_mlp  # noqa: E402


def test_reference_mlp_has_the_reference_parameter_count():
    m = reference_mlp()
    assert m.count_params() == 407050            # 784*512 + 512 + 512*10 + 10
    ws = m.get_weights()
    assert [w.shape for w in ws] == [(784, 512), (512,), (512, 10), (10,)]   # Keras kernel layout (in, out)
    p = m(torch.rand(5, 28, 28, 1))
    assert p.shape == (5, 10) and torch.allclose(p.sum(1), torch.ones(5), atol=1e-5)


def test_flat_params_views_and_reverse_order():
    m = reference_mlp()
    before = [p.detach().clone() for p in m.parameters()]
    f = FlatParams.from_module(m, shard_ways=4)
    assert f.total % (ALIGN * 4) == 0
    for p, b in zip(m.parameters(), before):
        assert torch.equal(p.detach(), b)
    names = [s.name for s in f.slots]
    assert names == list(reversed([n for n, _ in m.named_parameters()]))   # last layer first
    # parameters are views: writing the arena changes the module
    f.master.zero_()
    assert all(float(p.detach().abs().sum()) == 0 for p in m.parameters())


def test_flat_params_keep_channels_last_weights():
    m = mnist_cnn()
    conv_w = m.layers[0].weight
    assert conv_w.is_contiguous(memory_format=torch.channels_last)
    ref = conv_w.detach().clone()
    f = FlatParams.from_module(m)
    assert m.layers[0].weight.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(m.layers[0].weight.detach(), ref)
    assert f.by_name["layers.0.weight"].stride == tuple(ref.stride())


def test_bf16_params_get_master_and_shadow():
    m = reference_mlp()
    rep = Replica(m, "cpu", TrainConfig(param_dtype=torch.bfloat16))
    f = rep.flat
    assert f.shadow is not None and f.master.dtype == torch.float32
    for s in f.slots:
        assert s.param.dtype == torch.bfloat16
        assert s.param.data_ptr() == f.shadow[s.offset:].data_ptr()
    opt = FlatOptimizer(f, SGD(learning_rate=0.5))
    f.grad.fill_(1.0)
    m0 = f.master.clone()
    opt.step()
    assert torch.allclose(f.master, m0 - 0.5)
    assert torch.equal(f.shadow, f.master.to(torch.bfloat16))   # compute copy refreshed in the same pass


@pytest.mark.parametrize("ways", [1, 2, 8])
def test_bucket_planner_partitions_the_arena(ways):
    m = mnist_cnn()
    f = FlatParams.from_module(m, shard_ways=ways)
    pl = BucketPlanner(f, cap_bytes=200_000, first_cap_bytes=20_000)
    assert pl.buckets[0].start == 0 and pl.buckets[-1].end == f.total
    for a, b in zip(pl.buckets, pl.buckets[1:]):
        assert a.end == b.start
    assert sum(len(b.slots) for b in pl.buckets) == len(f.slots)
    for b in pl.buckets:
        assert b.numel % (ALIGN * ways) == 0
        shards = [pl.shard_range(b, r) for r in range(ways)]
        assert shards[0][0] == b.start and shards[-1][1] == b.end
        assert all(e - s == b.numel // ways for s, e in shards)
    assert len(pl.buckets) > 2


def test_bucket_planner_cuts_a_small_last_bucket():
    m = mnist_cnn()
    f = FlatParams.from_module(m)
    whole = BucketPlanner(f, cap_bytes=1 << 30)
    assert len(whole.buckets) == 1
    esize = f.grad.element_size()
    tail = (f.total - f.slots[-2].offset) * esize           # the last two slots
    pl = BucketPlanner(f, cap_bytes=1 << 30, last_cap_bytes=tail)
    assert len(pl.buckets) == 2 and len(pl.buckets[1].slots) == 2
    assert pl.buckets[0].end == pl.buckets[1].start == f.slots[-2].offset and pl.buckets[1].end == f.total
    assert sum(len(b.slots) for b in pl.buckets) == len(f.slots)
    # a cap smaller than the last slot still leaves that slot a bucket of its own
    pl = BucketPlanner(f, cap_bytes=1 << 30, last_cap_bytes=1)
    assert len(pl.buckets) == 2 and len(pl.buckets[1].slots) == 1
    # and a last bucket that already fits is left alone
    assert len(BucketPlanner(f, cap_bytes=1 << 30, last_cap_bytes=1 << 30).buckets) == 1


def test_adam_keras_formula_matches_handwritten():
    torch.manual_seed(0)
    m = reference_mlp()
    f = FlatParams.from_module(m)
    opt = FlatOptimizer(f, Adam())
    p0 = f.master.clone()
    g = torch.randn_like(f.grad)
    mm = torch.zeros_like(g); vv = torch.zeros_like(g); p = p0.clone()
    for t in range(1, 4):
        f.grad.copy_(g)
        opt.step()
        mm = 0.9 * mm + 0.1 * g
        vv = 0.999 * vv + 0.001 * g * g
        lr_t = 1e-3 * (1 - 0.999 ** t) ** 0.5 / (1 - 0.9 ** t)
        p = p - lr_t * mm / (vv.sqrt() + 1e-7)     # Keras: epsilon outside the bias-corrected sqrt
    assert torch.allclose(f.master, p, atol=1e-6)


def test_torch_adam_and_sgd_formulas_match_torch_optim():
    torch.manual_seed(1)
    for spec, ref_cls, kw_ in [(Adam(formula="torch", epsilon=1e-8), torch.optim.Adam, dict(lr=1e-3)),
                               (SGD(learning_rate=0.1, momentum=0.9, formula="torch", weight_decay=1e-3), torch.optim.SGD,
                                dict(lr=0.1, momentum=0.9, weight_decay=1e-3))]:
        m = reference_mlp()
        f = FlatParams.from_module(m)
        ref_p = torch.nn.Parameter(f.master.clone())
        ref = ref_cls([ref_p], **kw_)
        opt = FlatOptimizer(f, spec)
        for _ in range(3):
            g = torch.randn_like(f.grad)
            f.grad.copy_(g); ref_p.grad = g.clone()
            opt.step(); ref.step()
        assert torch.allclose(f.master, ref_p.detach(), atol=2e-6)


def test_keras_weight_layouts_and_roundtrip(tmp_path):
    m = Sequential([Conv2D(4, 3, input_shape=(8, 8, 2), padding="same"), BatchNormalization(), Flatten(), Dense(5)])
    ws = m.get_weights()
    assert [w.shape for w in ws] == [(3, 3, 2, 4), (4,), (4,), (4,), (4,), (4,), (256, 5), (5,)]
    # HWIO <-> OIHW: kernel[h, w, i, o] == weight[o, i, h, w]
    tw = m.layers[0].weight.detach().numpy()
    assert ws[0][1, 2, 1, 3] == tw[3, 1, 1, 2]
    # Dense kernel is (in, out)
    assert ws[6][7, 2] == m.layers[3].weight.detach().numpy()[2, 7]
    new = [np.random.default_rng(i).normal(size=w.shape).astype(np.float32) for i, w in enumerate(ws)]
    new[5] = np.abs(new[5]) + 0.5     # moving_variance must be positive
    m.set_weights(new)
    for a, b in zip(m.get_weights(), new):
        assert np.array_equal(a, b)
    path = str(tmp_path / "w.npz")
    m.save_weights(path)
    m2 = Sequential([Conv2D(4, 3, input_shape=(8, 8, 2), padding="same"), BatchNormalization(), Flatten(), Dense(5)])
    m2.load_weights(path)
    x = torch.rand(3, 8, 8, 2)
    m.eval(); m2.eval()
    assert torch.allclose(m(x), m2(x), atol=1e-6)
    with pytest.raises(ValueError):
        m.set_weights(new[:-1])


def test_keras_weights_of_plain_torch_modules():
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3), torch.nn.BatchNorm2d(8), torch.nn.Flatten(), torch.nn.Linear(8, 2))
    ws = kw.get_weights(net)
    assert [w.shape for w in ws] == [(3, 3, 3, 8), (8,), (8,), (8,), (8,), (8,), (8, 2), (2,)]
    kw.set_weights(net, ws)
    wrapped = as_model(net)
    assert len(wrapped.get_weights()) == len(ws) and len(wrapped.trainable_variables) == 6


def test_flatten_uses_nhwc_order_like_keras():
    m = Sequential([Conv2D(2, 1, input_shape=(2, 2, 1), use_bias=False), Flatten()])
    with torch.no_grad():
        m.layers[0].weight.copy_(torch.tensor([1.0, 10.0]).view(2, 1, 1, 1))
    x = torch.arange(4.0).view(1, 2, 2, 1)
    out = m(x)[0].tolist()
    assert out == [0.0, 0.0, 1.0, 10.0, 2.0, 20.0, 3.0, 30.0]   # (h, w, c) fastest-last


def test_dataset_shard_batch_shuffle_semantics():
    x = np.arange(20, dtype=np.float32).reshape(20, 1)
    y = np.arange(20, dtype=np.int32)
    ds = Dataset.from_tensor_slices((x, y)).shard(2, 1).batch(3).shuffle(100, seed=7)
    batches = [b for b in ds]
    assert len(batches) == 4 and ds.num_batches() == 4
    rows = sorted(int(v) for _, yy in batches for v in yy)
    assert rows == list(range(1, 20, 2))                      # every 2nd element starting at 1
    # shuffle after batch permutes batches, never examples inside a batch (reference initializer.py:44-45)
    for _, yy in batches:
        v = yy.tolist()
        assert v == sorted(v) and all(b - a == 2 for a, b in zip(v, v[1:]))
    with pytest.raises(ValueError):
        Dataset.from_tensor_slices((x, y)).batch(2).shard(2, 0)
    # a new epoch gives a new batch order with the same content
    again = [yy.tolist() for _, yy in ds]
    assert sorted(map(tuple, again)) == sorted(tuple(yy.tolist()) for _, yy in batches)


def test_replica_sum_vs_mean_loss_gradients():
    torch.manual_seed(0)
    x = torch.rand(8, 28, 28, 1); y = torch.randint(0, 10, (8,), dtype=torch.int32)
    grads = {}
    for red in ("sum", "mean"):
        torch.manual_seed(0)
        rep = Replica(reference_mlp(dropout=0.0), "cpu", TrainConfig(loss_reduction=red))
        rep.forward_backward(x, y)
        rep.finish_backward()
        grads[red] = rep.flat.grad.clone()
    assert torch.allclose(grads["sum"], grads["mean"] * 8, rtol=1e-4, atol=1e-6)


def test_reference_loss_on_clipped_probabilities():
    """probs_clip_eps reproduces sparse_categorical_crossentropy applied to softmax outputs."""
    from distributed_tensorflow_amd.ops.softmax_ce import softmax_cross_entropy
    torch.manual_seed(0)
    logits = torch.randn(6, 10) * 30          # saturating
    y = torch.randint(0, 10, (6,))
    exact = softmax_cross_entropy(logits, y)
    clipped = softmax_cross_entropy(logits, y, probs_clip_eps=1e-7)
    assert float(clipped.max()) <= -np.log(1e-7) + 1e-3      # bounded by -log(eps)
    assert float(exact.max()) > float(clipped.max())
    small = torch.randn(6, 10)
    assert torch.allclose(softmax_cross_entropy(small, y), softmax_cross_entropy(small, y, probs_clip_eps=1e-7), atol=1e-5)


def test_conv_norm_hand_over_api_degrades_gracefully_off_the_gpu_path():
    """The statistics hand-over is a GPU fast path; on CPU the same calls return the documented fall-backs."""
    import torch
    from distributed_tensorflow_amd.models import resnet as R
    from distributed_tensorflow_amd.ops import conv, norm
    x = torch.randn(2, 8, 6, 6).contiguous(memory_format=torch.channels_last)
    w = torch.randn(16, 8, 3, 3).contiguous(memory_format=torch.channels_last)
    y, st = conv.conv2d(x, w, 1, 1, bn_stats=True)
    assert st is None and y.shape == (2, 16, 6, 6)
    y2, alias, st2 = conv.conv2d_fork(x.requires_grad_(True), w, 1, 1, bn_stats=True)
    assert st2 is None and alias is x and torch.equal(y2, y)
    assert norm.request_of(y) is None
    # mask bits: bit k of byte j keeps element 8 j + k (channels_last memory order)
    g = torch.arange(2 * 8 * 2 * 2, dtype=torch.float32).reshape(2, 2, 2, 8).permute(0, 3, 1, 2)
    bits = torch.tensor([0b00000001, 0b10000000, 0xFF, 0x00, 0x0F, 0xF0, 0xAA, 0x55], dtype=torch.uint8)
    out = conv._apply_mask_bits(g, bits).permute(0, 2, 3, 1).reshape(8, 8)
    ref = g.permute(0, 2, 3, 1).reshape(8, 8) * torch.tensor(
        [[(int(b) >> k) & 1 for k in range(8)] for b in bits], dtype=torch.float32)
    assert torch.equal(out, ref)
    # a bottleneck block trains and evaluates through the same code on CPU
    blk = R.Bottleneck(16, 8, 2)
    xin = torch.randn(2, 16, 8, 8, requires_grad=True)
    blk.train()
    blk(xin).sum().backward()
    assert xin.grad is not None and all(p.grad is not None for p in blk.parameters())
    blk.eval()
    with torch.no_grad():
        assert blk(xin).shape == (2, 32, 4, 4)


def test_standalone_parser_keeps_the_reference_role_flags():
    from distributed_tensorflow_amd.cli import standalone
    p = standalone.build_parser()
    a, extra = p.parse_known_args(["worker", "-ps", "1", "-n", "3", "-b", "64", "-ba", "10.0.0.", "-sa", "h", "-s", "srv",
                                   "--device", "cpu"])
    assert (a.role, a.index, a.n_nodes, a.batch_size, a.server_address) == ("worker", 1, 3, 64, "srv")
    assert extra == ["--device", "cpu"]
    a, _ = p.parse_known_args(["server", "-n", "2", "-sm", "async"])
    assert (a.role, a.sync_mode, a.server_address) == ("server", "async", "127.0.0.1")
    a, _ = p.parse_known_args(["keras", "-n", "2", "-i", "1", "-w", "a:1,b:2"])
    assert (a.role, a.index, a.batch_size, a.worker_addresses) == ("keras", 1, 100, "a:1,b:2")


def test_package_import_keeps_the_conv_library_reference_kernels_out_of_its_search(monkeypatch):
    """Importing the package must switch the vendor convolution library's reference solvers off by default (they
    cost over a minute of the first large-batch step) and must leave a user's own setting alone."""
    import importlib
    import os
    import distributed_tensorflow_amd as pkg
    names = ["MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD", "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_BWD",
             "MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_WRW"]
    for n in names:
        monkeypatch.delenv(n, raising=False)
    monkeypatch.setenv(names[0], "1")
    importlib.reload(pkg)
    assert os.environ[names[0]] == "1"
    assert os.environ[names[1]] == "0" and os.environ[names[2]] == "0"


def test_layer_norm_column_sum_hand_over_matches_only_the_untouched_gradient():
    """``ops.layernorm.take_col_sum``: the sums recorded for a gradient are handed out once, and only for that very
    tensor: same address, same version counter, same size, contiguous."""
    from distributed_tensorflow_amd.ops import layernorm as ln
    ln._handed.clear()
    dx = torch.randn(6, 8)
    cs = dx.sum(0)

    def record():
        ln._handed.clear()
        ln._handed[dx.data_ptr()] = (dx, dx._version, cs)
    record()
    assert ln.take_col_sum(dx.reshape(-1, 8)) is cs          # a view of the same memory, untouched
    assert ln.take_col_sum(dx) is None                       # handed out once
    record()
    assert ln.take_col_sum(dx.clone()) is None               # another tensor
    assert ln.take_col_sum(dx[:3]) is None                   # same address, fewer rows
    record()
    dx.add_(1.0)                                             # summed into in place: the version counter moved
    assert ln.take_col_sum(dx) is None
    record()
    assert ln.take_col_sum(dx.reshape(8, 6)) is None         # same memory read with another row length
    ln._handed.clear()
    # the CPU composition never records anything
    x = torch.randn(4, 8, requires_grad=True)
    ln.layer_norm(x, torch.ones(8), torch.zeros(8)).sum().backward()
    assert not ln._handed
