"""hipGraph capture of a whole train step: replayed steps must reproduce eager steps."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _mlp_strategy(graph: bool, kind: str = "keras"):
    from distributed_tensorflow_amd.comm.backend import Comm
    from distributed_tensorflow_amd.engine.optimizer import Adam
    from distributed_tensorflow_amd.engine.replica import TrainConfig
    from distributed_tensorflow_amd.models.mlp import reference_mlp
    from distributed_tensorflow_amd.parallel.mirrored import MirroredStrategy
    from distributed_tensorflow_amd.parallel.ps import SyncPSStrategy

    torch.manual_seed(0)
    model = reference_mlp(dropout=0.0)
    cfg = TrainConfig(loss_reduction="mean", seed=0, graph=graph)
    cls = MirroredStrategy if kind == "keras" else SyncPSStrategy
    return cls(model, Comm(0, 1, torch.device("cuda:0")), cfg, optimizer=Adam(1e-3))


@pytest.mark.parametrize("kind", ["keras", "sync"])
def test_graphed_step_matches_eager(kind):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    batches = [(torch.rand(100, 28, 28, 1, generator=g).to(dev), torch.randint(0, 10, (100,), generator=g).to(dev))
               for _ in range(12)]
    eager, graphed = _mlp_strategy(False, kind), _mlp_strategy(True, kind)
    step = graphed.step_callable()
    le, lg = [], []
    for x, y in batches:
        le.append(float(eager.train_step(x, y)))
        lg.append(float(step(x, y)))
    torch.cuda.synchronize()
    assert step.captures == 1 and step.replays >= 8
    assert le == pytest.approx(lg, rel=1e-6, abs=1e-7)
    assert torch.equal(eager.flat.master, graphed.flat.master)
    # a different batch size gets a graph of its own
    x2, y2 = batches[0][0][:64], batches[0][1][:64]
    assert float(step(x2, y2)) == pytest.approx(float(eager.train_step(x2, y2)), rel=1e-6)
    assert step.captures == 2
    assert torch.equal(eager.flat.master, graphed.flat.master)


def test_graphed_resnet20_step_trains():
    from distributed_tensorflow_amd.comm.backend import Comm
    from distributed_tensorflow_amd.engine.optimizer import SGD
    from distributed_tensorflow_amd.engine.replica import TrainConfig
    from distributed_tensorflow_amd.models.resnet import resnet20
    from distributed_tensorflow_amd.parallel.mirrored import MirroredStrategy

    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    cfg = TrainConfig(param_dtype=torch.bfloat16, grad_dtype=torch.bfloat16, graph=True)
    st = MirroredStrategy(resnet20(10), Comm(0, 1, dev), cfg, optimizer=SGD(0.05, momentum=0.9))
    x = torch.randn(64, 32, 32, 3, device=dev).bfloat16()
    y = torch.randint(0, 10, (64,), device=dev)
    step = st.step_callable()
    losses = [float(step(x, y)) for _ in range(25)]
    assert step.captures == 1
    assert losses[-1] < losses[0] * 0.7, losses


@pytest.mark.parametrize("model_name", ["resnet20", "bert_tiny"])
def test_training_is_bit_identical_across_runs_with_a_fixed_seed(model_name):
    """Race / determinism check: every reduction in the hand-written kernels has a fixed order (two-stage partial
    sums, split-K slabs, sorted-run embedding gradients), so two runs from the same seed agree to the bit."""
    from distributed_tensorflow_amd.comm.backend import Comm
    from distributed_tensorflow_amd.engine.optimizer import Adam
    from distributed_tensorflow_amd.engine.replica import TrainConfig
    from distributed_tensorflow_amd.ops import conv
    from distributed_tensorflow_amd.parallel.mirrored import MirroredStrategy

    dev = torch.device("cuda:0")
    try:
        def run():
            torch.manual_seed(0)
            if model_name == "resnet20":
                from distributed_tensorflow_amd.models.resnet import resnet20
                model, cfg = resnet20(10), TrainConfig(param_dtype=torch.bfloat16, grad_dtype=torch.bfloat16)
                g = torch.Generator().manual_seed(3)
                x = torch.randn(64, 32, 32, 3, generator=g).to(dev).bfloat16()
                y = torch.randint(0, 10, (64,), generator=g).to(dev)
            else:
                from distributed_tensorflow_amd.models.bert import bert_tiny, synthetic_mlm_batch
                model = bert_tiny(64)
                cfg = TrainConfig(loss="mlm", param_dtype=torch.bfloat16, grad_dtype=torch.bfloat16)
                x, y = synthetic_mlm_batch(8, 64, 1024, dev, n_pred=10, seed=3)
            st = MirroredStrategy(model, Comm(0, 1, dev), cfg, optimizer=Adam(1e-3))
            losses = [float(st.train_step(x, y)) for _ in range(4)]
            torch.cuda.synchronize()
            # the 3-channel stem cannot run on the own kernels; the vendor library's weight gradient for it
            # accumulates with atomics and differs in the last bit from run to run
            own = {s.name: st.flat.master[s.offset:s.offset + s.numel].clone() for s in st.flat.slots
                   if s.name != "stem.weight"}
            return losses, own
        l1, w1 = run()
        l2, w2 = run()
    finally:
        pass
    assert l1 == l2
    assert w1.keys() == w2.keys()
    for k in w1:
        assert torch.equal(w1[k], w2[k]), k


def test_weight_gradients_on_the_side_stream_change_nothing():
    """Weight gradients run on a second stream between the reducer's begin_step() and finish(); a training run
    with them there must equal the run with everything on one stream, bit for bit."""
    from distributed_tensorflow_amd.comm.backend import Comm
    from distributed_tensorflow_amd.engine.optimizer import SGD
    from distributed_tensorflow_amd.engine.replica import TrainConfig
    from distributed_tensorflow_amd.models.resnet import resnet20
    from distributed_tensorflow_amd.ops import conv
    from distributed_tensorflow_amd.parallel.mirrored import MirroredStrategy

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    batches = [(torch.randn(96, 32, 32, 3, generator=g).to(dev).bfloat16(), torch.randint(0, 10, (96,), generator=g).to(dev))
               for _ in range(6)]
    results = []
    was, was_min = conv.side_grads.enabled, conv._SIDE_MIN_ELEMS
    conv._SIDE_MIN_ELEMS = 0          # every weight gradient, however small
    try:
        for side in (False, True):
            conv.side_grads.enabled = side
            torch.manual_seed(0)
            cfg = TrainConfig(param_dtype=torch.bfloat16, grad_dtype=torch.bfloat16)
            st = MirroredStrategy(resnet20(10), Comm(0, 1, dev), cfg, optimizer=SGD(0.05, momentum=0.9))
            losses = [float(st.train_step(x, y)) for x, y in batches]
            torch.cuda.synchronize()
            results.append((losses, st.flat.master.clone()))
            if side:
                assert conv.side_grads._streams, "the side stream was never used"
                assert not conv.side_grads._held and not conv.side_grads._events      # everything handed back
    finally:
        conv.side_grads.enabled, conv._SIDE_MIN_ELEMS = was, was_min
    assert results[0][0] == results[1][0]
    assert torch.equal(results[0][1], results[1][1])
