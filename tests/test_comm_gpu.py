"""GPU tests of the native communication layer on ONE device: the RCCL binding at world size 1,
the peer-pointer kernels with virtual ranks, and the HIP-IPC peer group with several processes
that share the GPU (kernel logic, flags and schedules are exercised without xGMI)."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


# ----------------------------------------------------------------------------- RCCL binding
def test_rccl_binding_world1_collectives():
    from distributed_tensorflow_amd.comm.rccl import RcclComm, _rt

    assert _rt().dtf_rccl_version() > 0
    comm = RcclComm.create(0, 1, _dev())
    try:
        x = torch.randn(1 << 20, device=_dev())
        ref = x.clone()
        w = comm.all_reduce(x, async_op=True)
        w.wait()
        torch.cuda.synchronize()
        assert torch.equal(x, ref)
        xb = torch.randn(4096, device=_dev()).bfloat16()
        refb = xb.clone()
        comm.all_reduce(xb)
        out = torch.empty_like(xb)
        comm.reduce_scatter(out, xb)
        comm.all_gather(xb, out)
        comm.broadcast(xb, 0)
        comm.reduce(xb, 0)
        a2a = torch.empty_like(xb)
        comm.all_to_all(a2a, xb)
        torch.cuda.synchronize()
        assert torch.equal(xb, refb) and torch.equal(out, refb) and torch.equal(a2a, refb)
        comm.barrier()
        comm.check_async_error()
        assert comm.async_error() is None
        assert comm.all_reduce_scalar(3.0) == 3.0
        # sub-communicator by ncclCommSplit: a working communicator of its own
        sub = comm.subgroup([0])
        assert (sub.rank, sub.world) == (0, 1)
        y = torch.randn(4096, device=_dev())
        ry = y.clone()
        sub.all_reduce(y)
        sub.barrier()
        torch.cuda.synchronize()
        assert torch.equal(y, ry)
        sub.shutdown()
    finally:
        comm.shutdown()


# ------------------------------------------------------------------- kernels, virtual ranks
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("algorithm", ["direct", "oneshot", "ring", "tree", "fused"])
@pytest.mark.parametrize("world", [2, 3, 8])
def test_peer_pointer_kernels_virtual_ranks(algorithm, world, dtype):
    from distributed_tensorflow_amd.comm.ipc import LocalPeers

    torch.manual_seed(world)
    for n in (8, 1000, 262147):
        bufs = [torch.randn(n, device=_dev()).to(dtype) for _ in range(world)]
        want = sum(b.double() for b in bufs) * 0.25
        peers = LocalPeers(bufs)
        peers.all_reduce(algorithm, scale=0.25)
        torch.cuda.synchronize()
        if algorithm == "fused":
            assert peers.fused_status() == [0] * world      # no barrier of the single-launch kernel timed out
        tol = 1e-5 if dtype == torch.float32 else 0.05 * world ** 0.5
        for r, b in enumerate(bufs):
            assert torch.allclose(b.double(), want, atol=tol, rtol=1e-2 if dtype == torch.bfloat16 else 1e-5), \
                (algorithm, world, n, r, float((b.double() - want).abs().max()))
        if algorithm in ("direct", "oneshot", "fused"):   # fixed summation order: every rank holds the same bits
            for b in bufs[1:]:
                assert torch.equal(b, bufs[0])


def test_peer_pointer_region_of_a_larger_buffer():
    from distributed_tensorflow_amd.comm.ipc import LocalPeers

    world, n = 4, 5000
    bufs = [torch.randn(n, device=_dev()) for _ in range(world)]
    keep = [b.clone() for b in bufs]
    want = sum(b[1024:1024 + 3001].double() for b in bufs)
    LocalPeers(bufs).all_reduce("direct", start=1024, numel=3001)
    torch.cuda.synchronize()
    for b, k in zip(bufs, keep):
        assert torch.allclose(b[1024:4025].double(), want, atol=1e-5)
        assert torch.equal(b[:1024], k[:1024]) and torch.equal(b[4025:], k[4025:])


# ----------------------------------------------------------- several processes, one device
def _ipc_worker(rank, world, port, q):
    try:
        os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
        from distributed_tensorflow_amd.comm.backend import Comm
        from distributed_tensorflow_amd.comm.ipc import PeerGroup

        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        comm = Comm.init(rank, world, dev, timeout_s=60, backend="gloo", port_cell=port)
        pg = PeerGroup(comm, barrier_budget_ms=10000)
        torch.manual_seed(100 + rank)
        arena = torch.randn(300000, device=dev)
        mine = arena.clone()
        pg.register(arena)
        res = {}
        for alg, lo, n in (("direct", 0, 300000), ("oneshot", 1024, 4099), ("ring", 8192, 100003),
                           ("tree", 0, 777), ("auto", 2048, 262144), ("fused", 0, 300000), ("fused", 4096, 70003)):
            arena.copy_(mine)
            torch.cuda.synchronize()
            pg.all_reduce(arena[lo:lo + n], alg, scale=1.0 / world, base=arena)
            pg.check()
            # reference through gloo on the host
            ref = mine[lo:lo + n].cpu()
            torch.distributed.all_reduce(ref)
            ref /= world
            res[alg] = float((arena[lo:lo + n].cpu() - ref).abs().max())
            outside = torch.equal(arena[:lo], mine[:lo]) and torch.equal(arena[lo + n:], mine[lo + n:])
            res[alg + "_outside_untouched"] = bool(outside)
        pg.close()
        comm.shutdown()
        q.put((rank, res))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, {"error": f"{e}\n{traceback.format_exc()}"}))


@pytest.mark.parametrize("world", [2, 4])
def test_peer_group_processes_sharing_one_gpu(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = mp.get_context("spawn").Value("i", 0)      # rank 0 publishes the rendezvous port it bound
    procs = [ctx.Process(target=_ipc_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    try:
        for _ in range(world):
            r, res = q.get(timeout=100)
            got[r] = res
    finally:
        for p in procs:
            p.join(timeout=20)
            if p.is_alive():
                p.kill()
    for r, res in got.items():
        assert "error" not in res, res.get("error")
        for k, v in res.items():
            if k.endswith("_outside_untouched"):
                assert v, (r, k)
            else:
                assert v < 1e-5, (r, k, v)


# ------------------------------------- whole strategy on peer-pointer all-reduce, ranks share the GPU
def _ipc_strategy_worker(rank, world, port, q):
    try:
        os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
        from distributed_tensorflow_amd.comm.backend import Comm
        from distributed_tensorflow_amd.engine.optimizer import Adam
        from distributed_tensorflow_amd.engine.replica import TrainConfig
        from distributed_tensorflow_amd.models.mlp import reference_mlp
        from distributed_tensorflow_amd.parallel.mirrored import CustomAllReduceStrategy

        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        comm = Comm.init(rank, world, dev, timeout_s=60, backend="gloo", port_cell=port)
        torch.manual_seed(0)
        st = CustomAllReduceStrategy(reference_mlp(dropout=0.0), comm, TrainConfig(loss_reduction="mean", seed=0),
                                     optimizer=Adam(1e-3), algorithm="ipc-direct")
        g = torch.Generator().manual_seed(7)
        xs = torch.rand(6, 32 * world, 28, 28, 1, generator=g)
        ys = torch.randint(0, 10, (6, 32 * world), generator=g)
        for i in range(6):
            x = xs[i, rank * 32:(rank + 1) * 32].to(dev)
            y = ys[i, rank * 32:(rank + 1) * 32].to(dev)
            st.train_step(x, y)
        torch.cuda.synchronize()
        st._peers.check()
        used = sorted(set(st.used_algorithms))
        import numpy as np
        w = np.concatenate([a.reshape(-1) for a in st.replica.model.get_weights()])
        q.put((rank, {"w": w, "used": used}))
        st._peers.close()
        comm.shutdown()
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, {"error": f"{e}\n{traceback.format_exc()}"}))


def test_custom_strategy_on_peer_pointer_allreduce_matches_single_process():
    """Two ranks that share the GPU train with the hand-written IPC all-reduce; the result equals one
    process training on the concatenated batch (mean loss, so the averaged gradients coincide)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    cell = ctx.Value("i", 0)
    procs = [ctx.Process(target=_ipc_strategy_worker, args=(r, world, cell, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    try:
        for _ in range(world):
            r, res = q.get(timeout=110)
            got[r] = res
    finally:
        for p in procs:
            p.join(timeout=20)
            if p.is_alive():
                p.kill()
    for r, res in got.items():
        assert "error" not in res, res.get("error")
    import numpy as np
    assert np.array_equal(got[0]["w"], got[1]["w"])          # replicas stay bit-identical
    assert all(u.startswith("ipc-") for u in got[0]["used"])

    from distributed_tensorflow_amd.comm.backend import Comm
    from distributed_tensorflow_amd.engine.optimizer import Adam
    from distributed_tensorflow_amd.engine.replica import TrainConfig
    from distributed_tensorflow_amd.models.mlp import reference_mlp
    from distributed_tensorflow_amd.parallel.mirrored import MirroredStrategy
    dev = _dev()
    torch.manual_seed(0)
    ref = MirroredStrategy(reference_mlp(dropout=0.0), Comm(0, 1, dev), TrainConfig(loss_reduction="mean", seed=0),
                           optimizer=Adam(1e-3))
    g = torch.Generator().manual_seed(7)
    xs = torch.rand(6, 32 * world, 28, 28, 1, generator=g)
    ys = torch.randint(0, 10, (6, 32 * world), generator=g)
    for i in range(6):
        ref.train_step(xs[i].to(dev), ys[i].to(dev))
    w_ref = np.concatenate([a.reshape(-1) for a in ref.replica.model.get_weights()])
    assert np.allclose(got[0]["w"], w_ref, atol=2e-5, rtol=1e-4), float(np.abs(got[0]["w"] - w_ref).max())


# --------------------- every collective strategy with two ranks that share the GPU (gloo carries GPU tensors)
_KINDS = ["keras", "custom-ring", "sync", "async", "graph"]


def _strategy_worker(rank, world, port, kinds, q):
    """One rank; runs every strategy in turn over ONE process group (a pair of processes for all five keeps the
    number of GPU contexts that come and go on the shared device small)."""
    import faulthandler
    dump_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    dump = open(os.path.join(dump_dir, f"hang_strategies_{rank}.txt"), "w") if os.path.isdir(dump_dir) else sys.stderr
    faulthandler.dump_traceback_later(150, exit=True, file=dump)      # a hung rank reports where and ends
    out = {}
    try:
        import numpy as np
        from distributed_tensorflow_amd.comm.backend import Comm
        from distributed_tensorflow_amd.engine.optimizer import Adam
        from distributed_tensorflow_amd.engine.replica import TrainConfig
        from distributed_tensorflow_amd.models.mlp import reference_mlp
        from distributed_tensorflow_amd.parallel import mirrored, ps

        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        comm = Comm.init(rank, world, dev, timeout_s=60, backend="gloo", port_cell=port)
        g = torch.Generator().manual_seed(7)
        xs = torch.rand(6, 32 * world, 28, 28, 1, generator=g)
        ys = torch.randint(0, 10, (6, 32 * world), generator=g)
        for kind in kinds:
            try:
                torch.manual_seed(0)
                cfg = TrainConfig(loss_reduction="mean", seed=0, bucket_bytes=256 << 10, first_bucket_bytes=64 << 10)
                model = reference_mlp(dropout=0.0)
                if kind == "keras":
                    st = mirrored.MirroredStrategy(model, comm, cfg, optimizer=Adam(1e-3))
                elif kind == "custom-ring":
                    st = mirrored.CustomAllReduceStrategy(model, comm, cfg, optimizer=Adam(1e-3), algorithm="ring")
                elif kind == "sync":
                    st = ps.SyncPSStrategy(model, comm, cfg, optimizer=Adam(1e-3))
                elif kind == "async":
                    st = ps.AsyncPSStrategy(model, comm, cfg, optimizer=Adam(1e-3))
                else:
                    st = mirrored.GraphStrategy(model, comm, cfg, optimizer=Adam(1e-3), degree=1)
                losses = []
                for i in range(6):
                    x = xs[i, rank * 32:(rank + 1) * 32].to(dev)
                    y = ys[i, rank * 32:(rank + 1) * 32].to(dev)
                    losses.append(float(st.train_step(x, y)))
                st.finalize()
                torch.cuda.synchronize()
                w = np.concatenate([a.reshape(-1) for a in st.replica.model.get_weights()])
                out[kind] = {"w": w, "losses": losses}
                comm.barrier()
            except Exception as e:  # pragma: no cover
                import traceback
                out[kind] = {"error": f"{e}\n{traceback.format_exc()}"}
                break               # the ranks are out of step from here on
        q.put((rank, out))
    except Exception as e:  # pragma: no cover
        import traceback
        out["init"] = {"error": f"{e}\n{traceback.format_exc()}"}
        q.put((rank, out))
        return
    try:                    # the results are out: a slow peer at tear-down is not a test failure
        comm.barrier()
        comm.shutdown()
    except Exception:  # pragma: no cover
        pass


@pytest.fixture(scope="module")
def two_rank_results():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    cell = ctx.Value("i", 0)
    procs = [ctx.Process(target=_strategy_worker, args=(r, world, cell, _KINDS, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    try:
        while len(got) < world:
            r, res = q.get(timeout=170)
            got.setdefault(r, res)
    finally:
        for p in procs:
            p.join(timeout=20)
            if p.is_alive():
                p.kill()
    return got


@pytest.mark.parametrize("kind", _KINDS)
def test_collective_strategies_two_ranks_on_one_gpu(kind, two_rank_results):
    import numpy as np
    world = 2
    for r in range(world):
        assert "init" not in two_rank_results[r], two_rank_results[r]["init"]["error"]
        assert kind in two_rank_results[r], f"rank {r} stopped before {kind}: {list(two_rank_results[r])}"
    got = {r: two_rank_results[r][kind] for r in range(world)}
    for r, res in got.items():
        assert "error" not in res, res.get("error")
    assert np.allclose(got[0]["w"], got[1]["w"], atol=1e-6)          # replicas agree after finalize()
    assert all(np.isfinite(got[r]["losses"]).all() for r in got)
    if kind in ("keras", "custom-ring", "sync"):
        # synchronous strategies reproduce single-process training on the concatenated batch
        from distributed_tensorflow_amd.comm.backend import Comm
        from distributed_tensorflow_amd.engine.optimizer import Adam
        from distributed_tensorflow_amd.engine.replica import TrainConfig
        from distributed_tensorflow_amd.models.mlp import reference_mlp
        from distributed_tensorflow_amd.parallel.mirrored import MirroredStrategy
        dev = _dev()
        torch.manual_seed(0)
        ref = MirroredStrategy(reference_mlp(dropout=0.0), Comm(0, 1, dev), TrainConfig(loss_reduction="mean", seed=0),
                               optimizer=Adam(1e-3))
        g = torch.Generator().manual_seed(7)
        xs = torch.rand(6, 32 * world, 28, 28, 1, generator=g)
        ys = torch.randint(0, 10, (6, 32 * world), generator=g)
        for i in range(6):
            ref.train_step(xs[i].to(dev), ys[i].to(dev))
        w_ref = np.concatenate([a.reshape(-1) for a in ref.replica.model.get_weights()])
        assert np.allclose(got[0]["w"], w_ref, atol=3e-5, rtol=1e-4), float(np.abs(got[0]["w"] - w_ref).max())


@pytest.mark.parametrize("transport,world", [("ipc", 4), ("ipc", 8), ("p2p", 4)])
def test_async_ps_loses_no_update_with_several_service_threads_per_owner_on_the_gpu(transport, world):
    """Four / eight ranks share the GPU: every owner serves the other ranks from one host thread each, all enqueueing
    onto its one apply stream next to the owner's own applies. ``ipc``: the optimizer kernel reads each worker's
    gradient through the HIP-IPC peer pointer and the workers pull the replies the same way (no transfer kernel
    anywhere); ``p2p``: send-first transfers on gloo. Plain SGD with a power-of-two learning rate on integer gradients
    is exact in fp32, so after world * steps applies per shard the parameters must equal init - lr * sum(gradients)
    bit for bit; a lost update, a doubled one, a torn reply or a stale cached read of peer memory would show."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_distributed_cpu import run_exact_async
    steps = 50 if world == 4 else 25
    got = run_exact_async(world, "cuda:0", steps, transport)
    for r in range(world):
        assert "error" not in got[r], got[r].get("error")
        assert got[r]["transport"] == transport
        assert got[r]["applies"] == world * steps
        assert got[r]["equal"], got[r]


# ----------------------------------------------------------------------------- more ranks than GPUs
@pytest.mark.parametrize("flags", [["-m", "d", "-ds", "keras"], ["-m", "c", "-cs", "sync"]], ids=["keras", "sync"])
def test_launcher_with_more_ranks_than_gpus_shares_devices_over_gloo(flags):
    """The reference runs every role as a local process (reference initializer.py:134-145,164-173); with more
    ranks than cards RCCL refuses the communicator ("Duplicate GPU detected"), so the launcher lets the ranks share
    the devices and moves the collectives through host memory. The job must finish with exit code 0."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = torch.cuda.device_count() + 1
    if n > 9:
        pytest.skip("a node with more than 8 GPUs: the case needs more processes than the box allows")
    env = dict(os.environ, DTF_SYNTH_MNIST_TRAIN="2000", DTF_SYNTH_MNIST_TEST="500")
    env.pop("DTF_COMM", None)
    r = subprocess.run([sys.executable, os.path.join(root, "examples", "mnist_cnn.py"), *flags, "-n", str(n), "-b", "50",
                        "--max-steps", "6", "--comm-timeout-s", "120"], env=env, capture_output=True, text=True,
                       timeout=170)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "ranks share devices" in r.stderr
