"""Launcher: the ``initializer.py`` contract of the reference.

Flags (reference initializer.py:72-112): ``-m`` centralized|c|decentralized|d, ``-cs`` sync|async,
``-ds`` graph|keras|custom, ``-n`` number of WORKER nodes (the parameter server is an extra
process in the TCP topology, reference initializer.py:134-145), ``-b`` per-worker batch
(default 100), ``-tt`` server|worker and ``-ti`` index and ``-sa`` address for multi-machine
centralized runs, ``-ca`` pin every process to one core. ``-d`` (graph degree) is the flag the
reference left commented out (initializer.py:90-92).

Dispatch (reference initializer.py:128-184) with the differences this framework needs:

* processes are started with the ``spawn`` method (a HIP context does not survive ``fork``),
  joined, and their exit codes propagated -- the reference never joins its children;
* ``--transport`` picks how a centralized run moves tensors: ``tcp`` is the reference's
  topology (1 server + n workers over framed TCP, the CPU plumbing mode), ``collective`` is
  the device-resident sharded parameter server on RCCL (``gloo`` on CPU) with n ranks;
* ``graph`` and ``custom`` are implemented instead of raising ``NotImplementedError``;
* an invalid ``-m`` or ``-cs`` raises ``ValueError`` exactly like the reference
  (initializer.py:158,184); an invalid ``-ds`` does too (the reference silently does nothing).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import sys
import time
import traceback
from typing import Callable

import torch
import torch.multiprocessing as mp


def str2bool(v):
    """argparse helper with the reference's accepted spellings (reference initializer.py:59-67)."""
    if isinstance(v, bool):
        return v
    if v.lower() in ("yes", "true", "t", "y", "1"):
        return True
    if v.lower() in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("Boolean value expected.")


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="MI355X-native distributed data-parallel training launcher")
    p.add_argument("-m", action="store", dest="mode", default=None,
                   help="training mode: centralized | decentralized (c/d)")
    p.add_argument("-cs", action="store", dest="centralized_strategy", default=None,
                   help="centralized strategy: sync | async")
    p.add_argument("-ds", action="store", dest="decentralized_strategy", default=None,
                   help="decentralized strategy: graph | keras | custom")
    p.add_argument("-n", action="store", dest="n_nodes", type=int, help="number of worker nodes on the network")
    p.add_argument("-d", action="store", dest="degree", type=int, default=2,
                   help="degree of the network (only used if strategy == graph)")
    p.add_argument("-b", action="store", dest="batch_size", type=int, default=100, help="batch size")
    p.add_argument("-tt", action="store", dest="task_type", default=None,
                   help="node task type, only used on centralized parameter server (server/worker)")
    p.add_argument("-ti", action="store", dest="task_index", type=int, default=None,
                   help="node task index, only used on centralized parameter server")
    p.add_argument("-sa", action="store", dest="server_address", default="127.0.0.1",
                   help="server address, only used on centralized parameter server")
    p.add_argument("-ca", action="store", dest="cpu_affinity", type=str2bool, default=False,
                   help="restrain each process to only 1 cpu core (y/n)")
    # ---- extensions (not in the reference)
    p.add_argument("--transport", choices=["auto", "tcp", "collective"], default="auto",
                   help="centralized tensor transport: tcp (1 server + n workers) | collective (sharded PS on RCCL/gloo)")
    p.add_argument("--device", choices=["auto", "cpu", "gpu"], default="auto")
    p.add_argument("--epochs", type=int, default=1)
    p.add_argument("--max-steps", type=int, default=None, help="stop each epoch after this many steps")
    p.add_argument("--apply", choices=["reduced", "sequential"], default="reduced",
                   help="sync PS: one step on the reduced gradient | one apply per worker (reference behaviour)")
    p.add_argument("--loss-reduction", choices=["mean", "sum"], default=None,
                   help="default: sum for centralized (reference semantics), mean for decentralized")
    p.add_argument("--placement", choices=["colocated", "dedicated"], default="colocated",
                   help="async collective PS: shard owners are the workers | one extra server rank")
    p.add_argument("--ps-data-plane", choices=["auto", "ipc", "p2p"], default="auto",
                   help="async collective PS: how gradients and parameters move -- ipc: pull-only over HIP-IPC peer pointers "
                        "(GPUs of one host, the default there), p2p: send-first transfers on pair communicators")
    p.add_argument("--algorithm", default="auto", help="custom strategy: auto|direct|ring|multiring|tree|oneshot on grouped send/recv, or "
                   "ipc|ipc-direct|ipc-fused|ipc-oneshot|ipc-ring|ipc-tree: kernels on HIP IPC mapped peer arenas (ipc-fused: one launch per bucket)")
    p.add_argument("--dtype", choices=["fp32", "bf16"], default="fp32")
    p.add_argument("--optimizer", choices=["adam", "sgd", "momentum"], default="adam",
                   help="the optimizer object the reference's server.start(optimizer=...) accepts (default: Keras Adam)")
    p.add_argument("--learning-rate", type=float, default=None, help="default: the optimizer's Keras default")
    p.add_argument("--port", type=int, default=6000,
                   help="parameter-server port (reference: 6000); 0 = one the operating system picks (single machine)")
    p.add_argument("--metrics", default=None, help="JSONL metrics path (%%r = rank)")
    p.add_argument("--results", default=None, help="write a JSON summary of the run here")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--graph", action="store_true",
                   help="capture the train step into a hipGraph (GPU; keras and collective sync strategies)")
    p.add_argument("--verify-replicas", type=int, default=0, metavar="N",
                   help="debug: every N steps compare the master-weight checksum across ranks (synchronous strategies)")
    p.add_argument("--fault-inject", default=None,
                   help="RANK:STEP -- kill that worker at that step; RANK:STEP:hang -- make it stall there (testing)")
    p.add_argument("--checkpoint-dir", default=None,
                   help="directory of the run's checkpoint (weights, optimizer state per rank, RNG state, step)")
    p.add_argument("--checkpoint-every", type=int, default=0, metavar="N",
                   help="write the checkpoint every N steps (it is always written at the end of training)")
    p.add_argument("--resume", action="store_true", help="continue from the checkpoint in --checkpoint-dir")
    p.add_argument("--save-weights", default=None, metavar="PATH",
                   help="after training write the model as a Keras-ordered weight list (.npz, or .h5 with h5py)")
    p.add_argument("--load-weights", default=None, metavar="PATH",
                   help="initialise the model from a Keras-ordered weight list before training")
    p.add_argument("--supervisor-address", default=None,
                   help="host of a supervisor listening on port 4000 for 'start' / 'done' / ('results', acc) frames")
    p.add_argument("--watchdog-s", type=float, default=0.0, metavar="SECONDS",
                   help="abort a rank (exit code 19) that makes no training progress for this long; 0 = off")
    p.add_argument("--comm-timeout-s", type=float, default=300.0, help="timeout of every collective / rendezvous")
    p.add_argument("--vram-pretouch", type=str2bool, default=False,
                   help="before a rank initialises its GPU, a throwaway child process writes the device's free memory "
                        "once (utils/vram.py: the first process to work through the HBM of a freshly booted machine "
                        "runs 1-1.5 %% slower for as long as it lives); one rank per GPU only, a few seconds per launch")
    return p


# ----------------------------------------------------------------------------- helpers
def _gpu_count() -> int:
    try:
        return torch.cuda.device_count()
    except Exception:
        return 0


def _device_for(local_rank: int, want: str) -> torch.device:
    n = _gpu_count()
    if want == "cpu" or (want == "auto" and n == 0):
        return torch.device("cpu")
    if n == 0:
        raise RuntimeError("--device gpu requested but no GPU is visible")
    return torch.device("cuda", local_rank % n)


def _fault_hook(spec: str | None):
    if not spec:
        return None
    parts = spec.split(":")
    rank, step = int(parts[0]), int(parts[1])
    hang = len(parts) > 2 and parts[2] == "hang"

    def hook(r, s):
        if r == rank and s == step:
            if hang:
                time.sleep(3600)
            os._exit(17)
    return hook


def _train_config(args, centralized: bool, shard_ways: int = 1):
    from ..engine.replica import TrainConfig
    red = args.loss_reduction or ("sum" if centralized else "mean")
    return TrainConfig(loss_reduction=red, param_dtype=torch.bfloat16 if args.dtype == "bf16" else None,
                       shard_ways=shard_ways, seed=args.seed, graph=bool(getattr(args, "graph", False)),
                       verify_replicas=int(getattr(args, "verify_replicas", 0) or 0))


def _optimizer_spec(args):
    """The optimizer description handed to every strategy (reference centralized/server.py:42,52-55: an optional
    optimizer object, Adam by default)."""
    from ..engine.optimizer import resolve_spec
    spec = resolve_spec(getattr(args, "optimizer", None) or "adam")
    lr = getattr(args, "learning_rate", None)
    if lr is not None:
        spec.learning_rate = float(lr)
    return spec


# ----------------------------------------------------------------------------- rank bodies
def _collective_rank(rank: int, world: int, args, model_fn, dataset_fn, kind: str, port, queue) -> None:
    """One rank of a collective (RCCL / gloo) strategy."""
    try:
        from ..comm.backend import Comm
        from ..obs.metrics import MetricsLogger
        from ..parallel import mirrored, ps
        from ..parallel.ps_tcp import call_dataset_fn
        from ..utils.affinity import pin_to_core

        dedicated = kind == "async" and args.placement == "dedicated"
        if args.cpu_affinity:
            pin_to_core(rank)
        dev = _device_for(rank, args.device)
        if dev.type == "cuda" and getattr(args, "vram_pretouch", False) and world <= _gpu_count():
            from ..utils.vram import pretouch
            pretouch(dev.index or 0)
        if dev.type == "cuda":
            torch.cuda.set_device(dev)
        from ..comm.watchdog import Watchdog, configure_async_errors
        from ..obs.metrics import SupervisorSink
        configure_async_errors()
        # ``port``: a port number (ranks started by hand, one command each), or the launcher's shared cell -- rank 0
        # then binds an ephemeral rendezvous port and publishes it there
        by_hand = isinstance(port, int)
        # More ranks than GPUs on this machine (the reference's habit of running every role as a local process,
        # reference initializer.py:134-145,164-173, on a box with fewer cards than roles): RCCL refuses two ranks of
        # one communicator on one device, so the ranks share the cards and gloo carries the collectives through
        # host memory. Correct, not fast; ranks started by hand on several machines are taken at their word.
        backend = None
        if dev.type == "cuda" and not by_hand and world > _gpu_count() and not os.environ.get("DTF_COMM"):
            backend = "gloo"
            if rank == 0:
                print(f"{world} ranks on {_gpu_count()} GPU(s): ranks share devices, collectives go through host "
                      "memory (gloo) instead of RCCL", file=sys.stderr)
        comm = Comm.init(rank, world, dev, getattr(args, "master_addr", None) or "127.0.0.1", port if by_hand else 0,
                         timeout_s=float(getattr(args, "comm_timeout_s", 300.0) or 300.0), backend=backend,
                         port_cell=None if by_hand else port)
        metrics = MetricsLogger(args.metrics, rank=rank)
        sup = SupervisorSink(getattr(args, "supervisor_address", None))
        cfg = _train_config(args, centralized=kind in ("sync", "async"))
        torch.manual_seed(args.seed)          # the initial weights are a function of --seed
        model = model_fn()
        opt = _optimizer_spec(args)
        if kind == "keras":
            st = mirrored.MirroredStrategy(model, comm, cfg, optimizer=opt, metrics=metrics)
        elif kind == "custom":
            st = mirrored.CustomAllReduceStrategy(model, comm, cfg, optimizer=opt, metrics=metrics,
                                                  algorithm=args.algorithm)
        elif kind == "graph":
            st = mirrored.GraphStrategy(model, comm, cfg, optimizer=opt, metrics=metrics, degree=args.degree)
        elif kind == "sync":
            st = ps.SyncPSStrategy(model, comm, cfg, optimizer=opt, metrics=metrics, apply=args.apply)
        elif kind == "async":
            st = ps.AsyncPSStrategy(model, comm, cfg, optimizer=opt, metrics=metrics, placement=args.placement,
                                    transport=getattr(args, "ps_data_plane", "auto"))
        else:
            raise ValueError(kind)
        if getattr(args, "load_weights", None):
            st.replica.model.load_weights(args.load_weights)      # every rank reads the same file
            st.flat.refresh_shadow()
            st.on_weights_loaded()
        wd_s = float(getattr(args, "watchdog_s", 0.0) or 0.0)
        if wd_s > 0:
            st.watchdog = Watchdog(wd_s, rank=rank, poll=getattr(comm, "async_error", None))
        n_workers = world - 1 if dedicated else world
        widx = rank - 1 if dedicated else rank
        reshape = kind in ("sync", "async")  # the reference reshapes to NHWC only in the PS path
        res = {"steps": 0}
        if not dedicated or rank > 0:
            train_ds = call_dataset_fn(dataset_fn, args.batch_size, n_nodes=n_workers, type="train", shard=True,
                                       index=widx, reshape=reshape)
        else:
            train_ds = None
        sup.send("start")             # reference dist_keras.py:34-39, server.py:155-158
        if train_ds is not None or dedicated:
            res = st.run_epochs(train_ds, epochs=args.epochs, verbose=1, max_steps=args.max_steps,
                                fault_hook=_fault_hook(args.fault_inject),
                                checkpoint_dir=getattr(args, "checkpoint_dir", None),
                                checkpoint_every=int(getattr(args, "checkpoint_every", 0) or 0),
                                resume=bool(getattr(args, "resume", False)))
        if st.watchdog is not None:
            st.watchdog.pause()
        if getattr(args, "save_weights", None) and rank == 0:
            st.replica.model.save_weights(args.save_weights)
        # the mirrored worker reports 'done', the parameter server ('done', elapsed)
        # (reference dist_keras.py:45-47, server.py:121-124)
        sup.send(["done", res.get("elapsed")] if kind in ("sync", "async") else "done")
        if kind in ("sync", "async"):
            # the parameter server evaluates the unsharded test split (reference server.py:176-180)
            acc = None
            if rank == 0:
                test_ds = call_dataset_fn(dataset_fn, 100, n_nodes=n_workers, type="test", shard=False)
                _, acc = st.evaluate(test_ds, reduce=False)
                print("elapsed time = " + str(res["elapsed"]), flush=True)
                print("Test set accuracy: {:.3%}".format(acc), flush=True)
            res["test_accuracy"] = acc
        else:
            # every mirrored worker evaluates its shard of the test split (reference dist_keras.py:25,53)
            test_ds = call_dataset_fn(dataset_fn, args.batch_size, n_nodes=world, type="test", shard=True,
                                      index=rank, reshape=False)
            print("-----------------------------------------", flush=True)
            print("elapsed time: " + str(res["elapsed"]) + " seconds", flush=True)
            print("-----------------------------------------", flush=True)
            _, acc = st.evaluate(test_ds, reduce=True)
            print("eval: " + str(acc), flush=True)
            res["test_accuracy"] = acc
        if res.get("test_accuracy") is not None:
            sup.send(["results", res["test_accuracy"]])        # reference dist_keras.py:56-58, server.py:182-187
        sup.close()
        if st.watchdog is not None:
            st.watchdog.close()
        metrics.summary(strategy=kind, **{k: v for k, v in res.items() if not isinstance(v, torch.Tensor)})
        metrics.close()
        comm.barrier()
        comm.shutdown()
        if queue is not None:
            queue.put((f"rank{rank}", {"ok": True, **res}))
    except BaseException as e:
        traceback.print_exc()
        if queue is not None:
            queue.put((f"rank{rank}", {"ok": False, "error": repr(e)}))
        sys.exit(1)


def _tcp_server_extras(args) -> dict:
    """Persistence and time limits of the framed-TCP topology: the server owns the model, so it is the one that
    imports / exports weights and writes / resumes checkpoints."""
    t = float(getattr(args, "comm_timeout_s", 300.0) or 300.0)
    return dict(load_weights=getattr(args, "load_weights", None), save_weights=getattr(args, "save_weights", None),
                checkpoint_dir=getattr(args, "checkpoint_dir", None),
                checkpoint_every=int(getattr(args, "checkpoint_every", 0) or 0),
                resume=bool(getattr(args, "resume", False)), barrier_timeout=t, accept_timeout=max(t, 120.0),
                optimizer=_explicit_optimizer(args))


def _explicit_optimizer(args):
    """None unless the command line names an optimizer or a learning rate: with None the TCP server keeps the
    reference's default, one private Adam per connection under apply='sequential' (server.py:52-55,170)."""
    if getattr(args, "learning_rate", None) is None and (getattr(args, "optimizer", None) or "adam") == "adam":
        return None
    return _optimizer_spec(args)


def _tcp_worker_extras(args) -> dict:
    return dict(checkpoint_dir=getattr(args, "checkpoint_dir", None), resume=bool(getattr(args, "resume", False)),
                watchdog_s=float(getattr(args, "watchdog_s", 0.0) or 0.0),
                io_timeout=float(getattr(args, "comm_timeout_s", 300.0) or 300.0))


def _tcp_server(args, model_fn, dataset_fn, queue, ready, cell=None) -> None:
    try:
        from ..parallel import ps_tcp
        dev = _device_for(0, args.device)
        ps_tcp.start(args.centralized_strategy, model_fn, dataset_fn, args.n_nodes, cpu_affinity=args.cpu_affinity,
                     server_address=args.server_address, port=args.port, apply=args.apply, device=dev,
                     cfg=_train_config(args, True), result_queue=queue, metrics_path=args.metrics, ready_event=ready,
                     port_cell=cell, supervisor_address=getattr(args, "supervisor_address", None),
                     **_tcp_server_extras(args))
    except BaseException:
        traceback.print_exc()
        sys.exit(1)


def _tcp_worker(index: int, args, model_fn, dataset_fn, queue, cell=None) -> None:
    try:
        from ..parallel import ps_tcp
        dev = _device_for(index, args.device)
        ps_tcp.train(args.batch_size, model_fn=model_fn, dataset_fn=dataset_fn, num_epochs=args.epochs, index=index,
                     n_nodes=args.n_nodes, cpu_affinity=args.cpu_affinity, server_address=args.server_address,
                     port=args.port, device=dev, cfg=_train_config(args, True), max_steps=args.max_steps,
                     result_queue=queue, fault_hook=_fault_hook(args.fault_inject), metrics_path=args.metrics,
                     port_cell=cell, **_tcp_worker_extras(args))
    except BaseException:
        traceback.print_exc()
        sys.exit(1)


# ----------------------------------------------------------------------------- dispatch
def _join_all(procs, queue, timeout: float | None = None) -> dict:
    """Joins children, collects their result records and propagates failure."""
    results = {}
    deadline = None if timeout is None else time.time() + timeout
    alive = list(procs)
    failed = False
    while alive:
        for p in list(alive):
            p.join(0.2)
            if p.exitcode is not None:
                alive.remove(p)
                if p.exitcode != 0:
                    failed = True
        while not queue.empty():
            k, v = queue.get()
            results[k] = v
        if failed:
            # a peer died: give the rest a moment to notice, then stop them
            t_end = time.time() + 20
            while alive and time.time() < t_end:
                for p in list(alive):
                    p.join(0.2)
                    if p.exitcode is not None:
                        alive.remove(p)
            for p in alive:
                p.terminate()
            alive = []
        if deadline is not None and time.time() > deadline:
            for p in alive:
                p.terminate()
            raise TimeoutError("launcher timed out waiting for its child processes")
    while not queue.empty():
        k, v = queue.get()
        results[k] = v
    results["exit_codes"] = [p.exitcode for p in procs]
    results["ok"] = all(c == 0 for c in results["exit_codes"])
    return results


def _spawn_collective(args, model_fn, dataset_fn, kind: str, world: int) -> dict:
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = ctx.Value("i", 0)
    procs = [ctx.Process(target=_collective_rank, args=(r, world, args, model_fn, dataset_fn, kind, port, queue))
             for r in range(world)]
    for p in procs:
        p.start()
    return _join_all(procs, queue)


def run(args, model_fn: Callable, dataset_fn: Callable) -> dict:
    """The dispatch tree of the reference's ``__main__`` (reference initializer.py:128-184)."""
    mode = args.mode
    transport = args.transport
    if transport == "auto":
        multi_machine = args.task_type is not None or args.task_index is not None
        transport = "tcp" if (multi_machine or _gpu_count() == 0 or args.device == "cpu") else "collective"

    if mode in ("centralized", "c"):
        cs = args.centralized_strategy
        if cs not in ("sync", "async"):
            raise ValueError("Invalid centralized strategy")
        if args.task_type is None and args.task_index is None:  # single machine
            if transport == "collective":
                world = args.n_nodes + (1 if (cs == "async" and args.placement == "dedicated") else 0)
                return _spawn_collective(args, model_fn, dataset_fn, cs, world)
            ctx = mp.get_context("spawn")
            queue = ctx.Queue()
            ready = ctx.Event()
            # --port 0: the server binds a port the operating system picks and publishes it to the workers
            cell = ctx.Value("i", 0) if args.port == 0 else None
            ps = ctx.Process(target=_tcp_server, args=(args, model_fn, dataset_fn, queue, ready, cell))
            ps.start()
            nodes = [ctx.Process(target=_tcp_worker, args=(i, args, model_fn, dataset_fn, queue, cell))
                     for i in range(args.n_nodes)]
            for p in nodes:
                p.start()
            return _join_all([ps] + nodes, queue)
        if args.task_type == "server":
            from ..parallel import ps_tcp
            res = ps_tcp.start(cs, model_fn, dataset_fn, args.n_nodes, cpu_affinity=args.cpu_affinity,
                               server_address=args.server_address, port=args.port, apply=args.apply,
                               device=_device_for(0, args.device), cfg=_train_config(args, True),
                               metrics_path=args.metrics, supervisor_address=getattr(args, "supervisor_address", None),
                               **_tcp_server_extras(args))
            res.pop("weights", None)
            return {"ok": True, "server": res}
        if args.task_type == "worker" and args.task_index is not None:
            from ..parallel import ps_tcp
            res = ps_tcp.train(args.batch_size, index=args.task_index, model_fn=model_fn, dataset_fn=dataset_fn,
                               num_epochs=args.epochs, n_nodes=args.n_nodes or 0, cpu_affinity=args.cpu_affinity,
                               server_address=args.server_address, port=args.port,
                               device=_device_for(args.task_index, args.device), cfg=_train_config(args, True),
                               max_steps=args.max_steps, metrics_path=args.metrics,
                               fault_hook=_fault_hook(args.fault_inject), **_tcp_worker_extras(args))
            return {"ok": True, f"worker{args.task_index}": res}
        raise ValueError("Invalid task type: use -tt server, or -tt worker with -ti INDEX")

    if mode in ("decentralized", "d"):
        ds = args.decentralized_strategy
        if ds not in ("keras", "graph", "custom"):
            raise ValueError("Invalid decentralized strategy")
        return _spawn_collective(args, model_fn, dataset_fn, ds, args.n_nodes)

    raise ValueError("Invalid mode")


def main(model_fn: Callable, dataset_fn: Callable, argv=None) -> int:
    args = build_parser().parse_args(argv)
    if args.n_nodes is None and args.task_type != "worker":
        raise ValueError("-n (number of worker nodes) is required")
    res = run(args, model_fn, dataset_fn)
    if args.results:
        with open(args.results, "w") as f:
            json.dump(res, f, indent=1, default=str)
    return 0 if res.get("ok", False) else 1
