"""Stand-alone role entry points: one process of a job, started by hand or by a cluster scheduler.

The reference ships a ``__main__`` block in each of its three runtime modules (parameter server
``centralized/server.py:197-219``, worker ``centralized/client.py:112-147``, mirrored worker
``decentralized/native/dist_keras.py:83-116``) with their own small flag sets; there they cannot run, because the
modules' ``model_fn`` / ``dataset_fn`` are stubs that raise ``NotImplementedError`` (server.py:190-195,
client.py:104-109) and dist_keras.py:116 passes its arguments in the wrong order. Here the same three commands work:

    python -m distributed_tensorflow_amd.cli.standalone server -n 2 -s 0.0.0.0 -sm sync
    python -m distributed_tensorflow_amd.cli.standalone worker -ps 0 -n 2 -b 64 -s 10.0.0.1
    python -m distributed_tensorflow_amd.cli.standalone keras  -n 2 -b 100 -i 1 -w 10.0.0.1:6000,10.0.0.2:6000

with the reference's flag names (``-n -s -sm`` / ``-ps -n -b -ba -sa -s`` / ``-n -b -i -w``). ``model_fn`` and
``dataset_fn`` come from ``--functions MODULE`` (default: the repository's ``initializer`` module, i.e. the example
MLP on MNIST); everything else (``--device``, ``--dtype``, ``--epochs``, ``--max-steps``, ``--port`` ...) is the
launcher's extension flag set.
"""
from __future__ import annotations

import argparse
import importlib
import os
import sys

from . import launcher


def _functions(module_name: str):
    root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    if root not in sys.path:
        sys.path.insert(0, root)
    mod = importlib.import_module(module_name)
    return mod.model_fn, mod.dataset_fn


def _launcher_args(extra: list[str]):
    """The launcher's namespace with its defaults, overridden by the extension flags given after the role flags."""
    return launcher.build_parser().parse_args(["-m", "c"] + extra)


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(prog="standalone", description=__doc__.split("\n\n")[0])
    p.add_argument("--functions", default="initializer", help="module that defines model_fn and dataset_fn")
    sub = p.add_subparsers(dest="role", required=True)
    s = sub.add_parser("server", help="parameter server (reference centralized/server.py)")
    s.add_argument("-n", dest="n_nodes", type=int, required=True, help="number of nodes on the network")
    s.add_argument("-s", dest="server_address", default="127.0.0.1", help="server address")
    s.add_argument("-sm", dest="sync_mode", default="sync", help="sync mode: sync or async")
    w = sub.add_parser("worker", help="parameter-server worker (reference centralized/client.py)")
    w.add_argument("-ps", dest="index", type=int, required=True, help="node position on the network")
    w.add_argument("-n", dest="n_nodes", type=int, required=True, help="number of nodes on the network")
    w.add_argument("-b", dest="batch_size", type=int, default=100, help="batch size")
    w.add_argument("-ba", dest="base_address", default="127.0.0.", help="base node address (unused, as in the reference)")
    w.add_argument("-sa", dest="supervisor_address", default="127.0.0.1", help="supervisor address")
    w.add_argument("-s", dest="server_address", default="127.0.0.1", help="server address")
    k = sub.add_parser("keras", help="one mirrored all-reduce worker (reference decentralized/native/dist_keras.py)")
    k.add_argument("-n", dest="n_nodes", type=int, required=True, help="number of nodes on the network")
    k.add_argument("-b", dest="batch_size", type=int, default=100, help="batch size")
    k.add_argument("-i", dest="index", type=int, default=0, help="index")
    k.add_argument("-w", dest="worker_addresses", required=True, help="comma separated host:port of every worker")
    return p


def main(argv: list[str] | None = None) -> int:
    role_args, extra = build_parser().parse_known_args(argv)
    model_fn, dataset_fn = _functions(role_args.functions)
    a = _launcher_args(extra)
    a.n_nodes = role_args.n_nodes
    if role_args.role == "server":
        if role_args.sync_mode not in ("sync", "async"):
            raise ValueError("Invalid centralized strategy")
        a.centralized_strategy, a.server_address = role_args.sync_mode, role_args.server_address
        launcher._tcp_server(a, model_fn, dataset_fn, None, None)
    elif role_args.role == "worker":
        a.batch_size, a.server_address = role_args.batch_size, role_args.server_address
        launcher._tcp_worker(role_args.index, a, model_fn, dataset_fn, None)
    else:
        workers = [x.strip() for x in role_args.worker_addresses.split(",")]
        if len(workers) != role_args.n_nodes:
            raise ValueError(f"-w lists {len(workers)} workers, -n says {role_args.n_nodes}")
        host, _, port = workers[0].partition(":")          # worker 0 hosts the rendezvous
        a.batch_size, a.master_addr = role_args.batch_size, host
        launcher._collective_rank(role_args.index, role_args.n_nodes, a, model_fn, dataset_fn, "keras",
                                  int(port) if port else a.port, None)
    return 0


if __name__ == "__main__":
    sys.exit(main())
