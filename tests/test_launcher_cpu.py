"""Launcher contract: flags, dispatch and end-to-end multi-process runs on CPU (gloo / TCP)."""
import argparse
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from distributed_tensorflow_amd.cli import launcher  # noqa: E402


def test_flag_names_and_defaults_match_reference_contract():
    p = launcher.build_parser()
    a = p.parse_args([])
    assert a.mode is None and a.centralized_strategy is None and a.decentralized_strategy is None
    assert a.n_nodes is None and a.batch_size == 100
    assert a.task_type is None and a.task_index is None
    assert a.server_address == "127.0.0.1" and a.cpu_affinity is False
    a = p.parse_args("-m c -cs sync -ds keras -n 3 -b 64 -tt worker -ti 2 -sa 10.0.0.1 -ca y -d 4".split())
    assert (a.mode, a.centralized_strategy, a.decentralized_strategy) == ("c", "sync", "keras")
    assert (a.n_nodes, a.batch_size, a.task_type, a.task_index) == (3, 64, "worker", 2)
    assert a.server_address == "10.0.0.1" and a.cpu_affinity is True and a.degree == 4


@pytest.mark.parametrize("v,exp", [("yes", True), ("true", True), ("t", True), ("y", True), ("1", True),
                                   ("no", False), ("false", False), ("f", False), ("n", False), ("0", False),
                                   (True, True)])
def test_str2bool(v, exp):
    assert launcher.str2bool(v) is exp


def test_str2bool_rejects_garbage():
    with pytest.raises(argparse.ArgumentTypeError):
        launcher.str2bool("maybe")


def _probe_port():
    """Separate commands have to be TOLD a port: probe a free one (single-command jobs use --port 0 instead)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _args(s):
    return launcher.build_parser().parse_args(s.split())


def test_invalid_mode_and_strategy_raise_value_error():
    with pytest.raises(ValueError, match="Invalid mode"):
        launcher.run(_args("-m x -n 1"), None, None)
    with pytest.raises(ValueError, match="Invalid centralized strategy"):
        launcher.run(_args("-m c -cs nope -n 1"), None, None)
    with pytest.raises(ValueError, match="Invalid decentralized strategy"):
        launcher.run(_args("-m d -ds nope -n 1"), None, None)


ENV = dict(os.environ, DTF_SYNTH_MNIST_TRAIN="2000", DTF_SYNTH_MNIST_TEST="400", OMP_NUM_THREADS="2",
           PYTHONPATH=ROOT)


def _launch(flags, tmp_path, timeout=240):
    res = tmp_path / "res.json"
    cmd = [sys.executable, os.path.join(ROOT, "initializer.py"), *flags.split(), "--device", "cpu", "--results", str(res)]
    p = subprocess.run(cmd, env=ENV, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    return p.stdout, json.loads(res.read_text())


@pytest.mark.slow
@pytest.mark.parametrize("flags,port", [
    ("-m c -cs sync -n 2 -b 50 --transport tcp -ca y", 0),
    ("-m c -cs async -n 2 -b 50 --transport tcp", 0),
    ("-m c -cs sync -n 2 -b 50 --transport tcp --apply sequential", 0),
])
def test_centralized_tcp_end_to_end(flags, port, tmp_path):
    out, res = _launch(f"{flags} --port {port}", tmp_path)
    assert "elapsed time = " in out and "Test set accuracy: " in out   # the reference's two result lines
    assert res["ok"] and res["exit_codes"] == [0, 0, 0]               # 1 server + 2 workers
    assert res["server"]["test_accuracy"] > 0.3
    assert res["worker0"]["steps"] == 20 and res["worker1"]["steps"] == 20


@pytest.mark.slow
@pytest.mark.parametrize("flags", [
    "-m c -cs sync -n 2 -b 50 --transport collective",
    "-m c -cs sync -n 2 -b 50 --transport collective --apply sequential",
    "-m c -cs async -n 2 -b 50 --transport collective",
    "-m c -cs async -n 2 -b 50 --transport collective --placement dedicated",
    "-m d -ds keras -n 2 -b 50",
    "-m d -ds custom -n 2 -b 50 --algorithm direct",
    "-m d -ds graph -n 3 -b 50 -d 2",
])
def test_collective_strategies_end_to_end(flags, tmp_path):
    out, res = _launch(flags, tmp_path)
    assert res["ok"] and all(c == 0 for c in res["exit_codes"])
    if "-m d" in flags:
        assert "elapsed time: " in out and "eval: " in out
        accs = [v["test_accuracy"] for k, v in res.items() if k.startswith("rank")]
        assert len(set(round(a, 6) for a in accs)) == 1   # replicas agree on the final model
        assert accs[0] > 0.5
    else:
        assert "Test set accuracy: " in out


@pytest.mark.slow
def test_worker_failure_ends_sync_job_with_error(tmp_path):
    """A worker killed mid-run must end the job with a non-zero exit, not hang it (the reference
    hangs in barrier.wait(), reference centralized/server.py:95)."""
    cmd = [sys.executable, os.path.join(ROOT, "initializer.py"), *"-m c -cs sync -n 2 -b 50 --transport tcp --port 0".split(),
           "--device", "cpu", "--fault-inject", "1:3"]
    p = subprocess.run(cmd, env=ENV, capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert p.returncode != 0


@pytest.mark.slow
def test_multi_machine_roles(tmp_path):
    """-tt server / -tt worker -ti k -sa ADDR (reference initializer.py:147-155), run as separate commands."""
    base = [sys.executable, os.path.join(ROOT, "initializer.py"), "-m", "c", "-cs", "sync", "-n", "2", "-b", "50",
            "-sa", "127.0.0.1", "--port", str(_probe_port()), "--device", "cpu"]
    srv = subprocess.Popen(base + ["-tt", "server"], env=ENV, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd=ROOT)
    ws = [subprocess.Popen(base + ["-tt", "worker", "-ti", str(i)], env=ENV, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, cwd=ROOT) for i in range(2)]
    outs = [w.communicate(timeout=200)[0] for w in ws]
    sout = srv.communicate(timeout=200)[0]
    assert srv.returncode == 0, sout[-2000:]
    assert all(w.returncode == 0 for w in ws), outs
    assert "Test set accuracy: " in sout


def test_standalone_role_entry_points_run_a_job_started_by_hand(tmp_path):
    """The reference's three per-module ``__main__`` commands (broken there), one process each."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = [sys.executable, "-m", "distributed_tensorflow_amd.cli.standalone"]
    tail = ["--device", "cpu", "--max-steps", "12", "--port", str(port)]
    srv = subprocess.Popen(base + ["server", "-n", "2", "-s", "127.0.0.1", "-sm", "sync"] + tail, cwd=ROOT, env=ENV,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    ws = [subprocess.Popen(base + ["worker", "-ps", str(i), "-n", "2", "-b", "32", "-s", "127.0.0.1"] + tail, cwd=ROOT,
                           env=ENV, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for i in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in [srv] + ws]
    assert all(p.returncode == 0 for p in [srv] + ws), outs
    assert "Test set accuracy" in outs[0] and "elapsed time" in outs[0]
    # two mirrored workers, each its own command
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    addrs = f"127.0.0.1:{port},127.0.0.1:{port}"
    ks = [subprocess.Popen(base + ["keras", "-n", "2", "-b", "32", "-i", str(i), "-w", addrs, "--device", "cpu",
                                   "--max-steps", "12"], cwd=ROOT, env=ENV, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True) for i in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in ks]
    assert all(p.returncode == 0 for p in ks), outs
    assert all("eval:" in o for o in outs)


def _run_launcher(flags, timeout=300):
    cmd = [sys.executable, os.path.join(ROOT, "initializer.py"), *flags, "--device", "cpu"]
    return subprocess.run(cmd, env=ENV, capture_output=True, text=True, timeout=timeout, cwd=ROOT)


@pytest.mark.parametrize("mode", ["-m d -ds keras", "-m c -cs sync --transport collective"])
def test_killed_run_resumes_to_the_same_weights_bit_for_bit(tmp_path, mode):
    """10 steps uninterrupted == 10 steps with a worker killed at step 7 and a resume from the step-5 checkpoint
    (weights, optimizer moments, step counter, dropout RNG and data position all continue)."""
    import numpy as np
    base = mode.split() + "-n 2 -b 32 --max-steps 10 --checkpoint-every 5".split()
    a = _run_launcher(base + ["--checkpoint-dir", str(tmp_path / "ckA"), "--save-weights", str(tmp_path / "A.npz")])
    assert a.returncode == 0, a.stdout[-1500:] + a.stderr[-1500:]
    b1 = _run_launcher(base + ["--checkpoint-dir", str(tmp_path / "ckB"), "--fault-inject", "0:7",
                               "--comm-timeout-s", "30"])
    assert b1.returncode != 0                                  # the job died with its worker
    assert (tmp_path / "ckB" / "model.pt").exists()
    b2 = _run_launcher(base + ["--checkpoint-dir", str(tmp_path / "ckB"), "--resume", "--save-weights",
                               str(tmp_path / "B.npz"), "--results", str(tmp_path / "resB.json")])
    assert b2.returncode == 0, b2.stdout[-1500:] + b2.stderr[-1500:]
    res = json.load(open(tmp_path / "resB.json"))
    assert res["rank0"]["resumed_from"] == 5 and res["rank0"]["steps"] == 10
    wa, wb = np.load(tmp_path / "A.npz"), np.load(tmp_path / "B.npz")
    keys = [k for k in wa.files if k.startswith("weight_")]
    assert keys and set(keys) <= set(wb.files)
    for k in keys:
        assert np.array_equal(wa[k], wb[k]), k


def test_load_weights_flag_starts_training_from_the_file(tmp_path):
    import numpy as np
    a = _run_launcher("-m d -ds keras -n 1 -b 32 --max-steps 3".split() + ["--save-weights", str(tmp_path / "w.npz")])
    assert a.returncode == 0, a.stderr[-1500:]
    # zero steps from the loaded file (bf16 arenas: the loaded values must survive in the fp32 master) == the file
    b = _run_launcher("-m d -ds keras -n 1 -b 32 --max-steps 0 --dtype bf16".split()
                      + ["--load-weights", str(tmp_path / "w.npz"), "--save-weights", str(tmp_path / "w2.npz")])
    assert b.returncode == 0, b.stderr[-1500:]
    w, w2 = np.load(tmp_path / "w.npz"), np.load(tmp_path / "w2.npz")
    for k in (k for k in w.files if k.startswith("weight_")):
        np.testing.assert_allclose(w2[k], w[k], rtol=1e-2, atol=1e-3)     # bf16 parameters, exported from the shadow


def test_mirrored_workers_report_to_the_supervisor(tmp_path):
    """'start', 'done' and ('results', accuracy) frames of the keras path (reference dist_keras.py:34-58) arrive at
    a supervisor listening on port 4000's stand-in."""
    import socket
    import struct
    import threading
    srv = socket.socket()
    srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    try:
        srv.bind(("127.0.0.1", 4000))
    except OSError:
        pytest.skip("port 4000 is taken on this host")
    srv.listen(8)
    srv.settimeout(120)
    frames = []

    def reader(c):
        buf = b""
        while True:
            d = c.recv(65536)
            if not d:
                break
            buf += d
        while buf:
            n = struct.unpack(">I", buf[:4])[0]
            frames.append(json.loads(buf[4:4 + n]))
            buf = buf[4 + n:]

    def acceptor():
        ts = []
        try:
            for _ in range(2):
                c, _ = srv.accept()
                t = threading.Thread(target=reader, args=(c,))
                t.start()
                ts.append(t)
        except OSError:
            pass
        for t in ts:
            t.join(60)

    th = threading.Thread(target=acceptor)
    th.start()
    try:
        p = _run_launcher("-m d -ds keras -n 2 -b 32 --max-steps 4 --supervisor-address 127.0.0.1".split())
        assert p.returncode == 0, p.stderr[-1500:]
        th.join(60)
    finally:
        srv.close()
    assert frames.count("start") == 2 and frames.count("done") == 2
    results = [f for f in frames if isinstance(f, list) and f[0] == "results"]
    assert len(results) == 2 and all(0.0 <= r[1] <= 1.0 for r in results)


def test_watchdog_aborts_a_job_whose_peer_stalls(tmp_path):
    """A rank that stops (without dying) leaves its peer inside a collective: the watchdog ends that peer with exit
    code 19 and the launcher reports failure in bounded time."""
    import time
    t0 = time.time()
    p = _run_launcher("-m d -ds keras -n 2 -b 32 --max-steps 20 --fault-inject 1:3:hang --watchdog-s 5".split()
                      + ["--results", str(tmp_path / "r.json")], timeout=200)
    assert p.returncode != 0
    assert time.time() - t0 < 150
    assert "[watchdog] rank 0" in p.stderr
    res = json.load(open(tmp_path / "r.json"))
    assert 19 in res["exit_codes"]


def test_tcp_parameter_server_honours_checkpoint_resume_and_weight_flags(tmp_path):
    """The framed-TCP topology (the one --device cpu selects): 10 rounds uninterrupted == 10 rounds with a worker
    killed at step 7 and a resume from the server's step-5 checkpoint; --save-weights writes the server's model and
    --load-weights starts the server from a file."""
    import numpy as np
    base = "-m c -cs sync --transport tcp --port 0 -n 2 -b 32 --max-steps 10 --checkpoint-every 5".split()
    a = _run_launcher(base + ["--checkpoint-dir", str(tmp_path / "ckA"), "--save-weights", str(tmp_path / "A.npz")])
    assert a.returncode == 0, a.stdout[-1500:] + a.stderr[-1500:]
    assert (tmp_path / "ckA" / "ps_server.pt").exists() and (tmp_path / "A.npz").exists()
    b1 = _run_launcher(base + ["--checkpoint-dir", str(tmp_path / "ckB"), "--fault-inject", "0:7", "--comm-timeout-s", "30"])
    assert b1.returncode != 0
    assert (tmp_path / "ckB" / "ps_server.pt").exists()
    b2 = _run_launcher(base + ["--checkpoint-dir", str(tmp_path / "ckB"), "--resume", "--save-weights",
                               str(tmp_path / "B.npz"), "--results", str(tmp_path / "resB.json")])
    assert b2.returncode == 0, b2.stdout[-1500:] + b2.stderr[-1500:]
    res = json.load(open(tmp_path / "resB.json"))
    assert res["server"]["resumed"] is True
    assert res["worker0"]["resumed_from"] == 5 and res["worker0"]["steps"] == 10
    wa, wb = np.load(tmp_path / "A.npz"), np.load(tmp_path / "B.npz")
    keys = [k for k in wa.files if k.startswith("weight_")]
    assert keys
    for k in keys:
        assert np.array_equal(wa[k], wb[k]), k
    # zero steps from the loaded file == the file
    c = _run_launcher("-m c -cs sync --transport tcp --port 0 -n 1 -b 32 --max-steps 0".split()
                      + ["--load-weights", str(tmp_path / "A.npz"), "--save-weights", str(tmp_path / "C.npz")])
    assert c.returncode == 0, c.stderr[-1500:]
    wc = np.load(tmp_path / "C.npz")
    for k in keys:
        assert np.array_equal(wa[k], wc[k]), k


def test_load_weights_survives_the_first_asynchronous_apply(tmp_path):
    """-cs async keeps the authoritative shard apart from the replica's arena: imported weights must reach it, or the
    first apply continues from the seed-initialised weights. With a zero learning rate the exported model must equal
    the imported one after training steps."""
    import numpy as np
    a = _run_launcher("-m d -ds keras -n 1 -b 32 --max-steps 3 --seed 5".split() + ["--save-weights", str(tmp_path / "w.npz")])
    assert a.returncode == 0, a.stderr[-1500:]
    env = dict(ENV)
    cmd = [sys.executable, os.path.join(ROOT, "initializer.py"), *"-m c -cs async --transport collective -n 2 -b 32 --max-steps 2".split(),
           "--device", "cpu", "--learning-rate", "0", "--load-weights", str(tmp_path / "w.npz"), "--save-weights", str(tmp_path / "w2.npz")]
    b = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert b.returncode == 0, b.stdout[-1500:] + b.stderr[-1500:]
    w, w2 = np.load(tmp_path / "w.npz"), np.load(tmp_path / "w2.npz")
    for k in (k for k in w.files if k.startswith("weight_")):
        assert np.array_equal(w[k], w2[k]), k


def test_vram_pretouch_without_a_gpu_reports_failure_and_does_not_raise():
    """The throwaway first-touch process is best effort: on a machine without a GPU it reports nothing touched."""
    from distributed_tensorflow_amd.utils.vram import pretouch
    r = pretouch(0, timeout_s=120.0)
    assert r["ok"] is False and r["gib"] == 0.0 and r["seconds"] >= 0.0
