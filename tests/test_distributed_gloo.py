"""world_size-2 tests of the multi-rank host path on CPU: torch.distributed (gloo) as the optional
transport, and the product's own torch-free TCP control plane (kikuchipy_amd.parallel.SocketGroup)."""
import os
import socket
import subprocess
import sys

from conftest import ROOT


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_gloo():
    env = dict(os.environ)
    env["OMP_NUM_THREADS"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "_gloo_worker.py")]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    assert "GLOO_WORKER_OK" in p.stdout


def launch_plain(script, n_ranks, extra_env=None, args=()):
    """One plain process per rank with the environment a launcher would export (no torch anywhere)."""
    port = free_port()
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), KPDI_JOB_ID=f"test-{os.getpid()}-{port}", OMP_NUM_THREADS="1")
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, script, *args], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so[-3000:] + se[-3000:]
    return outs[0][0]


def test_two_ranks_plain_sockets():
    out = launch_plain(os.path.join(ROOT, "tests", "_gloo_worker.py"), 2, {"KPDI_TEST_TRANSPORT": "socket"})
    assert "SOCKET_WORKER_OK" in out


def test_socket_group_wire_format_and_port_walk():
    """No pickle on the wire; a busy MASTER_PORT (the launcher's own store) is stepped over; a
    foreign, silent server on a candidate port is never written to."""
    import socket
    import threading

    import numpy as np

    from kikuchipy_amd import parallel

    for obj in (None, b"\x00\x01", 3, 2.5, "id", [1, "a", None], (np.arange(6, dtype=np.float32).reshape(2, 3), 7),
                {"a": [1, 2]}, np.float64(1.5)):
        back = parallel._decode(parallel._encode(obj))
        if isinstance(obj, tuple):
            assert isinstance(back, tuple) and np.array_equal(back[0], obj[0]) and back[1] == obj[1]
        else:
            assert back == obj
    try:
        parallel._encode(object())
    except TypeError:
        pass
    else:
        raise AssertionError("arbitrary objects must not be serialisable")

    busy = socket.socket()
    busy.bind(("127.0.0.1", 0))
    busy.listen(8)  # silent: accepts (kernel backlog) but never greets - like torchrun's TCPStore
    port = busy.getsockname()[1]
    received = []
    os.environ["KPDI_JOB_ID"] = "wire-test"
    res = [None, None]

    def rank(r):
        g = parallel.SocketGroup(r, 2, "127.0.0.1", port, timeout=30)
        res[r] = (g.port, g.all_gather(np.full(3, r)), g.broadcast_bytes(b"uid" if r == 0 else None, 0),
                  g.all_reduce_max(10 + r))
        g.barrier()
        g.close()

    threads = [threading.Thread(target=rank, args=(r,)) for r in range(2)]
    [t.start() for t in threads]
    [t.join(60) for t in threads]
    busy.settimeout(0.2)
    try:
        while True:
            conn, _ = busy.accept()
            conn.settimeout(0.2)
            try:
                received.append(conn.recv(64))
            except OSError:
                received.append(b"")
            conn.close()
    except OSError:
        pass
    busy.close()
    del os.environ["KPDI_JOB_ID"]
    assert res[0] is not None and res[1] is not None
    assert res[0][0] == res[1][0] != port  # both ended on the same port next to the busy one
    for r in range(2):
        assert [list(a) for a in res[r][1]] == [[0, 0, 0], [1, 1, 1]] and res[r][2] == b"uid" and res[r][3] == 11.0
    assert all(b == b"" for b in received), received  # nothing was ever sent to the foreign server


def test_socket_group_rejects_what_is_not_a_rank_of_this_job(monkeypatch):
    """A connector must PROVE it knows the job token: the greeting carries only a digest of it and a fresh nonce, the
    answer is an HMAC under the token (a process that reads the greeting, or replays an answer given to another nonce,
    cannot claim a rank); a rank that is already taken is told at once, an absurd message length is refused, and a constructor that raises leaves an
    object whose __del__ is harmless."""
    import socket
    import struct
    import threading

    import pytest

    from kikuchipy_amd import parallel

    monkeypatch.setenv("KPDI_JOB_ID", "handshake-test")
    with pytest.raises(ValueError):
        parallel.SocketGroup(5, 2)  # (and its __del__ must not raise: attributes are set first)
    port = free_port()
    box = {}

    def server():
        box["g"] = parallel.SocketGroup(0, 3, "127.0.0.1", port, timeout=30)

    t = threading.Thread(target=server)
    t.start()
    token = parallel._job_token(3, "127.0.0.1", port)
    # an impostor that read the greeting and claims rank 1 without the proof
    for _ in range(100):
        try:
            s = socket.create_connection(("127.0.0.1", port), timeout=5)
            break
        except OSError:
            import time
            time.sleep(0.05)
    greeting = parallel._recv_exact(s, len(parallel._MAGIC) + 16)
    assert greeting == parallel._greeting(token) and token not in greeting  # the token itself is never on the wire
    nonce1 = parallel._recv_exact(s, 16)
    s.sendall(struct.pack("<ii", 1, 3) + b"\x00" * 16)
    assert parallel._recv_exact(s, 1) == parallel._REJECT
    s.close()
    # ... and one that replays a valid answer to ANOTHER connection's nonce
    s = socket.create_connection(("127.0.0.1", port), timeout=5)
    parallel._recv_exact(s, len(greeting))
    nonce2 = parallel._recv_exact(s, 16)
    assert nonce2 != nonce1
    s.sendall(struct.pack("<ii", 1, 3) + parallel._rank_proof(token, nonce1, 1, 3))
    assert parallel._recv_exact(s, 1) == parallel._REJECT
    s.close()
    g1 = parallel.SocketGroup(1, 3, "127.0.0.1", port, timeout=30)   # the real rank 1 still gets in
    with pytest.raises(PermissionError, match="rejected"):
        parallel.SocketGroup(1, 3, "127.0.0.1", port, timeout=5)      # a second rank 1 learns at once
    g2 = parallel.SocketGroup(2, 3, "127.0.0.1", port, timeout=30)
    t.join(30)
    g0 = box["g"]
    # a peer announcing a 2^40-byte message is not believed
    g1._up.sendall(struct.pack("<Q", 1 << 40))
    with pytest.raises(ConnectionError, match="limit"):
        parallel._recv_msg(g0._peers[1])
    for g in (g0, g1, g2):
        g.close()
    # no job id: the rendezvous address is part of the token (two jobs on neighbouring ports do not mix)
    monkeypatch.delenv("KPDI_JOB_ID")
    monkeypatch.delenv("TORCHELASTIC_RUN_ID", raising=False)
    assert parallel._job_token(2, "127.0.0.1", 29500) != parallel._job_token(2, "127.0.0.1", 29501)


def test_import_has_no_side_effect_on_the_environment():
    import subprocess
    import sys

    code = ("import os; os.environ.pop('HSA_ENABLE_IPC_MODE_LEGACY', None); import kikuchipy_amd, kikuchipy_amd.parallel as p; "
            "assert 'HSA_ENABLE_IPC_MODE_LEGACY' not in os.environ; p.Communicator(0, 1); "
            "assert 'HSA_ENABLE_IPC_MODE_LEGACY' not in os.environ; "
            "p.Communicator(0, 2, broadcast_bytes=lambda *a: b'', barrier=lambda: None, all_gather=lambda o: [o, o]); "
            "assert os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'")
    subprocess.run([sys.executable, "-c", code], check=True, cwd=ROOT)
