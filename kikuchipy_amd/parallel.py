"""Dictionary sharding over the GPUs of one node: one process per GPU.

The reference has no distributed layer; this is the
MI355X-native counterpart of its `n_per_iteration` chunking.  Rank r matches the
contiguous dictionary block `shard_range(N, r, R)` - so a global dictionary
index is local index + block start, like `simulation_indices_i += start`
(indexing/_dictionary_indexing.py:118) - against ALL experimental patterns
(replicated, <= 576 MB prepared at 40k patterns).  The only exchange step is
the merge of the per-rank best-k lists: one RCCL all-gather of M*k*(4+4) bytes
per rank over xGMI inside `kpdi_finalize`, followed by the same (score desc,
index asc) merge kernel used between chunks, so every rank ends with the
bit-identical global result.

Fallback chain of that exchange (`Communicator.attach`): RCCL all-gather -> when
the communicator cannot be created on some rank, or its first all-gather does
not complete in time (the ranks agree over the control plane), a HOST-STAGED
gather of the same lists over the control plane (`gather_lists`: 0.66 MB per rank
at configs[1], 6.4 MB at configs[3]) feeding the same merge kernel -> and
`bench.py` additionally retries as ONE process driving every GPU
(`--single-process`: a `kpdi_group`, in-process RCCL or peer copies).  Which
gather ran, and why, is kept in `Communicator.gather` / `.gather_reason`.

Refinement shards the other way: every rank refines a contiguous block of the
map's patterns (they are independent), and the per-pattern results (9 doubles)
are concatenated over the control plane (`Communicator.all_gather_rows`).

Control plane (rank discovery, the 128-byte RCCL unique id, barriers, the
max-over-ranks of a timing): `SocketGroup`, ~150 lines of plain TCP over the
loopback interface - no PyTorch anywhere in this package.  It reads the same
environment a launcher like `python -m torch.distributed.run` exports
(RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT), so the driver's command line
works unchanged; `Communicator(..., broadcast_bytes=, barrier=, all_gather=)`
accepts any other transport as three callables (tests/_gloo_worker.py passes
`torch.distributed` ones).
"""

import hashlib
import hmac
import io
import json
import os
import socket
import struct
import threading
import time

import numpy as np


def _prefer_dmabuf_ipc():
    """RCCL shares buffers between the PROCESSES of a node through dmabuf IPC; the legacy mode fails with
    `hipIpcGetMemHandle: invalid argument` on hosts whose driver only supports dmabuf.  HSA reads the variable when the
    runtime initialises (the first call into libkpdi), so it is set when a multi-rank `Communicator` is made - the only
    place that needs it - and not as a side effect of importing this package (`setdefault`: the user's choice stands)."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def _call_with_timeout(fn, timeout):
    """None when `fn()` returned within `timeout` seconds, else what went wrong as text.  A call that never returns
    (a collective bootstrap waiting for a rank that has already failed) is left behind on its daemon thread."""
    box = []

    def run():
        try:
            fn()
            box.append(None)
        except Exception as e:  # noqa: BLE001 - whatever it is, the ranks must hear about it
            box.append(f"{type(e).__name__}: {e}")

    t = threading.Thread(target=run, daemon=True)
    t.start()
    t.join(timeout)
    if not box:
        return f"no answer within {timeout:g} s"
    return box[0]


def shard_range(n_total, rank, world_size):
    """Contiguous block [start, end) of rank `rank`: sizes differ by at most 1."""
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError(f"bad rank {rank} / world size {world_size}")
    base, rem = divmod(int(n_total), int(world_size))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


# ---------------------------------------------------------------------------------------------
# wire format: no pickle (a socket is not a trusted source of code) - JSON for plain values,
# the .npy format (allow_pickle=False) for arrays, a tagged list for tuples / lists of those
# ---------------------------------------------------------------------------------------------
def _encode(obj):
    if obj is None:
        return b"N"
    if isinstance(obj, (bytes, bytearray, memoryview)):
        return b"B" + bytes(obj)
    if isinstance(obj, np.ndarray):
        buf = io.BytesIO()
        np.lib.format.write_array(buf, np.ascontiguousarray(obj), allow_pickle=False)
        return b"A" + buf.getvalue()
    if isinstance(obj, (tuple, list)):
        parts = [_encode(o) for o in obj]
        head = struct.pack("<cI", b"T" if isinstance(obj, tuple) else b"L", len(parts))
        return head + b"".join(struct.pack("<Q", len(p)) + p for p in parts)
    if isinstance(obj, np.generic):
        obj = obj.item()
    return b"J" + json.dumps(obj).encode()


def _decode(data):
    tag, body = data[:1], data[1:]
    if tag == b"N":
        return None
    if tag == b"B":
        return bytes(body)
    if tag == b"A":
        return np.lib.format.read_array(io.BytesIO(body), allow_pickle=False)
    if tag in (b"T", b"L"):
        (n,) = struct.unpack("<I", body[:4])
        out, at = [], 4
        for _ in range(n):
            (ln,) = struct.unpack("<Q", body[at:at + 8])
            out.append(_decode(body[at + 8:at + 8 + ln]))
            at += 8 + ln
        return tuple(out) if tag == b"T" else out
    if tag == b"J":
        return json.loads(body.decode())
    raise ValueError(f"unknown message tag {tag!r}")


def _send_msg(sock, payload):
    sock.sendall(struct.pack("<Q", len(payload)) + payload)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(n - len(buf), 1 << 20))
        if not chunk:
            raise ConnectionError("control-plane peer closed the connection (did another rank fail?)")
        buf += chunk
    return bytes(buf)


_MAX_MESSAGE = 1 << 30  # the control plane carries ids, timings and a few result rows per pattern


def _recv_msg(sock):
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    if n > _MAX_MESSAGE:
        raise ConnectionError(f"control-plane message of {n} bytes announced (limit {_MAX_MESSAGE}): not a peer of this job?")
    return _recv_exact(sock, n)


_MAGIC = b"KPDI-RDV1"
_PORT_CANDIDATES = 32


def _job_token(world_size, addr="", port=0):
    """What tells this job's rendezvous from a stale / foreign one on a neighbouring port: the
    launcher's run id when it exports one (torchrun: TORCHELASTIC_RUN_ID; bench.py's own spawner:
    KPDI_JOB_ID) - else the rendezvous address itself, so that two jobs of equal size whose port
    ranges overlap still cannot join each other - and the world size."""
    run = (os.environ.get("KPDI_JOB_KEY") or os.environ.get("KPDI_JOB_ID") or os.environ.get("TORCHELASTIC_RUN_ID")
           or f"rendezvous-{addr}:{int(port)}")
    return hashlib.sha256(f"{run}/{int(world_size)}".encode()).digest()[:16]


def _greeting(token):
    """What rank 0 says first: the magic string and a DIGEST of the job token (never the token itself) - enough for a
    rank of this job to recognise its server, useless for computing a proof."""
    return _MAGIC + hashlib.sha256(b"kpdi-greeting" + token).digest()[:16]


def _rank_proof(token, nonce, rank, world_size):
    """A joining rank's answer to the server's challenge: HMAC(job token, nonce | rank | world size).  The token is
    never on the wire, the nonce is fresh per connection: a process that only READS a greeting (or replays an old
    answer) cannot claim a rank.  What the token is worth depends on the launcher: with $KPDI_JOB_ID /
    $TORCHELASTIC_RUN_ID (or $KPDI_JOB_KEY) it is a secret of the job's processes; without them it is derived from
    the rendezvous address and the world size - then this is collision avoidance between jobs, not authentication,
    and the control plane should stay on the loopback interface (it binds MASTER_ADDR, 127.0.0.1 by default)."""
    return hmac.new(token, nonce + struct.pack("<ii", int(rank), int(world_size)), hashlib.sha256).digest()[:16]


_ACCEPT, _REJECT = b"\x01", b"\x00"


class SocketGroup:
    """The ranks of one job as a TCP star around rank 0.

    Rendezvous: rank 0 listens on MASTER_ADDR at the first free port of MASTER_PORT,
    MASTER_PORT + 1, ... (under `torch.distributed.run` MASTER_PORT itself is taken by the
    launcher's own store, so the first candidate is usually busy) and GREETS every connection with
    a magic string + a digest of the job token + a fresh nonce; the other ranks walk the same candidates, only talk to a
    server that greeted them correctly - they never write to a foreign server - and answer the nonce with an HMAC under
    the token (`_rank_proof`).  Every collective
    goes through rank 0 (world sizes are <= 8 and the payloads are a 128-byte id, a timing, or a few
    result rows: latency of tens of microseconds on loopback, irrelevant next to a sweep)."""

    def __init__(self, rank, world_size, master_addr=None, master_port=None, timeout=120.0):
        self._peers = {}  # rank 0: rank -> socket
        self._up = None   # other ranks: socket to rank 0
        self._server = None  # (set before anything can raise: __del__ closes what is there)
        self.rank, self.world_size = int(rank), int(world_size)
        if not 0 <= self.rank < self.world_size:
            raise ValueError(f"bad rank {rank} / world size {world_size}")
        self.timeout = float(timeout)
        if self.world_size == 1:
            return
        addr = master_addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = int(master_port or os.environ.get("MASTER_PORT", "29500"))
        token = _job_token(self.world_size, addr, port)
        if self.rank == 0:
            self._listen(addr, port, token)
        else:
            self._connect(addr, port, token)

    @classmethod
    def from_env(cls, timeout=120.0):
        """RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT as exported by `python -m
        torch.distributed.run` or by `bench.py --gpus N`'s own spawner."""
        return cls(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), timeout=timeout)

    # -- rendezvous
    def _listen(self, addr, port, token):
        srv, err = None, None
        for cand in range(port, port + _PORT_CANDIDATES):
            s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            try:
                s.bind((addr, cand))
                s.listen(self.world_size)
                srv = s
                break
            except OSError as e:  # busy (the launcher's store, a previous job): next candidate
                err = e
                s.close()
        if srv is None:
            raise ConnectionError(f"rank 0 found no free rendezvous port in {port}..{port + _PORT_CANDIDATES - 1}: {err}")
        self._server = srv
        self.port = srv.getsockname()[1]
        deadline = time.monotonic() + self.timeout
        while len(self._peers) < self.world_size - 1:
            srv.settimeout(max(0.05, deadline - time.monotonic()))
            try:
                conn, _ = srv.accept()
            except socket.timeout:
                missing = sorted(set(range(1, self.world_size)) - set(self._peers))
                raise TimeoutError(f"rendezvous: ranks {missing} did not join within {self.timeout:.0f} s") from None
            conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            conn.settimeout(5.0)
            try:
                nonce = os.urandom(16)
                conn.sendall(_greeting(token) + nonce)
                hello = _recv_exact(conn, 8 + 16)
                peer, world = struct.unpack("<ii", hello[:8])
                ok = (world == self.world_size and 0 < peer < world and peer not in self._peers
                      and hmac.compare_digest(hello[8:], _rank_proof(token, nonce, peer, world)))
                conn.sendall(_ACCEPT if ok else _REJECT)  # the peer learns NOW, not at its first collective
            except (OSError, ConnectionError):
                conn.close()  # a port scanner, a rank of another job that read the greeting and left
                continue
            if not ok:
                conn.close()
                continue
            conn.settimeout(self.timeout)
            self._peers[peer] = conn

    def _connect(self, addr, port, token):
        deadline = time.monotonic() + self.timeout
        want = _greeting(token)
        while True:
            for cand in range(port, port + _PORT_CANDIDATES):
                s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                s.settimeout(1.0)
                try:
                    s.connect((addr, cand))
                    # only OUR rank 0 speaks first; a foreign server stays silent -> timeout -> next
                    s.settimeout(0.5)
                    if _recv_exact(s, len(want)) == want:
                        nonce = _recv_exact(s, 16)
                        s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                        s.settimeout(self.timeout)
                        s.sendall(struct.pack("<ii", self.rank, self.world_size) + _rank_proof(token, nonce, self.rank, self.world_size))
                        if _recv_exact(s, 1) != _ACCEPT:
                            s.close()
                            raise PermissionError(f"rank {self.rank}: the rendezvous server on {addr}:{cand} rejected this rank "
                                                  "(rank already taken, or another world size)")
                        self._up, self.port = s, cand
                        return
                except PermissionError:
                    raise
                except (OSError, ConnectionError):
                    pass
                s.close()
            if time.monotonic() > deadline:
                raise TimeoutError(f"rank {self.rank}: no rendezvous server of this job on {addr}:{port}.."
                                   f"{port + _PORT_CANDIDATES - 1} within {self.timeout:.0f} s")
            time.sleep(0.05)

    # -- collectives (every rank must call them in the same order)
    def all_gather(self, obj):
        """[rank 0's obj, rank 1's obj, ...] on every rank."""
        if self.world_size == 1:
            return [obj]
        mine = _encode(obj)
        if self.rank == 0:
            parts = [mine] + [_recv_msg(self._peers[r]) for r in range(1, self.world_size)]
            blob = _encode(parts)
            for r in range(1, self.world_size):
                _send_msg(self._peers[r], blob)
        else:
            _send_msg(self._up, mine)
            parts = _decode(_recv_msg(self._up))
        return [_decode(p) for p in parts]

    def broadcast_bytes(self, payload, src=0):
        """`payload` of rank `src` on every rank."""
        if self.world_size == 1:
            return payload
        return self.all_gather(payload if self.rank == src else None)[src]

    def barrier(self):
        self.all_gather(None)

    def all_reduce_max(self, value):
        return max(float(v) for v in self.all_gather(float(value)))

    def close(self):
        for s in list(getattr(self, "_peers", {}).values()) + [getattr(self, "_up", None), getattr(self, "_server", None)]:
            if s is not None:
                try:
                    s.close()
                except OSError:
                    pass
        self._peers, self._up, self._server = {}, None, None

    def __del__(self):
        self.close()


class Communicator:
    """Rank/world bookkeeping + creation of the RCCL communicator inside a libkpdi context.
    Transport of the control plane: a `SocketGroup` (default: from the environment), or three
    callables - `broadcast_bytes(payload_or_None, src) -> bytes`, `barrier()`,
    `all_gather(obj) -> list` - of whatever the caller already runs (MPI, torch.distributed)."""

    def __init__(self, rank, world_size, broadcast_bytes=None, barrier=None, all_gather=None, group=None):
        self.rank = int(rank)
        self.world_size = int(world_size)
        self.group = group
        if self.world_size > 1:
            _prefer_dmabuf_ipc()
        if self.world_size > 1 and group is None and not (broadcast_bytes and barrier and all_gather):
            self.group = group = SocketGroup(self.rank, self.world_size)
        self._broadcast = broadcast_bytes or (group.broadcast_bytes if group else None)
        self._barrier = barrier or (group.barrier if group else None)
        self._all_gather = all_gather or (group.all_gather if group else None)

    @classmethod
    def from_env(cls):
        """RANK / WORLD_SIZE (+ MASTER_ADDR / MASTER_PORT for the rendezvous) as exported by
        `python -m torch.distributed.run` or `bench.py --gpus N`."""
        return cls(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))

    def barrier(self):
        if self.world_size > 1:
            self._barrier()

    def all_gather(self, obj):
        """One object per rank (rank order) on every rank, over the control plane."""
        if self.world_size == 1:
            return [obj]
        return list(self._all_gather(obj))

    def all_reduce_max(self, value):
        return max(float(v) for v in self.all_gather(float(value)))

    def exchange_unique_id(self, make_id):
        """Rank 0 creates the id with `make_id()`; every rank returns it."""
        payload = make_id() if self.rank == 0 else None
        if self.world_size == 1:
            return payload
        return self._broadcast(payload, 0)

    def all_gather_rows(self, array):
        """Concatenate the ranks' row blocks (rank order) on every rank - for small
        per-pattern results (refinement: a few doubles per pattern) over the control plane."""
        if self.world_size == 1:
            return array
        return np.concatenate([np.asarray(b) for b in self._all_gather(array)], axis=0)

    gather = None         # "rccl" | "host": how the ranks' lists are gathered (set by the first attach)
    gather_reason = ""    # why the host-staged gather was taken

    def attach(self, ctx):
        """Make `ctx` part of the job's gather, once (collective call: every rank must attach its context in the
        same order).  RCCL first: rank 0's unique id travels over the control plane, every rank creates its
        communicator (`kpdi_comm_init`) and runs one all-gather through it (`kpdi_comm_selftest`), each under a timeout
        ($KPDI_COMM_TIMEOUT seconds, default 60), and the ranks tell each other how that went.  Unless it went well
        everywhere, every rank drops its communicator and the lists are gathered over the control plane instead
        (`gather_lists`, called by the context's `finalize`); $KPDI_GATHER=host goes there directly.
        The attachment is recorded ON the context - not by `id(ctx)`, which CPython hands to the next context once
        this one is collected: a new context then looked attached, skipped `kpdi_comm_init` and silently merged nothing."""
        if self.world_size == 1 or getattr(ctx, "_comm", None) is self:
            return
        mode, why = self._negotiate(ctx)
        ctx._comm = self
        ctx._host_gather = self if mode == "host" else None
        self.gather, self.gather_reason = mode, why

    def _negotiate(self, ctx):
        timeout = float(os.environ.get("KPDI_COMM_TIMEOUT", "60"))
        payload = None
        if self.rank == 0:
            if os.environ.get("KPDI_GATHER", "").lower() == "host":
                payload = b"E" + b"KPDI_GATHER=host"
            else:
                try:
                    payload = b"U" + ctx.comm_unique_id()
                except Exception as e:  # noqa: BLE001 - librccl missing, no device ...
                    payload = b"E" + f"rank 0: {e}".encode()
        msg = self._broadcast(payload, 0)
        if msg[:1] != b"U":
            return "host", msg[1:].decode()
        uid = msg[1:]
        status = _call_with_timeout(lambda: ctx.comm_init(self.rank, self.world_size, uid), timeout)
        statuses = self.all_gather(status)
        stage = "kpdi_comm_init"
        if all(st is None for st in statuses):
            n_bytes = int(os.environ.get("KPDI_COMM_SELFTEST_BYTES", str(1 << 20)))
            status = _call_with_timeout(lambda: ctx.comm_selftest(n_bytes, int(timeout * 1000)), timeout + 5)
            statuses = self.all_gather(status)
            stage = "first all-gather"
        bad = [r for r, st in enumerate(statuses) if st is not None]
        if not bad:
            return "rccl", ""
        try:
            ctx.comm_drop()  # (every rank: a communicator that only some ranks hold would hang the first finalize)
        except Exception:  # noqa: BLE001
            pass
        # (a rank that failed outright says more than the ranks that then waited for it in vain)
        told = [r for r in bad if not statuses[r].startswith("no answer")] or bad
        return "host", f"{stage} failed on rank {told[0]} ({statuses[told[0]]})"

    def gather_lists(self, ctx):
        """The host-staged gather: every rank's own best-k lists (`kpdi_export_lists`) all-gathered over the control
        plane and handed to `ctx` (`kpdi_import_lists`), whose finalize then merges them like all-gathered ones."""
        scores, indices = ctx.export_lists()
        parts = self.all_gather((scores, indices))
        ctx.import_lists(np.stack([np.asarray(p[0]) for p in parts]), np.stack([np.asarray(p[1]) for p in parts]))

    def close(self):
        if self.group is not None:
            self.group.close()
