"""Rendezvous of the ranks of ONE node without PyTorch: a star of TCP connections on the loopback
interface, used for the 128-byte RCCL unique id, barriers and the max-over-ranks of timings.

The data path never goes through here: candidate pairs, video-level keys and hash shards are
exchanged by RCCL over xGMI inside the C-ABI (`hvd_comm_*`). This is the control channel only, so
plain sockets are enough -- and they keep `torch` out of the product (north_star: "no PyTorch").

Rank 0 listens on an ephemeral loopback port and publishes "port token" in a file that all ranks can
derive: ``$HVD_RDZV_FILE`` or ``<per-user dir>/hvd_rdzv_<MASTER_PORT>_<parent pid>`` -- the workers that
`python -m torch.distributed.run` (or any other launcher) starts share their parent process, and
MASTER_PORT distinguishes concurrent launches. Every collective is one round trip through rank 0.

Local hardening: the per-user directory is ``$XDG_RUNTIME_DIR`` or ``<tmp>/hvd_rdzv_<uid>`` (0700, must be
owned by this user and not a symlink); the file is created O_CREAT|O_EXCL|O_NOFOLLOW with mode 0600 and carries a
random 128-bit token that a peer must echo in its hello -- another local user can neither plant the file nor
claim a rank; the handshake has its own short timeout, so a stale file whose port was reused costs seconds,
not the collective timeout.
"""

from __future__ import annotations

import os
import secrets
import socket
import stat
import struct
import tempfile
import threading
import time

_MAGIC = b"HVDRDZV2"
_TOKEN_BYTES = 16
_HANDSHAKE_TIMEOUT = 5.0


def _user_dir() -> str:
    """A directory only this user can write: $XDG_RUNTIME_DIR, else <tmp>/hvd_rdzv_<uid> (created 0700, verified)."""
    xdg = os.environ.get("XDG_RUNTIME_DIR")
    if xdg and os.path.isdir(xdg) and os.access(xdg, os.W_OK):
        return xdg
    d = os.path.join(tempfile.gettempdir(), f"hvd_rdzv_{os.getuid()}")
    try:
        os.mkdir(d, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(d)
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise PermissionError(f"rendezvous directory {d} is not a private directory of this user")
    return d


def _default_file() -> str:
    explicit = os.environ.get("HVD_RDZV_FILE")
    if explicit:
        return explicit
    return os.path.join(_user_dir(), f"hvd_rdzv_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}")


def _recv_exact(sock: socket.socket, n: int) -> bytes:
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("rendezvous peer closed the connection")
        buf += chunk
    return bytes(buf)


def _send_msg(sock: socket.socket, payload: bytes) -> None:
    sock.sendall(struct.pack("<Q", len(payload)) + payload)


def _recv_msg(sock: socket.socket) -> bytes:
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    return _recv_exact(sock, n)


class Rendezvous:
    """rank/world default to $RANK / $WORLD_SIZE. world == 1 needs no sockets at all."""

    def __init__(self, rank: int | None = None, world: int | None = None, path: str | None = None, timeout: float = 300.0):
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        self.timeout = timeout
        self.path = path or _default_file()
        self._peers: list[socket.socket | None] = []
        self._up: socket.socket | None = None
        self._listener: socket.socket | None = None
        if self.world > 1:
            if self.rank == 0:
                self._serve()
            else:
                self._connect()

    # ---- set-up -------------------------------------------------------------------------
    def _serve(self) -> None:
        try:
            os.unlink(self.path)  # a stale file of an earlier launch must not be believed
        except FileNotFoundError:
            pass
        ls = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        ls.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        ls.bind(("127.0.0.1", 0))
        ls.listen(self.world)
        ls.settimeout(self.timeout)
        self._listener = ls
        token = secrets.token_bytes(_TOKEN_BYTES)
        tmp = f"{self.path}.{os.getpid()}.tmp"
        try:
            os.unlink(tmp)
        except FileNotFoundError:
            pass
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
        with os.fdopen(fd, "w") as f:
            f.write(f"{ls.getsockname()[1]} {token.hex()}")
        os.replace(tmp, self.path)  # atomic: readers see nothing or the whole record
        peers: list[socket.socket | None] = [None] * self.world
        deadline = time.monotonic() + self.timeout
        while any(p is None for p in peers[1:]):
            if time.monotonic() > deadline:
                raise TimeoutError(f"rendezvous: {sum(p is None for p in peers[1:])} rank(s) never connected")
            conn, _ = ls.accept()
            conn.settimeout(_HANDSHAKE_TIMEOUT)
            try:
                hello = _recv_exact(conn, len(_MAGIC) + _TOKEN_BYTES + 8)
            except (ConnectionError, socket.timeout, OSError):
                conn.close()
                continue
            r, w = struct.unpack("<II", hello[len(_MAGIC) + _TOKEN_BYTES:])
            if (hello[: len(_MAGIC)] != _MAGIC or not secrets.compare_digest(hello[len(_MAGIC): len(_MAGIC) + _TOKEN_BYTES], token)
                    or w != self.world or not (0 < r < self.world) or peers[r] is not None):
                conn.close()
                continue
            conn.settimeout(self.timeout)
            conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            conn.sendall(_MAGIC)
            peers[r] = conn
        self._peers = peers

    def _connect(self) -> None:
        deadline = time.monotonic() + self.timeout
        last = None
        while time.monotonic() < deadline:
            try:
                fd = os.open(self.path, os.O_RDONLY | getattr(os, "O_NOFOLLOW", 0))
                with os.fdopen(fd) as f:
                    if os.fstat(f.fileno()).st_uid != os.getuid():
                        raise PermissionError(f"{self.path} belongs to another user")
                    port_s, token_s = f.read().split()
                token = bytes.fromhex(token_s)
                if len(token) != _TOKEN_BYTES:
                    raise ValueError("malformed rendezvous record")
                s = socket.create_connection(("127.0.0.1", int(port_s)), timeout=_HANDSHAKE_TIMEOUT)
                s.settimeout(_HANDSHAKE_TIMEOUT)  # a stale file whose port was reused must not cost the collective timeout
                s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                s.sendall(_MAGIC + token + struct.pack("<II", self.rank, self.world))
                if _recv_exact(s, len(_MAGIC)) == _MAGIC:
                    s.settimeout(self.timeout)
                    self._up = s
                    return
                s.close()
            except (OSError, ValueError, ConnectionError) as exc:  # not published yet / stale file / refused
                last = exc
            time.sleep(0.05)
        raise TimeoutError(f"rendezvous: rank {self.rank} could not reach rank 0 via {self.path}: {last!r}")

    # ---- collectives (all of them: gather at rank 0, then fan out) ---------------------------
    def allgather(self, data: bytes) -> list[bytes]:
        if self.world == 1:
            return [bytes(data)]
        if self.rank == 0:
            parts = [bytes(data)] + [_recv_msg(p) for p in self._peers[1:]]
            blob = b"".join(struct.pack("<Q", len(x)) + x for x in parts)
            for p in self._peers[1:]:
                _send_msg(p, blob)
            return parts
        _send_msg(self._up, bytes(data))
        blob = _recv_msg(self._up)
        parts, pos = [], 0
        for _ in range(self.world):
            (n,) = struct.unpack_from("<Q", blob, pos)
            parts.append(blob[pos + 8: pos + 8 + n])
            pos += 8 + n
        return parts

    def barrier(self) -> None:
        self.allgather(b"")

    def broadcast(self, data: bytes | None, src: int = 0) -> bytes:
        return self.allgather(bytes(data) if self.rank == src else b"")[src]

    def allreduce_max(self, values) -> list[float]:
        vals = [float(v) for v in values]
        parts = self.allgather(struct.pack(f"<{len(vals)}d", *vals))
        cols = [struct.unpack(f"<{len(vals)}d", p) for p in parts]
        return [max(c[k] for c in cols) for k in range(len(vals))]

    def allreduce_min(self, values) -> list[float]:
        return [-v for v in self.allreduce_max([-float(x) for x in values])]

    def close(self) -> None:
        for s in [self._up, self._listener, *self._peers]:
            if s is not None:
                try:
                    s.close()
                except OSError:
                    pass
        self._up = self._listener = None
        self._peers = []
        if self.rank == 0 and self.world > 1:
            try:
                os.unlink(self.path)
            except OSError:
                pass


class ThreadRendezvous:
    """The same control-plane interface for the ranks of an IN-PROCESS device group (one thread per context:
    hvd_init_devices, `bench.py --single-process`): all-gather of bytes through shared memory and a barrier. Make one group
    with `ThreadRendezvous.group(world)` and hand member r to rank r's thread."""

    class _Shared:
        def __init__(self, world: int):
            self.world = world
            self.slots = [None] * world
            # every wait is bounded (HVD_RDZV_TIMEOUT seconds, default 900): a rank that died before a collective turns into
            # threading.BrokenBarrierError on the others instead of a join() that never returns
            self.barrier = threading.Barrier(world, timeout=float(os.environ.get("HVD_RDZV_TIMEOUT", "900")))

    def __init__(self, rank: int, shared: "ThreadRendezvous._Shared"):
        self.rank, self.world, self._s = rank, shared.world, shared

    @classmethod
    def group(cls, world: int) -> list["ThreadRendezvous"]:
        shared = cls._Shared(world)
        return [cls(r, shared) for r in range(world)]

    def allgather(self, data: bytes) -> list[bytes]:
        s = self._s
        s.slots[self.rank] = bytes(data)
        s.barrier.wait()
        out = list(s.slots)
        s.barrier.wait()  # nobody overwrites a slot before everybody has read all of them
        return out

    def barrier(self) -> None:
        self._s.barrier.wait()

    def abort(self) -> None:
        """A rank that fails calls this on its way out: every rank waiting in (or arriving at) a collective of this group
        gets threading.BrokenBarrierError."""
        self._s.barrier.abort()

    def broadcast(self, data: bytes | None, src: int = 0) -> bytes:
        return self.allgather(data if self.rank == src else b"")[src]

    def allreduce_max(self, values) -> list[float]:
        vals = [float(v) for v in values]
        parts = [struct.unpack(f"<{len(vals)}d", p) for p in self.allgather(struct.pack(f"<{len(vals)}d", *vals))]
        return [max(col) for col in zip(*parts)]

    def allreduce_min(self, values) -> list[float]:
        return [-v for v in self.allreduce_max([-float(v) for v in values])]

    def close(self) -> None:
        pass
