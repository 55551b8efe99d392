"""A minimal host-side process group over TCP — what the multi-GPU path needs from the host and nothing
else: ship the 128-byte RCCL unique id, agree on success, move the (small) boundary records when RCCL is
not available, barrier.  Plain sockets, no PyTorch, no MPI.

Topology: a star through rank 0 (``world`` <= the GPUs of one node; the payloads are a few hundred KB at
most).  Rendezvous: rank 0 listens on ``addr:port`` (default ``MASTER_ADDR`` / ``MASTER_PORT + 23`` — next to,
not on, the port a launcher such as ``torch.distributed.run`` uses for its own store).
"""
from __future__ import annotations

import os
import socket
import struct
import time


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("peer closed the connection")
        buf += chunk
    return bytes(buf)


def _send_msg(sock, payload: bytes):
    sock.sendall(struct.pack("<Q", len(payload)) + payload)


MAX_MESSAGE = 1 << 30  # the largest payload is an all-gather of boundary records (a few MB): anything beyond 1 GiB is
                        # a corrupt or hostile length word, not a message


def _recv_msg(sock) -> bytes:
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    if n > MAX_MESSAGE:
        raise ConnectionError(f"peer announced a {n}-byte message (limit {MAX_MESSAGE})")
    return _recv_exact(sock, n)


class HostGroup:
    def __init__(self, rank: int = None, world: int = None, addr: str = None, port: int = None, timeout: float = 300.0):
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        if port is None:
            port = int(os.environ.get("PFD_HOSTGROUP_PORT", "0")) or int(os.environ.get("MASTER_PORT", "29500")) + 23
        self._peers = {}
        self._sock = None
        if not 0 <= self.rank < self.world:
            raise ValueError(f"rank {self.rank} outside a world of {self.world}")
        if self.world == 1:
            return
        # 16-byte job token (optional shared secret PFD_HOSTGROUP_TOKEN): a connection that does not present it is
        # dropped at accept time instead of becoming a rank
        import hashlib
        import hmac

        secret = os.environ.get("PFD_HOSTGROUP_TOKEN", "")
        token = hashlib.sha256(("pfd-hostgroup:" + secret).encode()).digest()[:16]
        if not secret and self.rank == 0 and addr not in ("127.0.0.1", "localhost", "::1"):
            import warnings

            warnings.warn("HostGroup: no PFD_HOSTGROUP_TOKEN is set — the job token is derived from an empty secret and "
                          f"authenticates nothing on the non-loopback address {addr}", RuntimeWarning, stacklevel=2)
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(self.world)
            srv.settimeout(timeout)
            try:
                while len(self._peers) < self.world - 1:
                    conn, _ = srv.accept()
                    conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    conn.settimeout(timeout)
                    try:
                        (r,) = struct.unpack("<I", _recv_exact(conn, 4))
                        tok = _recv_exact(conn, len(token))
                    except (OSError, ConnectionError):  # not one of ours (port scan, wrong protocol)
                        conn.close()
                        continue
                    if not (1 <= r < self.world) or r in self._peers or not hmac.compare_digest(tok, token):
                        conn.close()  # out of range / duplicate rank / wrong job: never part of the group
                        continue
                    self._peers[r] = conn
            finally:
                srv.close()
        else:
            deadline = time.time() + timeout
            while True:
                try:
                    s = socket.create_connection((addr, port), timeout=timeout)
                    break
                except OSError:
                    if time.time() > deadline:
                        raise
                    time.sleep(0.05)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.settimeout(timeout)
            s.sendall(struct.pack("<I", self.rank) + token)
            self._sock = s

    # -- collectives (every rank calls them in the same order) ---------------------------------
    def allgather(self, payload: bytes) -> list:
        """Every rank contributes one byte string; returns the list of all of them, in rank order."""
        if self.world == 1:
            return [bytes(payload)]
        if self.rank == 0:
            parts = [bytes(payload)] + [None] * (self.world - 1)
            for r, conn in self._peers.items():
                parts[r] = _recv_msg(conn)
            blob = b"".join(struct.pack("<Q", len(p)) + p for p in parts)
            for conn in self._peers.values():
                _send_msg(conn, blob)
            return parts
        _send_msg(self._sock, bytes(payload))
        blob = _recv_msg(self._sock)
        parts, off = [], 0
        for _ in range(self.world):
            (n,) = struct.unpack_from("<Q", blob, off)
            parts.append(blob[off + 8:off + 8 + n])
            off += 8 + n
        return parts

    def bcast(self, payload: bytes, src: int = 0) -> bytes:
        return self.allgather(payload if self.rank == src else b"")[src]

    def allreduce(self, value, op="max"):
        """Reduce one Python float / int over the ranks (op: "max", "min", "sum")."""
        is_int = isinstance(value, int)
        vals = [struct.unpack("<d" if len(p) == 9 and p[:1] == b"f" else "<q", p[1:])[0]
                for p in self.allgather((b"i" + struct.pack("<q", value)) if is_int else (b"f" + struct.pack("<d", float(value))))]
        return {"max": max, "min": min, "sum": sum}[op](vals)

    def barrier(self):
        self.allgather(b"")

    def close(self):
        for c in self._peers.values():
            try:
                c.close()
            except OSError:
                pass
        if self._sock is not None:
            try:
                self._sock.close()
            except OSError:
                pass
        self._peers, self._sock = {}, None


class TorchGroup:
    """The same interface on top of an initialised ``torch.distributed`` process group (any backend; ``gloo``
    on CPU is enough) — for drivers launched by ``torch.distributed.run``, whose rendezvous store is already
    there.  PyTorch is imported here only."""

    def __init__(self, group=None):
        import torch.distributed as dist

        self._dist, self._group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

    def allgather(self, payload: bytes) -> list:
        if self.world == 1:
            return [bytes(payload)]
        out = [None] * self.world
        self._dist.all_gather_object(out, bytes(payload), group=self._group)
        return out

    def bcast(self, payload: bytes, src: int = 0) -> bytes:
        return self.allgather(payload if self.rank == src else b"")[src]

    def allreduce(self, value, op="max"):
        return {"max": max, "min": min, "sum": sum}[op](self.allgather_values(value))

    def allgather_values(self, value):
        if self.world == 1:
            return [value]
        out = [None] * self.world
        self._dist.all_gather_object(out, value, group=self._group)
        return out

    def barrier(self):
        if self.world > 1:
            self._dist.barrier(group=self._group)

    def close(self):
        pass
