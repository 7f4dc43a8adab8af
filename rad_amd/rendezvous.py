"""Plain-TCP process group for one-process-per-GPU runs (no torch, no MPI): rank 0 listens on
MASTER_ADDR:MASTER_PORT, the other ranks connect, and everything is a star through rank 0.

It carries what the HOST side of a multi-GPU run needs — the 128-byte RCCL unique id of the library's
communicator (rad_amd/csrc/comm.hip), barriers, the max / sum of a few floats for the bench line — and,
as a stand-in for RCCL when several ranks rehearse on ONE GPU (`bench.py --exchange host`), the per-step
exchange of the row-sharded traversal.  The product exchange is RCCL on device buffers
(radhip_shard_run); this class never sees a fingerprint.
"""
from __future__ import annotations

import pickle
import socket
import struct
import time
from typing import Any, List

import numpy as np


def _send(sock: socket.socket, payload: bytes) -> None:
    sock.sendall(struct.pack("<Q", len(payload)) + payload)


def _recv(sock: socket.socket) -> bytes:
    hdr = b""
    while len(hdr) < 8:
        chunk = sock.recv(8 - len(hdr))
        if not chunk:
            raise ConnectionError("peer closed the rendezvous socket")
        hdr += chunk
    n = struct.unpack("<Q", hdr)[0]
    buf = bytearray(n)
    view = memoryview(buf)
    got = 0
    while got < n:
        k = sock.recv_into(view[got:], n - got)
        if k == 0:
            raise ConnectionError("peer closed the rendezvous socket")
        got += k
    return bytes(buf)


class TcpGroup:
    def __init__(self, rank: int, world: int, addr: str = "127.0.0.1", port: int = 29500, timeout: float = 120.0):
        self.rank, self.world = int(rank), int(world)
        self._peers: List[socket.socket] = []      # rank 0: sockets of ranks 1..world-1 (index rank-1)
        self._up = None                            # other ranks: socket to rank 0
        if self.world == 1:
            return
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(self.world)
            srv.settimeout(timeout)
            slots = [None] * (self.world - 1)
            for _ in range(self.world - 1):
                conn, _a = srv.accept()
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                conn.settimeout(None)
                r = struct.unpack("<I", _recv(conn))[0]
                if not (1 <= r < self.world) or slots[r - 1] is not None:
                    raise RuntimeError(f"rendezvous: unexpected rank {r}")
                slots[r - 1] = conn
            srv.close()
            self._peers = slots
        else:
            deadline = time.time() + timeout
            while True:
                try:
                    s = socket.create_connection((addr, port), timeout=5.0)
                    break
                except OSError:
                    if time.time() > deadline:
                        raise TimeoutError(f"rank {self.rank}: no rendezvous server at {addr}:{port}")
                    time.sleep(0.05)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.settimeout(None)
            _send(s, struct.pack("<I", self.rank))
            self._up = s

    # ---- collectives on picklable objects (small control data)
    def allgather_obj(self, obj: Any) -> List[Any]:
        if self.world == 1:
            return [obj]
        if self.rank == 0:
            allv = [obj] + [pickle.loads(_recv(p)) for p in self._peers]
            blob = pickle.dumps(allv)
            for p in self._peers:
                _send(p, blob)
            return allv
        _send(self._up, pickle.dumps(obj))
        return pickle.loads(_recv(self._up))

    def broadcast_obj(self, obj: Any = None) -> Any:
        """Rank 0's object on every rank."""
        return self.allgather_obj(obj if self.rank == 0 else None)[0]

    def barrier(self) -> None:
        self.allgather_obj(None)

    def allreduce(self, values, op: str = "sum") -> np.ndarray:
        allv = np.asarray(self.allgather_obj(np.asarray(values, np.float64)))
        return allv.max(0) if op == "max" else allv.min(0) if op == "min" else allv.sum(0)

    # ---- array collectives of the host-staged sharded step (rehearsal on one GPU)
    def allgather_u32(self, a: np.ndarray) -> np.ndarray:
        a = np.ascontiguousarray(a, np.uint32)
        if self.world == 1:
            return a[None].copy()
        if self.rank == 0:
            parts = [a.tobytes()] + [_recv(p) for p in self._peers]
            blob = b"".join(parts)
            for p in self._peers:
                _send(p, blob)
        else:
            _send(self._up, a.tobytes())
            blob = _recv(self._up)
        return np.frombuffer(blob, np.uint32).reshape((self.world,) + a.shape).copy()

    def reduce_scatter_sum_u32(self, a: np.ndarray) -> np.ndarray:
        """a: [world, ...] per rank; returns sum over ranks of block `rank`."""
        a = np.ascontiguousarray(a, np.uint32)
        if self.world == 1:
            return a[0].copy()
        if self.rank == 0:
            tot = a.copy()
            for p in self._peers:
                tot += np.frombuffer(_recv(p), np.uint32).reshape(a.shape)
            for r, p in enumerate(self._peers, start=1):
                _send(p, tot[r].tobytes())
            return tot[0]
        _send(self._up, a.tobytes())
        return np.frombuffer(_recv(self._up), np.uint32).reshape(a.shape[1:]).copy()

    def close(self) -> None:
        for p in self._peers:
            try:
                p.close()
            except OSError:
                pass
        if self._up is not None:
            try:
                self._up.close()
            except OSError:
                pass
        self._peers, self._up = [], None


def exchange_fds(rank: int, world: int, fd: int, key: str, timeout: float = 120.0) -> List[int]:
    """Every rank's file descriptor on every rank (the dmabuf descriptors of the row shards of a peer-mapped corpus,
    rad_amd.device.DeviceIndex.peer_export).  Descriptors cross process boundaries only as SCM_RIGHTS messages over a Unix
    domain socket: rank 0 listens on the abstract address `\\0radhip-<key>`, collects one descriptor per rank and sends all of
    them to every rank.  Returns [fd of rank 0, ..., fd of rank world - 1] (this rank's own entry is `fd` itself); the caller
    closes the received ones after importing them."""
    if world == 1:
        return [fd]
    addr = "\0radhip-" + key
    if rank == 0:
        srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
        srv.bind(addr)
        srv.listen(world)
        srv.settimeout(timeout)
        conns, fds = [None] * world, [None] * world
        fds[0] = fd
        for _ in range(world - 1):
            c, _a = srv.accept()
            c.settimeout(timeout)
            msg, got, _flags, _addr = socket.recv_fds(c, 16, 1)
            r = struct.unpack("<I", msg[:4])[0]
            if not (1 <= r < world) or conns[r] is not None or len(got) != 1:
                raise RuntimeError(f"descriptor exchange: unexpected message from rank {r}")
            conns[r], fds[r] = c, got[0]
        srv.close()
        for r in range(1, world):
            socket.send_fds(conns[r], [struct.pack("<I", world)], fds)
            conns[r].close()
        return fds
    deadline = time.time() + timeout
    s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    while True:
        try:
            s.connect(addr)
            break
        except OSError:
            if time.time() > deadline:
                raise TimeoutError(f"rank {rank}: nobody listens on the descriptor exchange {key!r}")
            time.sleep(0.05)
    s.settimeout(timeout)
    socket.send_fds(s, [struct.pack("<I", rank)], [fd])
    _msg, got, _flags, _addr = socket.recv_fds(s, 16, world)
    s.close()
    if len(got) != world:
        raise RuntimeError(f"descriptor exchange: {len(got)} descriptors for {world} ranks")
    import os as _os
    _os.close(got[rank])       # (a duplicate of this rank's own descriptor)
    got[rank] = fd
    return list(got)
