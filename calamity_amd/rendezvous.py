"""Rendezvous of the ranks of one job over plain TCP sockets -- what a launcher-started multi-rank run needs beside the data path:
hand the 128-byte RCCL id from rank 0 to the others, a barrier, and small reductions of host scalars (the max of a wall time, the
sum of a few counts).  No PyTorch, no MPI: rank 0 listens on MASTER_ADDR:MASTER_PORT (the variables every launcher exports), the
other ranks connect, and every operation is a gather to rank 0 followed by a scatter of the result.  The per-step exchange of a
fit never goes through here -- that is RCCL inside the library (`cal_solver_comm_init`) -- except as the functional stand-in for
ranks that share one GPU (`all_reduce_inplace`, bench.py --transport host), which moves a few megabytes per step and is not
measured.

The reference has no counterpart (one device per process: calibration.py:1796-1804)."""
import os
import pickle
import socket
import struct
import time

import numpy as np


def _send(sock, payload: bytes):
    sock.sendall(struct.pack("<Q", len(payload)) + payload)


def _recv(sock) -> bytes:
    def exactly(n):
        buf = bytearray()
        while len(buf) < n:
            chunk = sock.recv(min(1 << 20, n - len(buf)))
            if not chunk:
                raise ConnectionError("rendezvous: a peer closed its connection")
            buf += chunk
        return bytes(buf)

    (n,) = struct.unpack("<Q", exactly(8))
    return exactly(n)


class SocketGroup:
    """The ranks of one job.  ``rank`` / ``world`` / address default to RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT."""

    def __init__(self, rank=None, world=None, addr=None, port=None, timeout=120.0):
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = int(port or os.environ.get("MASTER_PORT", "29611"))
        self.peers = []  # rank 0: sockets of ranks 1..world-1 in rank order; others: [socket to rank 0]
        if self.world == 1:
            return
        if self.rank == 0:
            srv = socket.socket()
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr, port))
            srv.listen(self.world)
            srv.settimeout(timeout)
            got = {}
            while len(got) < self.world - 1:
                c, _ = srv.accept()
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                c.settimeout(None)
                (r,) = struct.unpack("<I", c.recv(4))
                got[r] = c
            srv.close()
            self.peers = [got[r] for r in range(1, self.world)]
        else:
            deadline = time.monotonic() + timeout
            while True:
                try:
                    c = socket.create_connection((addr, port), timeout=5.0)
                    break
                except OSError:
                    if time.monotonic() > deadline:
                        raise
                    time.sleep(0.05)
            c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            c.settimeout(None)
            c.sendall(struct.pack("<I", self.rank))
            self.peers = [c]

    # every collective: the others send to rank 0, rank 0 combines and answers
    def _gather_scatter(self, mine, combine):
        if self.world == 1:
            return combine([mine])
        if self.rank == 0:
            parts = [mine] + [pickle.loads(_recv(c)) for c in self.peers]
            out = combine(parts)
            blob = pickle.dumps(out, protocol=4)
            for c in self.peers:
                _send(c, blob)
            return out
        _send(self.peers[0], pickle.dumps(mine, protocol=4))
        return pickle.loads(_recv(self.peers[0]))

    def barrier(self):
        self._gather_scatter(None, lambda parts: None)

    def broadcast(self, obj, src=0):
        """``obj`` of rank ``src`` on every rank (the RCCL unique id)."""
        return self._gather_scatter(obj if self.rank == src else None, lambda parts: parts[src])

    def all_reduce(self, values, op="sum"):
        """Element-wise sum / min / max over the ranks of a small array of host numbers; returns a new array."""
        fn = {"sum": np.sum, "min": np.min, "max": np.max}[op]
        return self._gather_scatter(np.asarray(values), lambda parts: fn(np.stack(parts), axis=0))

    def all_reduce_inplace(self, arr, op="sum"):
        """The exchange-hook form (cal_solver_set_exchange_hook): the library's staging buffer, reduced in place."""
        arr[...] = self.all_reduce(arr, op)

    def close(self):
        for c in self.peers:
            try:
                c.close()
            except OSError:
                pass
        self.peers = []
