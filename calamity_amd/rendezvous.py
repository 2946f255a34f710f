"""Rendezvous of the ranks of one job over plain TCP sockets -- what a launcher-started multi-rank run needs beside the data path:
hand the 128-byte RCCL id from rank 0 to the others, a barrier, and small reductions of host scalars (the max of a wall time, the
sum of a few counts).  No PyTorch, no MPI: rank 0 listens on MASTER_ADDR (on MASTER_PORT when that is free -- the variables every
launcher exports; see SocketGroup for launchers that keep the port to themselves), the other ranks connect, and every operation is a
gather to rank 0 followed by a scatter of the result.  The per-step exchange of a
fit never goes through here -- that is RCCL inside the library (`cal_solver_comm_init`) -- except as the functional stand-in for
ranks that share one GPU (`all_reduce_inplace`, bench.py --transport host), which moves a few megabytes per step and is not
measured.

The reference has no counterpart (one device per process: calibration.py:1796-1804)."""
import os
import pickle
import socket
import struct
import tempfile
import time

import numpy as np

_MAGIC = b"CALRDZV1"


def _send(sock, payload: bytes):
    sock.sendall(struct.pack("<Q", len(payload)) + payload)


def _exactly(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(min(1 << 20, n - len(buf)))
        if not chunk:
            raise ConnectionError("rendezvous: a peer closed its connection")
        buf += chunk
    return bytes(buf)


def _recv(sock) -> bytes:
    (n,) = struct.unpack("<Q", _exactly(sock, 8))
    return _exactly(sock, n)


def _port_file(port):
    """Where rank 0 publishes the port it really listens on (ranks of one job run on one node: the contract of bench.py)."""
    return os.path.join(tempfile.gettempdir(), f"calamity_rdzv_{os.getuid()}_{port}.port")


class SocketGroup:
    """The ranks of one job.  ``rank`` / ``world`` / address default to RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT.

    MASTER_PORT names the job, it is not necessarily the port of this group: under ``python -m torch.distributed.run`` the
    launcher's own store LISTENS on MASTER_PORT (the workers are meant to join it as clients), so rank 0 could not bind it.  Rank 0
    therefore takes MASTER_PORT when it is free and any free port otherwise, and publishes the port it got in a small file named
    after MASTER_PORT; the other ranks read the file, connect, and both sides exchange a magic word (a stale file of an earlier job
    points at a dead port or at somebody else's server: the connection or the handshake fails and the rank looks again)."""

    def __init__(self, rank=None, world=None, addr=None, port=None, timeout=120.0):
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = int(port or os.environ.get("MASTER_PORT", "29611"))
        self.peers = []  # rank 0: sockets of ranks 1..world-1 in rank order; others: [socket to rank 0]
        if self.world == 1:
            return
        path = _port_file(port)
        deadline = time.monotonic() + timeout
        if self.rank == 0:
            try:
                os.remove(path)  # (an earlier job's)
            except OSError:
                pass
            srv = socket.socket()
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            try:
                srv.bind((addr, port))
            except OSError:  # in use: the launcher's store
                srv.bind((addr, 0))
            srv.listen(self.world)
            with open(path + f".{os.getpid()}", "w") as f:
                f.write(f"{srv.getsockname()[1]}\n")
            os.replace(path + f".{os.getpid()}", path)
            got = {}
            try:
                while len(got) < self.world - 1:
                    srv.settimeout(max(0.1, deadline - time.monotonic()))
                    try:
                        c, _ = srv.accept()
                    except socket.timeout:
                        raise TimeoutError(f"rendezvous: {len(got) + 1} of {self.world} ranks met within {timeout:.0f} s") from None
                    try:
                        c.settimeout(5.0)
                        hello = _exactly(c, len(_MAGIC) + 4)
                        if hello[: len(_MAGIC)] != _MAGIC:
                            raise ConnectionError("not a rank of this job")
                        c.sendall(_MAGIC)
                    except (OSError, ConnectionError):
                        c.close()
                        continue
                    c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    c.settimeout(None)
                    got[struct.unpack("<I", hello[len(_MAGIC):])[0]] = c
            finally:
                srv.close()
                try:
                    os.remove(path)
                except OSError:
                    pass
            self.peers = [got[r] for r in range(1, self.world)]
        else:
            c = None
            while c is None:
                if time.monotonic() > deadline:
                    raise TimeoutError(f"rendezvous: rank {self.rank} did not find rank 0 within {timeout:.0f} s ({path})")
                try:
                    with open(path) as f:
                        real = int(f.read().strip())
                    c = socket.create_connection((addr, real), timeout=2.0)
                    c.settimeout(5.0)
                    c.sendall(_MAGIC + struct.pack("<I", self.rank))
                    if _exactly(c, len(_MAGIC)) != _MAGIC:
                        raise ConnectionError("not rank 0 of this job")
                except (OSError, ValueError, ConnectionError):
                    if c is not None:
                        c.close()
                        c = None
                    time.sleep(0.05)
            c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            c.settimeout(None)
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
