"""Minimal single-node rendezvous for the one-process-per-GPU launch (torch.distributed.run sets
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT): broadcast of the 128-byte RCCL id,
barrier, and a max-reduction of one float, over a star of localhost TCP sockets.

Why not torch.distributed here: the torch wheel ships its own HIP runtime and RCCL; once torch is
imported into the process, ``ncclCommInitRank`` of the system RCCL that libsrx_hip.so uses fails
("unhandled cuda error"), and two HIP runtimes cannot both own the GPU.  The data path needs none
of torch — only this handshake — so the handshake is done with the standard library.

Rank 0 binds an ephemeral port on 127.0.0.1 and publishes it in a file keyed by the launcher
(MASTER_PORT + parent pid, the torchrun agent all ranks share); the others poll for the file.
"""
from __future__ import annotations

import os
import socket
import struct
import tempfile
import time


class StarGroup:
    def __init__(self, rank: int, world: int, key: str | None = None, timeout: float = 300.0):
        self.rank, self.world = int(rank), int(world)
        self._peers: list[socket.socket] = []
        self._up: socket.socket | None = None
        if self.world <= 1:
            return
        key = key or f"{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}_{os.getppid()}"
        path = os.path.join(tempfile.gettempdir(), f"srx_rdzv_{key}.port")
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind(("127.0.0.1", 0))
            srv.listen(self.world)
            tmp = path + f".{os.getpid()}"
            with open(tmp, "w") as f:
                f.write(str(srv.getsockname()[1]))
            os.replace(tmp, path)
            srv.settimeout(timeout)
            slots: dict[int, socket.socket] = {}
            while len(slots) < self.world - 1:
                c, _ = srv.accept()
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                (r,) = struct.unpack("<i", self._recv(c, 4))
                slots[r] = c
            self._peers = [slots[r] for r in sorted(slots)]
            srv.close()
            try:
                os.unlink(path)
            except OSError:
                pass
        else:
            t0 = time.time()
            port = None
            while port is None:
                try:
                    with open(path) as f:
                        port = int(f.read().strip())
                except (OSError, ValueError):
                    if time.time() - t0 > timeout:
                        raise TimeoutError(f"rendezvous file {path} never appeared")
                    time.sleep(0.01)
            s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            while True:
                try:
                    s.connect(("127.0.0.1", port))
                    break
                except OSError:
                    if time.time() - t0 > timeout:
                        raise
                    time.sleep(0.01)
            s.sendall(struct.pack("<i", self.rank))
            self._up = s

    @staticmethod
    def _recv(s: socket.socket, n: int) -> bytes:
        buf = b""
        while len(buf) < n:
            chunk = s.recv(n - len(buf))
            if not chunk:
                raise ConnectionError("rendezvous peer closed the connection")
            buf += chunk
        return buf

    def broadcast_bytes(self, data: bytes | None, n: int) -> bytes:
        """Rank 0's `data` (n bytes) on every rank."""
        if self.world <= 1:
            return data
        if self.rank == 0:
            for p in self._peers:
                p.sendall(data)
            return data
        return self._recv(self._up, n)

    def allreduce_max(self, x: float) -> float:
        if self.world <= 1:
            return x
        if self.rank == 0:
            vals = [x] + [struct.unpack("<d", self._recv(p, 8))[0] for p in self._peers]
            m = max(vals)
            for p in self._peers:
                p.sendall(struct.pack("<d", m))
            return m
        self._up.sendall(struct.pack("<d", x))
        return struct.unpack("<d", self._recv(self._up, 8))[0]

    def allreduce_sum_f64(self, a) -> None:
        """In-place sum of a contiguous float64 numpy array over all ranks (rank order: ((r0 + r1) + r2) ...)."""
        import numpy as np
        if self.world <= 1:
            return
        n = a.size * 8
        if self.rank == 0:
            for p in self._peers:
                a += np.frombuffer(self._recv(p, n), dtype=np.float64)
            data = a.tobytes()
            for p in self._peers:
                p.sendall(data)
        else:
            self._up.sendall(a.tobytes())
            a[:] = np.frombuffer(self._recv(self._up, n), dtype=np.float64)

    def barrier(self) -> None:
        self.allreduce_max(0.0)

    def close(self) -> None:
        for p in self._peers:
            p.close()
        if self._up:
            self._up.close()
        self._peers, self._up = [], None
