"""Host-side process group: rank / size, small collectives and point-to-point of host
buffers.  It replaces the handful of mpi4py calls the reference makes outside its MPI
datatypes (bcast, allreduce, Send/Recv of numpy arrays).

Two implementations:
  * ``SingleWorld``  - one process.
  * ``SocketWorld``  - one process per GPU, started by any launcher that exports RANK,
    WORLD_SIZE, MASTER_ADDR and MASTER_PORT (``python -m torch.distributed.run`` does).  Plain
    TCP sockets in a full mesh; it carries the RCCL unique id at start-up, scalar reductions
    and the host-staged test transport.  Bulk GPU data goes through RCCL inside libsophtmi.
    The worker processes never import torch: a second HIP runtime in the process (torch
    bundles its own) is exactly what this library must not share an address space with.
"""
import io
import json
import os
import socket
import struct
import threading
import time

import numpy as np

_PORT_OFFSET = 101  # the launcher's own store listens on MASTER_PORT


class SingleWorld:
    rank = 0
    size = 1

    def bcast_bytes(self, data, root=0):
        return data

    def bcast_object(self, obj, root=0):
        return obj

    def allreduce(self, value, op="sum"):
        return value

    def barrier(self):
        pass

    def sendrecv(self, send_arrays, recv_arrays):
        """send_arrays / recv_arrays: dict peer -> numpy array.  Nothing to do on one rank."""
        assert not send_arrays and not recv_arrays


def encode_object(obj):
    """Wire form of what the ranks broadcast to each other: None, a numpy array (npy format,
    no pickled object arrays) or plain data (JSON).  Nothing received is ever unpickled."""
    if obj is None:
        return b"N"
    if isinstance(obj, np.ndarray):
        buf = io.BytesIO()
        np.save(buf, obj, allow_pickle=False)
        return b"A" + buf.getvalue()
    if isinstance(obj, np.generic):
        obj = obj.item()
    return b"J" + json.dumps(obj).encode()


def decode_object(data):
    tag, body = data[:1], data[1:]
    if tag == b"N":
        return None
    if tag == b"A":
        return np.load(io.BytesIO(body), allow_pickle=False)
    if tag == b"J":
        return json.loads(body.decode())
    raise ValueError("unknown object tag in broadcast")


def _recv_exact(sock, n):
    buf = bytearray(n)
    view = memoryview(buf)
    got = 0
    while got < n:
        k = sock.recv_into(view[got:], n - got)
        if k == 0:
            raise ConnectionError("peer closed the connection")
        got += k
    return buf


class SocketWorld:
    def __init__(self, rank=None, size=None, addr=None, port=None, timeout=120.0):
        self.rank = int(os.environ["RANK"]) if rank is None else rank
        self.size = int(os.environ["WORLD_SIZE"]) if size is None else size
        addr = os.environ.get("MASTER_ADDR", "127.0.0.1") if addr is None else addr
        base = (int(os.environ.get("MASTER_PORT", "29500")) if port is None else port) + _PORT_OFFSET
        self._peers = {}
        server = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        server.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        host = "0.0.0.0" if addr not in ("127.0.0.1", "localhost") else "127.0.0.1"
        # the port may be held for a moment by somebody's outgoing connection (launchers pick rendezvous ports from
        # the range the system hands out to clients): try for a while before giving up
        deadline = time.time() + min(timeout, 20.0)
        while True:
            try:
                server.bind((host, base + self.rank))
                break
            except OSError:
                if time.time() > deadline:
                    raise
                time.sleep(0.1)
        server.listen(self.size)
        # connect to every lower rank, accept every higher rank
        for peer in range(self.rank):
            deadline = time.time() + timeout
            while True:
                try:
                    s = socket.create_connection((addr, base + peer), timeout=5.0)
                    break
                except OSError:
                    if time.time() > deadline:
                        raise
                    time.sleep(0.05)
            s.settimeout(None)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.sendall(struct.pack("<i", self.rank))
            self._peers[peer] = s
        server.settimeout(timeout)
        for _ in range(self.size - 1 - self.rank):
            s, _addr = server.accept()
            s.settimeout(None)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            (peer,) = struct.unpack("<i", bytes(_recv_exact(s, 4)))
            self._peers[peer] = s
        server.close()

    # -- point to point ---------------------------------------------------------------
    def _send(self, peer, data):
        self._peers[peer].sendall(struct.pack("<q", len(data)))
        self._peers[peer].sendall(data)

    def _recv(self, peer):
        (n,) = struct.unpack("<q", bytes(_recv_exact(self._peers[peer], 8)))
        return _recv_exact(self._peers[peer], n)

    def sendrecv(self, send_arrays, recv_arrays):
        """Exchange host numpy arrays with peers: sends run on helper threads, receives here."""
        threads = []
        for peer, arr in send_arrays.items():
            data = memoryview(np.ascontiguousarray(arr)).cast("B")
            t = threading.Thread(target=self._send, args=(peer, data))
            t.start()
            threads.append(t)
        for peer, arr in recv_arrays.items():
            buf = self._recv(peer)
            flat = arr.reshape(-1).view(np.uint8)
            if len(buf) != flat.size:
                raise ValueError(f"size mismatch in sendrecv from rank {peer}: {len(buf)} != {flat.size}")
            flat[:] = np.frombuffer(buf, dtype=np.uint8)
        for t in threads:
            t.join()

    # -- collectives --------------------------------------------------------------------
    def bcast_bytes(self, data, root=0):
        if self.rank == root:
            for peer in self._peers:
                self._send(peer, data)
            return bytes(data)
        return bytes(self._recv(root))

    def bcast_object(self, obj, root=0):
        return decode_object(self.bcast_bytes(encode_object(obj) if self.rank == root else b"", root))

    def allreduce(self, value, op="sum"):
        fn = {"sum": sum, "min": min, "max": max}[op]
        if self.rank == 0:
            vals = [float(value)] + [struct.unpack("<d", bytes(self._recv(p)))[0] for p in sorted(self._peers)]
            out = float(fn(vals))
            for p in self._peers:
                self._send(p, struct.pack("<d", out))
            return out
        self._send(0, struct.pack("<d", float(value)))
        return struct.unpack("<d", bytes(self._recv(0)))[0]

    def barrier(self):
        self.allreduce(0.0, "sum")

    def close(self):
        for s in self._peers.values():
            try:
                s.close()
            except OSError:
                pass
        self._peers = {}


_default = None


def get_world():
    """World of this process: multi-rank when launched with WORLD_SIZE > 1, else single."""
    global _default
    if _default is None:
        _default = SocketWorld() if int(os.environ.get("WORLD_SIZE", "1")) > 1 else SingleWorld()
    return _default


def set_world(world):
    global _default
    _default = world
