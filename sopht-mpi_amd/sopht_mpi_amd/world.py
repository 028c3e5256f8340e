"""Host-side process group: rank / size, small collectives and point-to-point of host
buffers.  It replaces the handful of mpi4py calls the reference makes outside its MPI
datatypes (bcast, allreduce, Send/Recv of numpy arrays).

Two implementations:
  * ``SingleWorld``  - one process.
  * ``TorchWorld``   - one process per GPU, launched by ``torch.distributed.run``; uses a
    gloo group on CPU tensors only.  It carries the RCCL unique id at start-up and the
    host-staged transport; bulk GPU data goes through RCCL inside libsophtmi.
"""
import os

import numpy as np


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


class TorchWorld:
    def __init__(self):
        import torch.distributed as dist

        self._dist = dist
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend="gloo")
        self.rank = dist.get_rank()
        self.size = dist.get_world_size()

    def bcast_bytes(self, data, root=0):
        import torch

        n = len(data)
        t = torch.frombuffer(bytearray(data), dtype=torch.uint8).clone() if self.rank == root else torch.zeros(
            n, dtype=torch.uint8
        )
        self._dist.broadcast(t, src=root)
        return bytes(t.numpy().tobytes())

    def bcast_object(self, obj, root=0):
        box = [obj]
        self._dist.broadcast_object_list(box, src=root)
        return box[0]

    def allreduce(self, value, op="sum"):
        import torch

        t = torch.tensor([float(value)], dtype=torch.float64)
        rop = {"sum": self._dist.ReduceOp.SUM, "min": self._dist.ReduceOp.MIN, "max": self._dist.ReduceOp.MAX}[op]
        self._dist.all_reduce(t, op=rop)
        return float(t.item())

    def barrier(self):
        self._dist.barrier()

    def sendrecv(self, send_arrays, recv_arrays):
        """Exchange host numpy arrays with peers (non-blocking pairs, then wait)."""
        import torch

        reqs, recv_t = [], {}
        for peer, arr in recv_arrays.items():
            t = torch.from_numpy(arr.reshape(-1).view(np.uint8))
            recv_t[peer] = t
            reqs.append(self._dist.irecv(t, src=peer))
        for peer, arr in send_arrays.items():
            t = torch.from_numpy(np.ascontiguousarray(arr).reshape(-1).view(np.uint8))
            reqs.append(self._dist.isend(t, dst=peer))
        for r in reqs:
            r.wait()


_default = None


def get_world():
    """World of this process: multi-rank when launched with WORLD_SIZE > 1, else single."""
    global _default
    if _default is None:
        _default = TorchWorld() if int(os.environ.get("WORLD_SIZE", "1")) > 1 else SingleWorld()
    return _default


def set_world(world):
    global _default
    _default = world
