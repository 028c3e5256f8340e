"""Device context and device-resident fields.

The reference's fields are caller-owned numpy arrays (with ghost cells) that operators
mutate in place.  Here they live in HBM; ``DeviceField`` keeps the numpy *calling
convention* (``field[...] = values``, ``field[idx]``, ``np.asarray(field)``,
``field[0]`` for a vector component, ``.view()``) with an explicit cost model: every
host access is a PCIe copy, every operator call is device-only.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import call, lib


def _dtype_code(real_t):
    dt = np.dtype(real_t)
    if dt == np.float32:
        return _lib.SMI_F32
    if dt == np.float64:
        return _lib.SMI_F64
    raise ValueError(f"unsupported real type {real_t}")


def local_device():
    """GPU this process drives: LOCAL_RANK modulo the visible device count."""
    n = C.c_int(0)
    call("smi_device_count", C.byref(n))
    if n.value < 1:
        raise _lib.SophtMIError("no HIP device visible")
    return int(os.environ.get("LOCAL_RANK", "0")) % n.value


class DeviceContext:
    """One GPU holding one block of the grid (a z-slab, or a pencil of a (ranks along z, ranks along y) grid):
    geometry + HIP stream + (optionally) RCCL comm."""

    def __init__(self, local_grid_size, ghost_size, real_t, rank=0, nranks=1, device=None, world=None,
                 rank_grid=None, block=None):
        self.real_t = np.dtype(real_t).type
        self.local_grid_size = tuple(int(n) for n in local_grid_size)
        self.ghost_size = int(ghost_size)
        self.rank, self.nranks = int(rank), int(nranks)
        self.world = world
        self.device = local_device() if device is None else device
        nz, ny, nx = self.local_grid_size
        h = C.c_void_p()
        if block is None:
            call("smi_ctx_create", C.byref(h), self.device, _dtype_code(real_t), nz, ny, nx, self.ghost_size,
                 self.rank, self.nranks)
        else:
            coords, topo, periodic = block
            call("smi_ctx_create_block", C.byref(h), self.device, _dtype_code(real_t), nz, ny, nx, self.ghost_size,
                 self.rank, self.nranks, (C.c_int * 3)(*(int(v) for v in coords)), (C.c_int * 3)(*(int(v) for v in topo)),
                 (C.c_int * 3)(*(int(bool(v)) for v in periodic)))
        self.handle = h
        # block = (coords, topology, periodic flags), each (z, y, x): a general decomposition, on which the library
        # serves the ghost exchange only (smi_ctx_set_block)
        self.block = block
        if block is not None:
            rank_grid = None
        grid = (self.nranks, 1, 1) if rank_grid is None else tuple(int(v) for v in rank_grid)
        self.rank_grid = grid if len(grid) == 3 else grid + (1,)
        self.rank_z, rest = divmod(self.rank, self.rank_grid[1] * self.rank_grid[2])
        self.rank_y, self.rank_x = divmod(rest, self.rank_grid[2])
        if self.rank_grid[1] > 1 or self.rank_grid[2] > 1:
            call("smi_ctx_set_grid3", h, self.rank_z, self.rank_grid[0], self.rank_y, self.rank_grid[1], self.rank_x,
                 self.rank_grid[2])
        self.padded_shape = tuple(n + 2 * self.ghost_size for n in self.local_grid_size)
        self.transport = "none" if nranks == 1 else None
        self._cb = None
        if os.environ.get("SOPHT_MI_ORDERED_SPREAD", "0") not in ("", "0"):
            self.set_ordered_spread(True)

    def set_ordered_spread(self, on=True):
        """Lagrangian -> Eulerian spreading on this context adds the nodes one after the other in index order (the
        reference's serial loop: reproducible, in the reference's order, slow) instead of with floating-point
        atomics.  For parity runs; also switched on by SOPHT_MI_ORDERED_SPREAD=1."""
        call("smi_set_ordered_spread", self.handle, 1 if on else 0)
        self.ordered_spread = bool(on)

    # -- transports ---------------------------------------------------------------
    def init_rccl(self):
        """RCCL communicator over the slab ranks (one process per GPU)."""
        buf = C.create_string_buffer(128)
        if self.rank == 0:
            call("smi_comm_unique_id", buf)
        uid = self.world.bcast_bytes(buf.raw, root=0)
        call("smi_comm_init", self.handle, C.create_string_buffer(uid, 128))
        self.transport = "rccl"

    def init_host_staged(self):
        """Host-staged plane transport through the world's point-to-point (ranks sharing a GPU)."""
        g, pz = self.ghost_size, self.padded_shape[0]
        plane = self.padded_shape[1] * self.padded_shape[2]
        plane_of_ranks = self.rank_grid[1] * self.rank_grid[2]
        prev = self.rank - plane_of_ranks if self.rank_z > 0 else None             # the z neighbours
        nxt = self.rank + plane_of_ranks if self.rank_z < self.rank_grid[0] - 1 else None
        dt = np.dtype(self.real_t)

        def exchange(_user, field, ncomp, kind):
            try:
                send, recv = {}, {}
                # kind 0 (halo): interior edge planes go out, ghost planes come in
                # kind 16 d (halo of depth d < g): only the d planes next to the interior
                # kind 1 (ghost sum): ghost planes go out, are ADDED to the neighbour's edge planes
                d = kind // 16 if kind >= 16 else g
                accumulate = 1 if kind == 1 else 0
                n = ncomp * d * plane
                src_lo, src_hi = (g, pz - g - d) if kind != 1 else (0, pz - g)
                dst_lo, dst_hi = (g - d, pz - g) if kind != 1 else (g, pz - 2 * g)
                if prev is not None:
                    send[prev] = np.empty(n, dtype=dt)
                    call("smi_planes_d2h", self.handle, field, ncomp, src_lo, d, send[prev].ctypes.data)
                    recv[prev] = np.empty(n, dtype=dt)
                if nxt is not None:
                    send[nxt] = np.empty(n, dtype=dt)
                    call("smi_planes_d2h", self.handle, field, ncomp, src_hi, d, send[nxt].ctypes.data)
                    recv[nxt] = np.empty(n, dtype=dt)
                self.world.sendrecv(send, recv)
                if prev is not None:
                    call("smi_planes_h2d", self.handle, field, ncomp, dst_lo, d, recv[prev].ctypes.data, accumulate)
                if nxt is not None:
                    call("smi_planes_h2d", self.handle, field, ncomp, dst_hi, d, recv[nxt].ctypes.data, accumulate)
                return 0
            except Exception as exc:  # never let an exception cross the C boundary
                import traceback

                traceback.print_exc()
                self._cb_error = exc
                return 1

        def alltoall(_user, send, recv, nbytes):
            try:
                P, r = self.nranks, self.rank
                host = np.empty(P * nbytes, dtype=np.uint8)
                call("smi_memcpy_d2h", self.handle, host.ctypes.data, send, P * nbytes)
                out = np.empty(P * nbytes, dtype=np.uint8)
                out[r * nbytes : (r + 1) * nbytes] = host[r * nbytes : (r + 1) * nbytes]
                sends = {p: host[p * nbytes : (p + 1) * nbytes] for p in range(P) if p != r}
                recvs = {p: out[p * nbytes : (p + 1) * nbytes] for p in range(P) if p != r}
                self.world.sendrecv(sends, recvs)
                call("smi_memcpy_h2d", self.handle, recv, out.ctypes.data, P * nbytes)
                return 0
            except Exception as exc:
                import traceback

                traceback.print_exc()
                self._cb_error = exc
                return 1

        def sendrecv(_user, peer_lo, peer_hi, send_lo, send_hi, recv_lo, recv_hi, nbytes):
            """Device buffers <-> the two y neighbours (peer -1: none)."""
            try:
                sends, recvs = {}, {}
                for peer, src in ((peer_lo, send_lo), (peer_hi, send_hi)):
                    if peer >= 0:
                        sends[peer] = np.empty(nbytes, dtype=np.uint8)
                        call("smi_memcpy_d2h", self.handle, sends[peer].ctypes.data, src, nbytes)
                        recvs[peer] = np.empty(nbytes, dtype=np.uint8)
                self.world.sendrecv(sends, recvs)
                for peer, dst in ((peer_lo, recv_lo), (peer_hi, recv_hi)):
                    if peer >= 0:
                        call("smi_memcpy_h2d", self.handle, dst, recvs[peer].ctypes.data, nbytes)
                return 0
            except Exception as exc:
                import traceback

                traceback.print_exc()
                self._cb_error = exc
                return 1

        def group_alltoall(_user, send, recv, nbytes, first, count):
            """Equal blocks among the ranks first .. first + count - 1 (this rank among them)."""
            try:
                me = self.rank - first
                host = np.empty(count * nbytes, dtype=np.uint8)
                call("smi_memcpy_d2h", self.handle, host.ctypes.data, send, count * nbytes)
                out = np.empty(count * nbytes, dtype=np.uint8)
                out[me * nbytes : (me + 1) * nbytes] = host[me * nbytes : (me + 1) * nbytes]
                sends = {first + j: host[j * nbytes : (j + 1) * nbytes] for j in range(count) if j != me}
                recvs = {first + j: out[j * nbytes : (j + 1) * nbytes] for j in range(count) if j != me}
                self.world.sendrecv(sends, recvs)
                call("smi_memcpy_h2d", self.handle, recv, out.ctypes.data, count * nbytes)
                return 0
            except Exception as exc:
                import traceback

                traceback.print_exc()
                self._cb_error = exc
                return 1

        def p2p(_user, nsend, send_peers, sends, nrecv, recv_peers, recvs, nbytes):
            """A few equal-sized messages between device buffers; one to this rank itself is a copy."""
            try:
                out, into, own = {}, {}, None
                for i in range(nsend):
                    host = np.empty(nbytes, dtype=np.uint8)
                    call("smi_memcpy_d2h", self.handle, host.ctypes.data, sends[i], nbytes)
                    if send_peers[i] == self.rank:
                        own = host
                    else:
                        out[send_peers[i]] = host
                for j in range(nrecv):
                    if recv_peers[j] != self.rank:
                        into[recv_peers[j]] = (np.empty(nbytes, dtype=np.uint8), recvs[j])
                self.world.sendrecv(out, {peer: buf for peer, (buf, _) in into.items()})
                for j in range(nrecv):
                    host = own if recv_peers[j] == self.rank else into[recv_peers[j]][0]
                    call("smi_memcpy_h2d", self.handle, recvs[j], host.ctypes.data, nbytes)
                return 0
            except Exception as exc:
                import traceback

                traceback.print_exc()
                self._cb_error = exc
                return 1

        self._cb = _lib.EXCHANGE_CB(exchange)
        self._cb2 = _lib.ALLTOALL_CB(alltoall)
        self._cb3 = _lib.SENDRECV_CB(sendrecv)
        self._cb4 = _lib.GROUP_ALLTOALL_CB(group_alltoall)
        call("smi_set_exchange_callback", self.handle, self._cb, None)
        call("smi_set_alltoall_callback", self.handle, self._cb2)
        call("smi_set_sendrecv_callback", self.handle, self._cb3)
        call("smi_set_group_alltoall_callback", self.handle, self._cb4)
        self._cb5 = _lib.P2P_CB(p2p)
        call("smi_set_p2p_callback", self.handle, self._cb5)
        self.transport = "host"

    def allreduce(self, value, op):
        if self.nranks == 1:
            return value
        if self.transport == "rccl":
            buf = _lib.darray([value])
            call("smi_allreduce_f64", self.handle, buf, 1, {"sum": 0, "min": 1, "max": 2}[op])
            return buf[0]
        return self.world.allreduce(value, op)

    # -- memory -------------------------------------------------------------------
    def sync(self):
        call("smi_sync", self.handle)

    def zeros(self, shape, dtype=None):
        return DeviceField(self, shape, self.real_t if dtype is None else dtype)

    def field(self, ncomp=None):
        """Ghosted scalar (ncomp None) or vector field of this slab, zero-initialised."""
        shape = self.padded_shape if ncomp is None else (ncomp,) + self.padded_shape
        return self.zeros(shape)

    def from_numpy(self, arr, dtype=None):
        arr = np.ascontiguousarray(arr, dtype=dtype)
        f = DeviceField(self, arr.shape, arr.dtype, zero=False)
        f.set(arr)
        return f

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                lib.smi_ctx_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


EMPTY_SUPPORT = "empty"


class _Allocation:
    """Owns one device allocation; views keep it alive.

    ``support`` records what is known about where the contents are non-zero: ``EMPTY_SUPPORT`` (all
    zero), a ``(lo, hi)`` pair of inclusive index corners, or ``None`` (unknown).  Handing out the
    device pointer makes it unknown, because whoever receives the pointer may write anywhere; only
    the writers that state their footprint (the immersed-boundary spread, the fused flow step) ask
    for the pointer with the record kept.
    """

    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        p = C.c_void_p()
        call("smi_malloc", ctx.handle, nbytes, C.byref(p))
        self.ptr = p.value
        self.support = None
        # names of by-products of the last writer that are still true of the contents (dropped with the
        # support record whenever the raw pointer is handed out)
        self.derived = set()

    def widen_support(self, lo, hi):
        """The contents may now be non-zero inside [lo, hi] as well."""
        if self.support is None:
            return
        lo, hi = tuple(int(v) for v in lo), tuple(int(v) for v in hi)
        if self.support == EMPTY_SUPPORT:
            self.support = (lo, hi)
        else:
            old_lo, old_hi = self.support
            self.support = (tuple(min(a, b) for a, b in zip(old_lo, lo)), tuple(max(a, b) for a, b in zip(old_hi, hi)))

    def __del__(self):
        try:
            if self.ptr and self.ctx.handle:
                lib.smi_free(self.ctx.handle, C.c_void_p(self.ptr))
        except Exception:
            pass


class DeviceField:
    """C-contiguous array in HBM with numpy-flavoured host access."""

    __array_priority__ = 100

    def __init__(self, ctx, shape, dtype, zero=True, _alloc=None, _offset=0):
        self.ctx = ctx
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.size = int(np.prod(self.shape)) if self.shape else 1
        self.nbytes = self.size * self.dtype.itemsize
        self.ndim = len(self.shape)
        if _alloc is None:
            self._alloc = _Allocation(ctx, max(self.nbytes, 16))
            self._offset = 0
            if zero and self.nbytes:
                call("smi_memset_zero", ctx.handle, self.tracked_ptr(), self.nbytes)
                self._alloc.support = EMPTY_SUPPORT
        else:
            self._alloc = _alloc
            self._offset = _offset

    @property
    def ptr(self):
        """Device pointer for a caller that may write through it: the support record is dropped."""
        self._alloc.support = None
        self._alloc.derived.clear()
        return C.c_void_p(self._alloc.ptr + self._offset)

    def tracked_ptr(self):
        """Device pointer for readers, and for writers that update ``support`` themselves."""
        return C.c_void_p(self._alloc.ptr + self._offset)

    @property
    def support(self):
        return self._alloc.support

    # -- host access --------------------------------------------------------------
    def get(self):
        out = np.empty(self.shape, dtype=self.dtype)
        if self.nbytes:
            call("smi_memcpy_d2h", self.ctx.handle, out.ctypes.data, self.tracked_ptr(), self.nbytes)
        return out

    def set(self, values):
        arr = np.ascontiguousarray(np.broadcast_to(np.asarray(values, dtype=self.dtype), self.shape))
        if self.nbytes:
            call("smi_memcpy_h2d", self.ctx.handle, self.ptr, arr.ctypes.data, self.nbytes)

    def __array__(self, dtype=None, copy=None):
        a = self.get()
        return a if dtype is None else a.astype(dtype)

    def view(self):
        return DeviceField(self.ctx, self.shape, self.dtype, _alloc=self._alloc, _offset=self._offset)

    def _is_full(self, idx):
        if idx is Ellipsis:
            return True
        if isinstance(idx, slice):
            return idx == slice(None)
        if isinstance(idx, tuple):
            return all(i is Ellipsis or (isinstance(i, slice) and i == slice(None)) for i in idx)
        return False

    def __getitem__(self, idx):
        if isinstance(idx, (int, np.integer)) and self.ndim >= 2:
            # leading-axis element (a vector component): a device view, no copy
            k = int(idx) % self.shape[0]
            sub = self.shape[1:]
            stride = int(np.prod(sub)) * self.dtype.itemsize
            return DeviceField(self.ctx, sub, self.dtype, _alloc=self._alloc, _offset=self._offset + k * stride)
        if isinstance(idx, slice) and self.ndim >= 2 and idx.step in (None, 1) and not self._is_full(idx):
            start, stop, _ = idx.indices(self.shape[0])
            sub = (max(stop - start, 0),) + self.shape[1:]
            stride = int(np.prod(self.shape[1:])) * self.dtype.itemsize
            return DeviceField(self.ctx, sub, self.dtype, _alloc=self._alloc, _offset=self._offset + start * stride)
        return self.get()[idx]

    def __setitem__(self, idx, values):
        if self._is_full(idx):
            if isinstance(values, DeviceField):
                if values.shape != self.shape or values.dtype != self.dtype:
                    raise ValueError("shape / dtype mismatch in device copy")
                call("smi_memcpy_d2d", self.ctx.handle, self.ptr, values.tracked_ptr(), self.nbytes)
            else:
                self.set(values)
            return
        host = self.get()
        host[idx] = np.asarray(values)
        self.set(host)

    def exchange_storage_with(self, other):
        """Trade device storage with ``other`` (same shape and type, both whole allocations): afterwards each field,
        and every view of it, addresses the memory the other one had.  No data moves."""
        if (other.shape != self.shape or other.dtype != self.dtype or self._offset or other._offset
                or self._alloc is other._alloc):
            raise ValueError("fields must be distinct whole allocations of the same shape and type")
        a, b = self._alloc, other._alloc
        a.ptr, b.ptr = b.ptr, a.ptr
        a.support, b.support = b.support, a.support
        a.derived, b.derived = b.derived, a.derived

    def copy(self):
        out = DeviceField(self.ctx, self.shape, self.dtype, zero=False)
        call("smi_memcpy_d2d", self.ctx.handle, out.ptr, self.tracked_ptr(), self.nbytes)
        return out

    def fill(self, value):
        self.set(value)

    def __len__(self):
        return self.shape[0]

    def __repr__(self):
        return f"DeviceField(shape={self.shape}, dtype={self.dtype}, device={self.ctx.device})"


def as_ptr(x):
    """Device pointer of a DeviceField argument."""
    if isinstance(x, DeviceField):
        return x.ptr
    raise TypeError(
        f"expected a DeviceField (device-resident array), got {type(x).__name__}; "
        "allocate fields through the simulator / DeviceContext"
    )
