"""A small HDF5 writer and reader for checkpoints, used when ``h5py`` is not installed.

It covers the subset of the HDF5 file format (HDF5 File Format Specification, version 0 superblock, i.e. what
libhdf5 writes with ``libver='earliest'``) that field checkpoints need and nothing else:

* groups stored as a symbol table (one v1 B-tree node of level 0, one symbol table node, one local heap per group);
* datasets of little-endian IEEE floats or integers with a simple (fixed-size) dataspace and CONTIGUOUS storage;
* attributes (scalars and small arrays of the same types) in version 1 object headers.

Contiguous storage is what makes parallel output simple here: the writer lays out all metadata, reports the file
offset of every dataset, and each rank then writes its own hyperslab through a memory map of that region.

The reader walks the same structures generically (B-tree nodes with several children, object header
continuation blocks, attribute messages of versions 1 to 3), so it also reads files of this subset that libhdf5 wrote.
Chunked or compressed datasets, new-style (fractal heap) groups and superblock versions 2 / 3 raise
``NotImplementedError``.

The structures are written from the published specification.  They are checked by this module's own reader, by
byte-level tests against the specification's layouts, and against libhdf5 itself: tests/test_cpu_mpi_io_libhdf5.py
opens files written here with h5py 3.3 on libhdf5 1.10.6 and with ``h5dump`` (a second interpreter of the image
carries them), and reads files that h5py wrote with this reader.
"""
import struct

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF
LEAF_K = 32        # a symbol table node holds up to 2 * LEAF_K links: one node per group is enough here
INTERNAL_K = 16
HEAP_FREE_NULL = 1   # "no free block" in a local heap's free-list head

MSG_DATASPACE, MSG_DATATYPE, MSG_FILL, MSG_LAYOUT = 0x0001, 0x0003, 0x0005, 0x0008
MSG_ATTRIBUTE, MSG_CONTINUATION, MSG_SYMBOL_TABLE = 0x000C, 0x0010, 0x0011


def _pad8(data):
    return data + b"\0" * (-len(data) % 8)


# ------------------------------------------------------------------------------------------ messages
def encode_datatype(dtype):
    """Datatype message (version 1) of a little-endian IEEE float or two's-complement integer."""
    dt = np.dtype(dtype)
    if dt.byteorder == ">":
        raise ValueError("big-endian arrays are not supported")
    size = dt.itemsize
    if dt.kind == "f" and size in (4, 8):
        exp_loc, exp_bits, man_bits, bias = (23, 8, 23, 127) if size == 4 else (52, 11, 52, 1023)
        # class 1, version 1; bit field: little endian, mantissa normalisation 2 (implied leading 1), sign position
        head = struct.pack("<BBBBI", 0x11, 0x20, 8 * size - 1, 0, size)
        return head + struct.pack("<HHBBBBI", 0, 8 * size, exp_loc, exp_bits, 0, man_bits, bias)
    if dt.kind in "iu" and size in (1, 2, 4, 8):
        head = struct.pack("<BBBBI", 0x10, 0x08 if dt.kind == "i" else 0x00, 0, 0, size)
        return head + struct.pack("<HH", 0, 8 * size)
    raise ValueError(f"unsupported dtype {dt}")


def decode_datatype(data):
    cls, version = data[0] & 0x0F, data[0] >> 4
    bits0 = data[1]
    (size,) = struct.unpack_from("<I", data, 4)
    if version not in (1, 2, 3):
        raise NotImplementedError(f"datatype message version {version}")
    if bits0 & 1:
        raise NotImplementedError("big-endian datatypes")
    if cls == 1 and size in (4, 8):
        return np.dtype(f"<f{size}")
    if cls == 0 and size in (1, 2, 4, 8):
        return np.dtype(("<i" if bits0 & 0x08 else "<u") + str(size))
    raise NotImplementedError(f"datatype class {cls} of {size} bytes")


def encode_dataspace(shape):
    """Dataspace message, version 1, no maximum dimensions (rank 0: a scalar)."""
    return struct.pack("<BBBB4x", 1, len(shape), 0, 0) + b"".join(struct.pack("<Q", int(n)) for n in shape)


def decode_dataspace(data):
    version, rank = data[0], data[1]
    if version == 1:
        start = 8
    elif version == 2:
        if data[3] == 2:
            raise NotImplementedError("null dataspace")
        start = 4
    else:
        raise NotImplementedError(f"dataspace message version {version}")
    return tuple(struct.unpack_from("<Q", data, start + 8 * i)[0] for i in range(rank))


def encode_attribute(name, value):
    arr = np.asarray(value)
    if arr.dtype.kind == "f" and arr.dtype.itemsize not in (4, 8):
        arr = arr.astype(np.float64)
    if arr.dtype.kind == "b":
        arr = arr.astype(np.int8)
    raw_name = name.encode() + b"\0"
    dt, ds = encode_datatype(arr.dtype), encode_dataspace(arr.shape)
    head = struct.pack("<BBHHH", 1, 0, len(raw_name), len(dt), len(ds))
    return head + _pad8(raw_name) + _pad8(dt) + _pad8(ds) + np.ascontiguousarray(arr).tobytes()


def decode_attribute(data):
    version = data[0]
    name_size, dt_size, ds_size = struct.unpack_from("<HHH", data, 2)
    if version == 1:
        pos, step = 8, lambda n: n + (-n % 8)
    elif version in (2, 3):
        pos, step = (8 if version == 2 else 9), lambda n: n
    else:
        raise NotImplementedError(f"attribute message version {version}")
    name = data[pos:pos + name_size].split(b"\0")[0].decode()
    pos += step(name_size)
    dtype = decode_datatype(data[pos:pos + dt_size])
    pos += step(dt_size)
    shape = decode_dataspace(data[pos:pos + ds_size])
    pos += step(ds_size)
    count = int(np.prod(shape)) if shape else 1
    value = np.frombuffer(data, dtype=dtype, count=count, offset=pos).reshape(shape).copy()
    return name, (value[()] if not shape else value)


# ------------------------------------------------------------------------------------------ writer
class _Node:
    def __init__(self):
        self.attrs = {}
        self.header_address = None


class Dataset(_Node):
    def __init__(self, shape, dtype, data=None):
        super().__init__()
        self.shape = tuple(int(n) for n in shape)
        self.dtype = np.dtype(dtype).newbyteorder("<") if np.dtype(dtype).byteorder == ">" else np.dtype(dtype)
        self.data = None if data is None else np.ascontiguousarray(data, dtype=self.dtype).reshape(self.shape)
        self.data_address = None

    @property
    def nbytes(self):
        return int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize if self.shape else self.dtype.itemsize


class Group(_Node):
    def __init__(self):
        super().__init__()
        self.links = {}
        self.btree_address = self.heap_address = None

    def create_group(self, name):
        return self._add(name, Group())

    def create_dataset(self, name, shape, dtype, data=None):
        return self._add(name, Dataset(shape, dtype, data))

    def _add(self, name, node):
        if not name or "/" in name or name in self.links:
            raise ValueError(f"invalid or duplicate link name {name!r}")
        if len(self.links) >= 2 * LEAF_K:
            raise ValueError(f"a group holds at most {2 * LEAF_K} links in this writer")
        self.links[name] = node
        return node


def _message(kind, data, flags=0):
    data = _pad8(data)
    return struct.pack("<HHB3x", kind, len(data), flags) + data


def _object_header(messages):
    """Version 1 object header: 12-byte prefix, 4 bytes of alignment, then the messages."""
    body = b"".join(messages)
    return struct.pack("<BBHII4x", 1, 0, len(messages), 1, len(body)) + body


class Writer:
    """Build a tree with ``root.create_group`` / ``create_dataset`` / ``.attrs``, then ``write(path)``: metadata and
    any dataset contents given as ``data`` go to the file; the other datasets are left allocated (zero filled) at
    ``dataset.data_address`` for the caller to fill."""

    def __init__(self):
        self.root = Group()

    def write(self, path):
        order = []          # every node, parents before children

        def walk(node):
            order.append(node)
            if isinstance(node, Group):
                for name in sorted(node.links):
                    walk(node.links[name])

        walk(self.root)
        # pass 1: sizes -> addresses (a header's size does not depend on the addresses it holds)
        pos = 96            # superblock, version 0, 8-byte offsets and lengths
        for node in order:
            node.header_address = pos
            pos += len(self._header(node))
            if isinstance(node, Group):
                node.btree_address = pos
                pos += 24 + 8 * (2 * INTERNAL_K + 1) + 8 * 2 * INTERNAL_K
                node.heap_address = pos
                pos += 32 + len(self._heap_data(node)[0])
                node.snod_address = pos if node.links else None
                pos += (8 + 40 * 2 * LEAF_K) if node.links else 0
        for node in order:
            if isinstance(node, Dataset):
                pos += -pos % 8
                node.data_address = pos
                pos += node.nbytes
        end_of_file = pos
        # pass 2: bytes
        with open(path, "wb") as f:
            f.write(self._superblock(end_of_file))
            for node in order:
                f.seek(node.header_address)
                f.write(self._header(node))
                if isinstance(node, Group):
                    heap, offsets = self._heap_data(node)
                    f.seek(node.btree_address)
                    f.write(self._btree(node, offsets))
                    f.seek(node.heap_address)
                    used = len(heap) - 32
                    f.write(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), used, node.heap_address + 32) + heap)
                    if node.links:
                        f.seek(node.snod_address)
                        f.write(self._snod(node, offsets))
            for node in order:
                if isinstance(node, Dataset) and node.data is not None:
                    f.seek(node.data_address)
                    f.write(node.data.tobytes())
            f.truncate(end_of_file)
        return end_of_file

    # -- pieces ---------------------------------------------------------------------------------
    def _superblock(self, end_of_file):
        root = self.root
        head = SIGNATURE + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, LEAF_K, INTERNAL_K, 0)
        addresses = struct.pack("<QQQQ", 0, UNDEF, end_of_file, UNDEF)
        entry = struct.pack("<QQII", 0, root.header_address, 1, 0) + struct.pack("<QQ", root.btree_address or 0,
                                                                                 root.heap_address or 0)
        out = head + addresses + entry
        assert len(out) == 96
        return out

    def _header(self, node):
        msgs = []
        if isinstance(node, Group):
            msgs.append(_message(MSG_SYMBOL_TABLE, struct.pack("<QQ", node.btree_address or 0, node.heap_address or 0)))
        else:
            msgs.append(_message(MSG_DATASPACE, encode_dataspace(node.shape)))
            msgs.append(_message(MSG_DATATYPE, encode_datatype(node.dtype), flags=1))
            # fill value, version 2: allocate late, write the fill value if the user set one, default fill value
            msgs.append(_message(MSG_FILL, struct.pack("<BBBBI", 2, 2, 2, 1, 0)))
            msgs.append(_message(MSG_LAYOUT, struct.pack("<BBQQ", 3, 1, node.data_address or 0, node.nbytes)))
        for name in node.attrs:
            msgs.append(_message(MSG_ATTRIBUTE, encode_attribute(name, node.attrs[name])))
        return _object_header(msgs)

    @staticmethod
    def _heap_data(group):
        """Data segment of the group's local heap: the empty name at offset 0, the link names (null terminated, 8-byte
        aligned), and one free block at the end.  Returns (segment, {name: offset})."""
        data, offsets = bytearray(8), {}
        for name in sorted(group.links):
            offsets[name] = len(data)
            data += _pad8(name.encode() + b"\0")
        data += struct.pack("<QQ", HEAP_FREE_NULL, 32) + bytes(16)   # free block: no successor, 32 bytes long
        return bytes(data), offsets

    @staticmethod
    def _btree(group, offsets):
        names = sorted(group.links)
        size = 24 + 8 * (2 * INTERNAL_K + 1) + 8 * 2 * INTERNAL_K
        out = bytearray(b"TREE" + struct.pack("<BBHQQ", 0, 0, 1 if names else 0, UNDEF, UNDEF))
        if names:
            # key 0: the empty name; child 0: the symbol table node; key 1: the largest name in that node
            out += struct.pack("<QQQ", 0, group.snod_address, offsets[names[-1]])
        return bytes(out) + bytes(size - len(out))

    @staticmethod
    def _snod(group, offsets):
        names = sorted(group.links)     # byte order of the names, as strcmp sorts them
        out = bytearray(b"SNOD" + struct.pack("<BBH", 1, 0, len(names)))
        for name in names:
            node = group.links[name]
            if isinstance(node, Group):
                out += struct.pack("<QQII", offsets[name], node.header_address, 1, 0)
                out += struct.pack("<QQ", node.btree_address, node.heap_address)
            else:
                out += struct.pack("<QQII", offsets[name], node.header_address, 0, 0) + bytes(16)
        return bytes(out) + bytes(8 + 40 * 2 * LEAF_K - len(out))


# ------------------------------------------------------------------------------------------ reader
class DatasetInfo:
    def __init__(self, path, shape, dtype, address, attrs):
        self.path, self.shape, self.dtype, self.address, self.attrs = path, shape, dtype, address, attrs

    def memmap(self, mode="r"):
        """numpy view of the dataset's bytes in the file (contiguous storage)."""
        if not self.shape:
            return np.memmap(self.path, dtype=self.dtype, mode=mode, offset=self.address, shape=(1,))
        return np.memmap(self.path, dtype=self.dtype, mode=mode, offset=self.address, shape=self.shape)

    def __getitem__(self, index):
        m = self.memmap()
        return np.array(m[index] if self.shape else m[0])


class GroupInfo:
    def __init__(self, attrs, links):
        self.attrs, self.links = attrs, links


class Reader:
    """Read-only view of a file of the supported subset: ``reader[path]`` gives a DatasetInfo or GroupInfo,
    ``reader.attrs`` the root attributes, ``reader.visit()`` every path."""

    def __init__(self, path):
        self.path = path
        with open(path, "rb") as f:
            self._buf = f.read(1 << 20)      # metadata of a checkpoint sits at the head of the file
            self._file_size = f.seek(0, 2)
        if self._buf[:8] != SIGNATURE:
            raise ValueError(f"{path}: not an HDF5 file")
        version = self._buf[8]
        if version not in (0, 1):
            raise NotImplementedError(f"{path}: superblock version {version} (only 0 and 1 are supported)")
        if self._buf[13] != 8 or self._buf[14] != 8:
            raise NotImplementedError("only 8-byte offsets and lengths are supported")
        self.leaf_k, self.internal_k = struct.unpack_from("<HH", self._buf, 16)
        pos = 24 + (4 if version == 1 else 0)
        self.base, _free, self.end_of_file, _driver = struct.unpack_from("<QQQQ", self._buf, pos)
        _name, root_header, _cache, _res = struct.unpack_from("<QQII", self._buf, pos + 32)
        self.root = self._object("", root_header)
        self.attrs = self.root.attrs

    def _read(self, address, size):
        address += self.base
        if address + size <= len(self._buf):
            return self._buf[address:address + size]
        with open(self.path, "rb") as f:
            f.seek(address)
            return f.read(size)

    def _messages(self, address):
        version = self._read(address, 1)[0]
        if version != 1:
            raise NotImplementedError(f"object header version {version} (only version 1 is supported)")
        _v, _r, count, _refs, size = struct.unpack("<BBHII", self._read(address, 12))
        blocks, out = [(address + 16, size)], []
        while blocks and len(out) < count:
            start, length = blocks.pop(0)
            data, pos = self._read(start, length), 0
            while pos + 8 <= length and len(out) < count:
                kind, msize, _flags = struct.unpack_from("<HHB", data, pos)
                body = data[pos + 8:pos + 8 + msize]
                if kind == MSG_CONTINUATION:
                    blocks.append(struct.unpack("<QQ", body[:16]))
                out.append((kind, body))
                pos += 8 + msize
        return out

    def _object(self, path, address):
        attrs, shape, dtype, data_address, table = {}, None, None, None, None
        for kind, body in self._messages(address):
            if kind == MSG_ATTRIBUTE:
                name, value = decode_attribute(body)
                attrs[name] = value
            elif kind == MSG_DATASPACE:
                shape = decode_dataspace(body)
            elif kind == MSG_DATATYPE:
                dtype = decode_datatype(body)
            elif kind == MSG_LAYOUT:
                if body[0] != 3:
                    raise NotImplementedError(f"data layout message version {body[0]}")
                if body[1] != 1:
                    raise NotImplementedError("only contiguous dataset storage is supported")
                data_address = struct.unpack_from("<Q", body, 2)[0] + self.base
            elif kind == MSG_SYMBOL_TABLE:
                table = struct.unpack("<QQ", body[:16])
        if table is not None:
            links = {}
            heap_data = self._heap(table[1])
            for name_offset, header in self._btree_entries(table[0]):
                name = heap_data[name_offset:heap_data.index(b"\0", name_offset)].decode()
                links[name] = self._object(f"{path}/{name}" if path else name, header)
            return GroupInfo(attrs, links)
        if shape is None or dtype is None or data_address is None:
            raise ValueError(f"object at {address} is neither a group nor a dataset of the supported kind")
        return DatasetInfo(self.path, shape, dtype, data_address, attrs)

    def _heap(self, address):
        head = self._read(address, 32)
        if head[:4] != b"HEAP":
            raise ValueError("bad local heap signature")
        size, _free, data_address = struct.unpack_from("<QQQ", head, 8)
        return self._read(data_address, size)

    def _btree_entries(self, address):
        head = self._read(address, 24)
        if head[:4] != b"TREE":
            raise ValueError("bad B-tree signature")
        _type, level, used = struct.unpack_from("<BBH", head, 4)
        body = self._read(address + 24, 8 * (2 * used + 1))
        children = [struct.unpack_from("<Q", body, 8 * (2 * i + 1))[0] for i in range(used)]
        out = []
        for child in children:
            if level > 0:
                out += self._btree_entries(child)
                continue
            node = self._read(child, 8)
            if node[:4] != b"SNOD":
                raise ValueError("bad symbol table node signature")
            (count,) = struct.unpack_from("<H", node, 6)
            entries = self._read(child + 8, 40 * count)
            out += [struct.unpack_from("<QQ", entries, 40 * i) for i in range(count)]
        return out

    # -- access ---------------------------------------------------------------------------------
    def __getitem__(self, path):
        node = self.root
        for part in [p for p in path.split("/") if p]:
            if not isinstance(node, GroupInfo) or part not in node.links:
                raise KeyError(path)
            node = node.links[part]
        return node

    def __contains__(self, path):
        try:
            self[path]
            return True
        except KeyError:
            return False

    def visit(self):
        out = []

        def walk(prefix, group):
            for name in sorted(group.links):
                full = f"{prefix}/{name}" if prefix else name
                out.append(full)
                if isinstance(group.links[name], GroupInfo):
                    walk(full, group.links[name])

        walk("", self.root)
        return out
