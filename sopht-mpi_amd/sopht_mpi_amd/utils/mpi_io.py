"""Checkpoint / visualisation output with the reference's ``MPIIO`` interface (sopht_mpi/utils/mpi_io.py:8-554):
an HDF5 file with the same hierarchy plus XDMF descriptions for Paraview, and ``load`` for restarts.

                /                   attrs: time
                  Eulerian/Scalar/<field>            (1, [nz,] ny, nx)
                  Eulerian/Vector/<field>_<c>        one scalar dataset per component
                  Eulerian/Parameters                attrs: origin, dx, grid_size
                  Lagrangian/<grid>/Grid             (N, dim)
                  Lagrangian/<grid>/Connection       (N,) int64, optional
                  Lagrangian/<grid>/Scalar/<field>   (N,)
                  Lagrangian/<grid>/Vector/<field>   (N, dim)

Fields may be ``DeviceField`` objects (fetched from / uploaded to HBM around the file access) or numpy arrays.
The file is written without an HDF5 library (``h5lite``): rank 0 lays out the metadata, every dataset is stored
contiguously, and each rank writes the block of the global array it owns through a memory map - the parallel
hyperslab write of the reference's ``mpio`` driver without MPI-IO.  ``load`` reads the same way; a file that
``h5lite`` cannot parse (written by other tools with chunking, say) is read through ``h5py`` when that is installed.
"""
import os
import xml.etree.ElementTree as ET

import numpy as np

from ..device import DeviceField
from . import h5lite


def _host(field):
    return field.get() if isinstance(field, DeviceField) else np.asarray(field)


class MPIIO:
    def __init__(self, mpi_construct, real_dtype=np.float64):
        self.mpi_construct = mpi_construct
        self.dim = mpi_construct.grid_dim
        if self.dim not in (2, 3):
            raise AssertionError("Invalid dimension (only 2D and 3D)")
        self.real_dtype = real_dtype
        self.precision = 8 if np.dtype(real_dtype) == np.float32 else 16   # digits in the XDMF text, as :42
        self.eulerian_grid_defined = False
        self.eulerian_fields, self.eulerian_fields_type = {}, {}
        self.lagrangian_fields, self.lagrangian_fields_type = {}, {}
        self.lagrangian_grids, self.lagrangian_fields_with_grid_name = {}, {}
        self.lagrangian_grid_count = 0
        self.lagrangian_grid_connection, self.lagrangian_grid_master_rank, self.lagrangian_grid_num_node = {}, {}, {}

    # ------------------------------------------------------------------ registration
    def define_eulerian_grid(self, origin, dx, grid_size, ghost_size):
        """:60-120.  origin, dx, grid_size: (dim,) arrays in z-y-x order."""
        for arr in (origin, dx, grid_size):
            if not isinstance(arr, np.ndarray):
                raise AssertionError("origin, dx and grid_size must be numpy arrays")
        if ghost_size < 0 or not isinstance(ghost_size, (int, np.integer)):
            raise ValueError(f"Ghost size {ghost_size} needs to be an integer >= 0 for eulerian field IO.")
        self.eulerian_origin, self.eulerian_dx, self.eulerian_grid_size = origin, dx, grid_size
        self.eulerian_grid_defined = True
        coords = np.asarray(self.mpi_construct.grid.coords)[: self.dim]
        local = np.asarray(self.mpi_construct.local_grid_size)
        # this rank's block of the global (1, [z,] y, x) dataset
        self.local_eulerian_index = (slice(None),) + tuple(
            slice(int(c * n), int((c + 1) * n)) for c, n in zip(coords, local))
        self.ghost_size = int(ghost_size)
        self.local_eulerian_grid_size = local
        self.local_eulerian_grid_size_with_ghost = local + 2 * self.ghost_size
        g = self.ghost_size
        self.eulerian_field_inner_index = (slice(g, -g),) * self.dim if g else (slice(None),) * self.dim

    def add_as_eulerian_fields_for_io(self, **fields_for_io):
        """:122-155: keyword name -> local (ghosted) scalar or vector field."""
        if not self.eulerian_grid_defined:
            raise AssertionError("Eulerian mesh is not defined!")
        padded = tuple(int(n) for n in self.local_eulerian_grid_size_with_ghost)
        for name, field in fields_for_io.items():
            if not np.issubdtype(field.dtype, self.real_dtype):
                raise AssertionError(f"{name} dtype ({field.dtype}) incompatible with IO dtype ({self.real_dtype})")
            if tuple(field.shape) == padded:
                kind = "Scalar"
            elif tuple(field.shape) == (self.dim,) + padded:
                kind = "Vector"
            else:
                raise ValueError(
                    f"Unable to identify eulerian field type (scalar / vector) based on field dimension {field.shape}")
            self.eulerian_fields[name], self.eulerian_fields_type[name] = field, kind

    def add_as_lagrangian_fields_for_io(self, lagrangian_grid_master_rank, lagrangian_grid, lagrangian_grid_name=None,
                                        lagrangian_grid_connect=False, **fields_for_io):
        """:157-258: a (dim, N) grid of nodes held by one rank, and the fields that live on it."""
        if len(lagrangian_grid.shape) != 2:
            raise AssertionError("lagrangian grid has to be a 2D (dim, N) array.")
        if lagrangian_grid.shape[0] != self.dim:
            raise AssertionError("Invalid lagrangian grid dimension (only 2D and 3D)")
        if lagrangian_grid_name is None:
            lagrangian_grid_name = f"Lagrangian_grid_{self.lagrangian_grid_count}"
            self.lagrangian_grid_count += 1
        bcast = self.mpi_construct.grid.bcast
        master = lagrangian_grid_master_rank
        num_nodes = int(bcast(int(lagrangian_grid.shape[1]), root=master))
        self.lagrangian_grid_num_node[lagrangian_grid_name] = num_nodes
        self.lagrangian_grid_master_rank[lagrangian_grid_name] = master
        if lagrangian_grid_connect:
            self.lagrangian_grid_connection[lagrangian_grid_name] = np.arange(num_nodes, dtype=np.int64)
        if not np.issubdtype(lagrangian_grid.dtype, self.real_dtype):
            raise AssertionError(f"{lagrangian_grid_name} dtype ({lagrangian_grid.dtype}) incompatible with IO dtype "
                                 f"({self.real_dtype})")
        self.lagrangian_grids[lagrangian_grid_name] = lagrangian_grid
        names = self.lagrangian_fields_with_grid_name[lagrangian_grid_name] = []
        for name, field in fields_for_io.items():
            if not np.issubdtype(field.dtype, self.real_dtype):
                raise AssertionError(f"{name} dtype ({field.dtype}) incompatible with IO dtype ({self.real_dtype})")
            self.lagrangian_fields[name] = field
            names.append(name)
            kind = None
            if self.mpi_construct.rank == master:     # only the master rank holds the data
                if field.shape[0] == lagrangian_grid.shape[1]:
                    kind = "Scalar"
                elif tuple(field.shape) == tuple(lagrangian_grid.shape):
                    kind = "Vector"
                else:
                    raise ValueError(
                        f"Unable to identify lagrangian field type (scalar / vector) based on field dimension "
                        f"{field.shape}")
            self.lagrangian_fields_type[name] = bcast(kind, root=master)

    # ------------------------------------------------------------------ save
    def save(self, h5_file_name, time=0.0):
        self._save(h5_file_name, time)

    def _tree(self, time):
        """Metadata of the file (rank 0): groups, empty datasets, attributes."""
        real = np.dtype(self.real_dtype)
        writer = h5lite.Writer()
        writer.root.attrs["time"] = np.float64(time)
        datasets = {}
        if self.eulerian_grid_defined and self.eulerian_fields:
            eul = writer.root.create_group("Eulerian")
            groups = {"Scalar": eul.create_group("Scalar"), "Vector": eul.create_group("Vector")}
            # one leading axis of length 1: a 2D field then renders as a slice of a 3D space (the reference's
            # workaround for Paraview's 2DCoRectMesh orientation, :306-316)
            shape = (1,) + tuple(int(n) for n in self.eulerian_grid_size)
            for name, kind in self.eulerian_fields_type.items():
                parts = [name] if kind == "Scalar" else [f"{name}_{c}" for c in range(self.dim)]
                for part in parts:
                    datasets[f"Eulerian/{kind}/{part}"] = groups[kind].create_dataset(part, shape, real)
            params = eul.create_group("Parameters")
            params.attrs["origin"] = np.asarray(self.eulerian_origin)
            params.attrs["dx"] = np.asarray(self.eulerian_dx)
            params.attrs["grid_size"] = np.asarray(self.eulerian_grid_size)
        lag = writer.root.create_group("Lagrangian")
        for grid_name in self.lagrangian_grids:
            n = self.lagrangian_grid_num_node[grid_name]
            grp = lag.create_group(grid_name)
            datasets[f"Lagrangian/{grid_name}/Grid"] = grp.create_dataset("Grid", (n, self.dim), real)
            if grid_name in self.lagrangian_grid_connection:
                datasets[f"Lagrangian/{grid_name}/Connection"] = grp.create_dataset("Connection", (n,), np.int64)
            sub = {"Scalar": grp.create_group("Scalar"), "Vector": grp.create_group("Vector")}
            for name in self.lagrangian_fields_with_grid_name[grid_name]:
                kind = self.lagrangian_fields_type[name]
                shape = (n,) if kind == "Scalar" else (n, self.dim)
                datasets[f"Lagrangian/{grid_name}/{kind}/{name}"] = sub[kind].create_dataset(name, shape, real)
        return writer, datasets

    def _save(self, h5_file_name, time=0.0):
        """:271-441.  Rank 0 writes the metadata; every rank then fills the blocks it owns."""
        world, rank = self.mpi_construct.world, self.mpi_construct.rank
        layout = None
        if rank == 0:
            writer, datasets = self._tree(time)
            writer.write(h5_file_name)
            layout = {path: [int(d.data_address), list(d.shape), d.dtype.str] for path, d in datasets.items()}
        layout = world.bcast_object(layout, root=0)      # also orders the other ranks behind the file's creation

        def region(path):
            address, shape, dtype = layout[path]
            return np.memmap(h5_file_name, dtype=np.dtype(dtype), mode="r+", offset=address, shape=tuple(shape))

        if self.eulerian_grid_defined and self.eulerian_fields:
            inner = self.eulerian_field_inner_index
            for name, field in self.eulerian_fields.items():
                kind = self.eulerian_fields_type[name]
                host = _host(field)
                if kind == "Scalar":
                    parts = [(name, host)]
                else:
                    parts = [(f"{name}_{c}", host[c]) for c in range(self.dim)]
                for part, values in parts:
                    out = region(f"Eulerian/{kind}/{part}")
                    out[self.local_eulerian_index] = values[inner][None]
                    out.flush()
            if rank == 0:
                self.generate_xdmf_eulerian(h5_file_name=h5_file_name, time=time)
        for grid_name, grid in self.lagrangian_grids.items():
            if rank != self.lagrangian_grid_master_rank[grid_name]:
                continue
            out = region(f"Lagrangian/{grid_name}/Grid")
            out[...] = np.transpose(_host(grid))     # (dim, N) -> (N, dim), what Paraview expects
            out.flush()
            if grid_name in self.lagrangian_grid_connection:
                out = region(f"Lagrangian/{grid_name}/Connection")
                out[...] = self.lagrangian_grid_connection[grid_name]
                out.flush()
            for name in self.lagrangian_fields_with_grid_name[grid_name]:
                kind = self.lagrangian_fields_type[name]
                out = region(f"Lagrangian/{grid_name}/{kind}/{name}")
                values = _host(self.lagrangian_fields[name])
                out[...] = values if kind == "Scalar" else np.moveaxis(values, 0, -1)
                out.flush()
            self.generate_xdmf_lagrangian(h5_file_name=h5_file_name, time=time)
        world.barrier()

    # ------------------------------------------------------------------ load
    @staticmethod
    def _open(h5_file_name):
        """(dataset lookup, root attributes, set of paths).  h5lite first; h5py for files outside its subset."""
        try:
            reader = h5lite.Reader(h5_file_name)
            return (lambda path: reader[path]), reader.attrs, set(reader.visit()), (lambda path: reader[path].attrs)
        except NotImplementedError:
            try:
                import h5py
            except ImportError:
                raise
            f = h5py.File(h5_file_name, "r")
            keys = []
            f.visit(keys.append)
            return (lambda path: f[path]), f.attrs, set(keys), (lambda path: f[path].attrs)

    def load(self, h5_file_name):
        """:443-554: fills the registered fields from the file and returns its time stamp."""
        dataset, root_attrs, keys, attrs_of = self._open(h5_file_name)
        time = float(root_attrs["time"])
        if self.eulerian_fields:
            if not self.eulerian_grid_defined:
                raise AssertionError("Eulerian grid is not defined!")
            inner = self.eulerian_field_inner_index
            for name, field in self.eulerian_fields.items():
                kind = self.eulerian_fields_type[name]
                parts = [name] if kind == "Scalar" else [f"{name}_{c}" for c in range(self.dim)]
                host = _host(field)
                for c, part in enumerate(parts):
                    if f"Eulerian/{kind}/{part}" not in keys:
                        raise AssertionError(f"Unable to find {kind.lower()} field {part} in loaded file!")
                    block = np.asarray(dataset(f"Eulerian/{kind}/{part}")[self.local_eulerian_index])[0]
                    (host if kind == "Scalar" else host[c])[inner] = block
                if isinstance(field, DeviceField):
                    field.set(host)
            params = attrs_of("Eulerian/Parameters")
            np.testing.assert_allclose(self.eulerian_origin, params["origin"])
            np.testing.assert_allclose(self.eulerian_dx, params["dx"])
            np.testing.assert_allclose(self.eulerian_grid_size, params["grid_size"])
        if self.lagrangian_fields or self.lagrangian_grids:
            for grid_name, grid in self.lagrangian_grids.items():
                if self.mpi_construct.rank != self.lagrangian_grid_master_rank[grid_name]:
                    continue
                if f"Lagrangian/{grid_name}/Grid" not in keys:
                    raise AssertionError(f"Unable to find grid '{grid_name}' in loaded file!")
                self._assign(grid, np.transpose(np.asarray(dataset(f"Lagrangian/{grid_name}/Grid")[...])))
                if f"Lagrangian/{grid_name}/Connection" in keys:
                    self.lagrangian_grid_connection[grid_name] = np.asarray(
                        dataset(f"Lagrangian/{grid_name}/Connection")[...])
                for name in self.lagrangian_fields_with_grid_name[grid_name]:
                    kind = self.lagrangian_fields_type[name]
                    path = f"Lagrangian/{grid_name}/{kind}/{name}"
                    if path not in keys:
                        raise AssertionError(
                            f"Unable to find {kind.lower()} field {name} on grid {grid_name} in loaded file!")
                    values = np.asarray(dataset(path)[...])
                    self._assign(self.lagrangian_fields[name], values if kind == "Scalar" else np.moveaxis(values, -1, 0))
        self.mpi_construct.world.barrier()
        return time

    @staticmethod
    def _assign(field, values):
        if isinstance(field, DeviceField):
            field.set(values)
        else:
            field[...] = values

    # ------------------------------------------------------------------ XDMF
    def _numbers(self, values):
        return " ".join(np.format_float_positional(v, precision=self.precision, trim="-") if isinstance(
            v, (float, np.floating)) else str(int(v)) for v in np.asarray(values).tolist())

    def _data_item(self, parent, dims, text, fmt="HDF", number_type="Float", name=None):
        item = ET.SubElement(parent, "DataItem", Dimensions=dims, NumberType=number_type,
                             Precision=str(self.precision), Format=fmt)
        if name:
            item.set("Name", name)
        item.text = text
        return item

    @staticmethod
    def _write_xdmf(root, path):
        ET.indent(root, space="    ")
        with open(path, "w") as f:
            f.write('<?xml version="1.0" ?>\n<!DOCTYPE Xdmf SYSTEM "Xdmf.dtd" []>\n')
            f.write(ET.tostring(root, encoding="unicode") + "\n")

    def _xdmf_grid(self, time):
        root = ET.Element("Xdmf", {"xmlns:xi": "http://www.w3.org/2003/XInclude", "Version": "2.2"})
        grid = ET.SubElement(ET.SubElement(root, "Domain"), "Grid", GridType="Uniform")
        ET.SubElement(grid, "Time", Value=str(time))
        return root, grid

    def generate_xdmf_eulerian(self, h5_file_name, time=0.0):
        """``<name>_eulerian.xmf``: a 3DCoRectMesh (a 2D grid is a one-cell-thick slab, as the datasets are) with one
        scalar node attribute per dataset (:556-645)."""
        pad = self.dim == 2
        size = np.insert(self.eulerian_grid_size, 0, 1) if pad else np.asarray(self.eulerian_grid_size)
        origin = np.insert(self.eulerian_origin, 0, 0.0) if pad else np.asarray(self.eulerian_origin)
        spacing = np.insert(self.eulerian_dx, 0, 0.0) if pad else np.asarray(self.eulerian_dx)
        dims = self._numbers(size.astype(np.int64))
        root, grid = self._xdmf_grid(time)
        ET.SubElement(grid, "Topology", TopologyType="3DCORECTMesh", Dimensions=dims)
        geometry = ET.SubElement(grid, "Geometry", GeometryType="ORIGIN_DXDYDZ")
        self._data_item(geometry, str(self.dim), self._numbers(origin.astype(np.float64)), fmt="XML", name="Origin")
        self._data_item(geometry, str(self.dim), self._numbers(spacing.astype(np.float64)), fmt="XML", name="Spacing")
        for name, kind in self.eulerian_fields_type.items():
            for part in ([name] if kind == "Scalar" else [f"{name}_{c}" for c in range(self.dim)]):
                attr = ET.SubElement(grid, "Attribute", Name=part, Active="1", AttributeType="Scalar", Center="Node")
                self._data_item(attr, dims, f"{h5_file_name}:/Eulerian/{kind}/{part}")
        self._write_xdmf(root, h5_file_name.replace(".h5", "_eulerian.xmf"))

    def generate_xdmf_lagrangian(self, h5_file_name, time):
        """``<name>_<grid>.xmf`` per Lagrangian grid: a Polyvertex (or Polyline, with a connection) topology over the
        (N, dim) node array and its fields (:647-760)."""
        for grid_name in self.lagrangian_grids:
            if self.mpi_construct.rank != self.lagrangian_grid_master_rank[grid_name]:
                continue
            n = self.lagrangian_grid_num_node[grid_name]
            root, grid = self._xdmf_grid(time)
            if grid_name in self.lagrangian_grid_connection:
                topo = ET.SubElement(grid, "Topology", TopologyType="Polyline")
                item = ET.SubElement(topo, "DataItem", DataType="Int", Dimensions=f"1 {n}", Format="HDF",
                                     Precision=str(self.precision))
                item.text = f"{h5_file_name}:/Lagrangian/{grid_name}/Connection"
            else:
                ET.SubElement(grid, "Topology", TopologyType="Polyvertex", NumberOfElements=str(n))
            geometry = ET.SubElement(grid, "Geometry", GeometryType="XYZ" if self.dim == 3 else "XY")
            self._data_item(geometry, f"{n} {self.dim}", f"{h5_file_name}:/Lagrangian/{grid_name}/Grid")
            for name in self.lagrangian_fields_with_grid_name[grid_name]:
                kind = self.lagrangian_fields_type[name]
                attr = ET.SubElement(grid, "Attribute", Name=name, Active="1", AttributeType=kind, Center="Node")
                self._data_item(attr, str(n) if kind == "Scalar" else f"{n} {self.dim}",
                                f"{h5_file_name}:/Lagrangian/{grid_name}/{kind}/{name}")
            self._write_xdmf(root, h5_file_name.replace(".h5", f"_{grid_name}.xmf"))


class CosseratRodMPIIO(MPIIO):
    """:763-805: a rod's element centres (and radii) as a connected Lagrangian grid named ``rod``.  ``cosserat_rod``
    is any object with ``position_collection`` (3, n_elems + 1), ``radius`` (n_elems,) and ``n_elems``."""

    def __init__(self, mpi_construct, cosserat_rod, master_rank=0):
        self.cosserat_rod = cosserat_rod
        real_dtype = cosserat_rod.position_collection.dtype.type
        super().__init__(mpi_construct, real_dtype)
        self.rod_element_position = np.zeros((self.dim, cosserat_rod.n_elems), dtype=real_dtype)
        self._update_rod_element_position()
        self.add_as_lagrangian_fields_for_io(
            lagrangian_grid=self.rod_element_position, lagrangian_grid_master_rank=master_rank,
            lagrangian_grid_name="rod", scalar_3d=self.cosserat_rod.radius, lagrangian_grid_connect=True)

    def save(self, h5_file_name, time=0.0):
        self._update_rod_element_position()
        self._save(h5_file_name=h5_file_name, time=time)

    def _update_rod_element_position(self):
        nodes = self.cosserat_rod.position_collection
        self.rod_element_position[...] = 0.5 * (nodes[: self.dim, 1:] + nodes[: self.dim, :-1])


__all__ = ["MPIIO", "CosseratRodMPIIO"]
