"""Field output with the call shape of ``dolfinx.io.VTXWriter`` (reference
demo/taylor_green.py:183-184,211-215: ``VTXWriter(comm, "u.bp", [solver.u], engine="BP4")``,
``.write(t)``, ``.close()``).

ADIOS2/BP4 is not available here; the writer produces the open VTK XML format instead: one
``<stem>_<step>.vtu`` (UnstructuredGrid, base64 "binary" arrays, float64) per ``write(t)`` and a
``<stem>.pvd`` collection that maps the files to their time values -- ParaView opens the ``.pvd``
as the time series it would get from the ``.bp`` directory.  P2 fields are written on quadratic
cells (VTK_QUADRATIC_TRIANGLE = 22, VTK_QUADRATIC_TETRA = 24) with the P2 nodes as points, P3 fields on triangles on
VTK_LAGRANGE_TRIANGLE = 69 cells with the (gll_warped) P3 nodes as points, so no interpolation to P1 happens, as with VTX.  In mesh-partitioned runs every rank writes the cells
of its own partition as a separate piece (``<stem>_p<rank>_<step>.vtu``), all listed in the .pvd
written by rank 0.

Host-side Python only: output is not on the per-step hot path.
"""
from __future__ import annotations

import base64
import os

import numpy as np

from .fem import Function, VectorFunctionSpace

# my local dof order (vertices, then fem.local_edges) -> VTK node order
_VTK_PERM = {(2, 1): [0, 1, 2], (3, 1): [0, 1, 2, 3],
             (2, 2): [0, 1, 2, 5, 3, 4],  # e01, e12, e20
             (3, 2): [0, 1, 2, 3, 9, 6, 8, 7, 5, 4],  # e01, e12, e02, e03, e13, e23
             # P3 triangle -> VTK_LAGRANGE_TRIANGLE: e01 (from 0 to 1), e12 (1 to 2), e20 (2 to 0), interior
             (2, 3): [0, 1, 2, 7, 8, 3, 4, 6, 5, 9],
             # P3 tetrahedron -> VTK_LAGRANGE_TETRAHEDRON: e01, e12, e20 (from 2 to 0), e03, e13, e23, then the faces
             # (0,1,3), (1,2,3), (0,2,3), (0,1,2) -- DOLFINx's own permutation for this element (io/cells.cpp)
             (3, 3): [0, 1, 2, 3, 14, 15, 8, 9, 13, 12, 10, 11, 6, 7, 4, 5, 18, 16, 17, 19]}
_VTK_TYPE = {(2, 1): 5, (3, 1): 10, (2, 2): 22, (3, 2): 24, (2, 3): 69, (3, 3): 71}


def _b64(a: np.ndarray) -> str:
    raw = np.ascontiguousarray(a).tobytes()
    return base64.b64encode(np.uint64(len(raw)).tobytes() + raw).decode("ascii")


def _data_array(name: str, a: np.ndarray, ncomp: int = 1) -> str:
    vt = {"float64": "Float64", "int64": "Int64", "uint8": "UInt8"}[str(a.dtype)]
    nc = f' NumberOfComponents="{ncomp}"' if ncomp > 1 else ""
    return f'<DataArray type="{vt}" Name="{name}"{nc} format="binary">{_b64(a)}</DataArray>\n'


class VTXWriter:
    """``VTXWriter(comm, filename, functions, engine="BP4")``: all functions must live on the same
    scalar space (or its blocked vector space), as DOLFINx requires."""

    def __init__(self, comm, filename, output, engine: str = "BP4", mesh_policy=None):
        fns = [output] if isinstance(output, Function) else list(output)
        if not fns:
            raise ValueError("VTXWriter: no functions to write")
        spaces = {id(self._scalar_space(f)) for f in fns}
        if len(spaces) != 1:
            raise RuntimeError("VTXWriter: all functions must share one element/space")
        self._fns = fns
        self._V = self._scalar_space(fns[0])
        self._comm = comm
        stem, _ = os.path.splitext(str(filename))  # "u.bp" -> "u"
        self._stem = stem
        self._dir = os.path.dirname(stem) or "."
        os.makedirs(self._dir, exist_ok=True)
        self._rank = getattr(comm, "rank", 0) if comm is not None else 0
        self._size = getattr(comm, "size", 1) if comm is not None else 1
        self._times, self._files = [], []
        self._topology = None
        self._closed = False

    @staticmethod
    def _scalar_space(f: Function):
        V = f.function_space
        return V.scalar if isinstance(V, VectorFunctionSpace) else V

    # ---- static part: points and cells ---------------------------------------------------
    def _build_topology(self):
        V = self._V
        d, deg = V.mesh.gdim, V.degree
        X = np.zeros((V.num_dofs, 3))
        X[:, :d] = V.tabulate_dof_coordinates()[:, :d]
        cd = V.cell_dofs.cpu().numpy().astype(np.int64)
        if V.part is not None:  # this rank's own cells only (the ghost layer belongs to the neighbours)
            own = (V.part.cell_rank[V.local_cells] == V.part.rank).cpu().numpy()
            cd = cd[own]
        conn = cd[:, _VTK_PERM[(d, deg)]]
        nper = conn.shape[1]
        offsets = (np.arange(conn.shape[0], dtype=np.int64) + 1) * nper
        types = np.full(conn.shape[0], _VTK_TYPE[(d, deg)], dtype=np.uint8)
        self._topology = (X, conn.reshape(-1), offsets, types)

    def _piece_name(self, step: int) -> str:
        part = f"_p{self._rank}" if self._size > 1 else ""
        return f"{self._stem}{part}_{step:06d}.vtu"

    def write(self, t: float):
        if self._closed:
            raise RuntimeError("VTXWriter: write after close")
        if self._topology is None:
            self._build_topology()
        X, conn, offsets, types = self._topology
        step = len(self._times)
        out = ['<?xml version="1.0"?>\n<VTKFile type="UnstructuredGrid" version="1.0" byte_order="LittleEndian" '
               'header_type="UInt64">\n<UnstructuredGrid>\n',
               f'<FieldData><DataArray type="Float64" Name="TimeValue" NumberOfTuples="1" format="ascii">{float(t)!r}'
               '</DataArray></FieldData>\n',
               f'<Piece NumberOfPoints="{X.shape[0]}" NumberOfCells="{offsets.shape[0]}">\n<PointData>\n']
        for f in self._fns:
            h = f._storage.rhost()  # (a read: the block is not checked out)
            if f._comp is None:  # blocked function: vector data, padded to 3 components
                v = np.zeros((h.shape[0], 3))
                v[:, : h.shape[1]] = h
                out.append(_data_array(f.name, v, 3))
            else:
                out.append(_data_array(f.name, np.ascontiguousarray(h[:, f._comp])))
        out.append("</PointData>\n<Points>\n" + _data_array("Points", X, 3) + "</Points>\n<Cells>\n")
        out.append(_data_array("connectivity", conn) + _data_array("offsets", offsets) + _data_array("types", types))
        out.append("</Cells>\n</Piece>\n</UnstructuredGrid>\n</VTKFile>\n")
        path = self._piece_name(step)
        with open(path, "w") as fh:
            fh.write("".join(out))
        self._times.append(float(t))
        self._files.append(path)
        self._write_pvd()

    def _write_pvd(self):
        if self._rank != 0:
            return
        lines = ['<?xml version="1.0"?>\n<VTKFile type="Collection" version="0.1" byte_order="LittleEndian">\n<Collection>\n']
        for step, t in enumerate(self._times):
            for r in range(self._size):
                part = f"_p{r}" if self._size > 1 else ""
                name = os.path.basename(f"{self._stem}{part}_{step:06d}.vtu")
                lines.append(f'<DataSet timestep="{t!r}" part="{r}" file="{name}"/>\n')
        lines.append("</Collection>\n</VTKFile>\n")
        with open(self._stem + ".pvd", "w") as fh:
            fh.write("".join(lines))

    def close(self):
        self._closed = True

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def read_vtu(path: str) -> dict:
    """Minimal reader of the files this module writes (tests, post-processing without VTK):
    {"points", "connectivity", "offsets", "types", "time", "point_data": {name: array}}."""
    import xml.etree.ElementTree as ET

    np_t = {"Float64": np.float64, "Int64": np.int64, "UInt8": np.uint8}
    root = ET.parse(path).getroot()

    def arr(el):
        raw = base64.b64decode(el.text.strip())
        n = int(np.frombuffer(raw[:8], dtype=np.uint64)[0])
        a = np.frombuffer(raw[8:8 + n], dtype=np_t[el.get("type")])
        nc = int(el.get("NumberOfComponents", "1"))
        return a.reshape(-1, nc) if nc > 1 else a

    piece = root.find("UnstructuredGrid/Piece")
    out = {"point_data": {}}
    out["time"] = float(root.find("UnstructuredGrid/FieldData/DataArray").text)
    out["points"] = arr(piece.find("Points/DataArray"))
    for el in piece.find("Cells"):
        out[el.get("Name")] = arr(el)
    for el in piece.find("PointData"):
        out["point_data"][el.get("Name")] = arr(el)
    return out
