"""VTK output of the fields at the Python surface (reference: ``VTKFile("solution.pvd").write(...)``,
src/driver.py:356-385; ``firedrake.output.VTKFile`` in src/auxilliary/callbacks.py:6,41).

Firedrake's writer interpolates every field to a piecewise-linear space on the mesh; for the broken (DG)
fields of this code that is one value per cell vertex, discontinuous across cells.  The same is written
here: every triangle contributes its own three points, the point data of a field are its values at the
three vertices of each cell.  The nodal spaces of this package contain the cell vertices as nodes (GLL and
equispaced families, degree >= 1), so no interpolation is needed -- only the positions of the three vertex
nodes within a cell's node list, found once per triangle shape from the node coordinates.

Files: ``name.pvd`` (collection, one entry per ``write`` call) and ``name_<n>.vtu`` (XML UnstructuredGrid,
inline base64 binary, uncompressed, little endian).  Host side only; nothing here touches the GPU.
"""
import base64
import os

import numpy as np

__all__ = ["VTKFile", "cell_vertex_nodes"]


def _cell_vertices(nx, ny, L=1.0):
    """Vertex coordinates (ncells, 3, 2) of the structured triangulation, cell c = 2 (j nx + i) + s:
    s = 0 lower-left triangle (x_i,y_j), (x_i+1,y_j), (x_i,y_j+1); s = 1 its point reflection
    (x_i+1,y_j+1), (x_i,y_j+1), (x_i+1,y_j)  (numbering documented in oracle/fem.py and DESIGN.md)."""
    hx, hy = L / nx, L / ny
    jj, ii = np.meshgrid(np.arange(ny), np.arange(nx), indexing="ij")
    x0, y0 = (ii * hx).ravel(), (jj * hy).ravel()
    x1, y1 = x0 + hx, y0 + hy
    low = np.stack([np.stack([x0, y0], -1), np.stack([x1, y0], -1), np.stack([x0, y1], -1)], 1)
    upp = np.stack([np.stack([x1, y1], -1), np.stack([x0, y1], -1), np.stack([x1, y0], -1)], 1)
    out = np.empty((2 * nx * ny, 3, 2))
    out[0::2] = low
    out[1::2] = upp
    return out


def _mesh_cell_vertices(mesh):
    """(ncells, 3, 2) vertex coordinates: structured numbering, or the cells of a general triangulation as given"""
    if getattr(mesh, "general", False):
        return np.asarray(mesh.vertices, dtype=float)[np.asarray(mesh.cells)]
    return _cell_vertices(mesh.nx, mesh.ny, getattr(mesh, "L", 1.0))


def cell_vertex_nodes(space):
    """Indices (2, 3) of the three vertex nodes inside a cell's node list, per triangle shape (general triangulations: every
    cell carries the same node lattice on its own vertices, so the two rows coincide)."""
    mesh = space.mesh()
    ncells = mesh.num_cells()
    X = np.asarray(space.coordinates).reshape(ncells, -1, 2)
    verts = _mesh_cell_vertices(mesh)
    if getattr(mesh, "general", False):
        tol = 1e-9 * np.sqrt(mesh.volume)
    else:
        tol = 1e-9 * getattr(mesh, "L", 1.0) / max(mesh.nx, mesh.ny)
    idx = np.empty((2, 3), dtype=int)
    for s in range(2):
        for v in range(3):
            if s >= ncells:  # a one-cell mesh
                idx[s, v] = idx[0, v]
                continue
            d = np.max(np.abs(X[s] - verts[s, v]), axis=1)
            hit = np.nonzero(d < tol)[0]
            if hit.size != 1:
                raise ValueError("function space has no node at a cell vertex (degree 0 is not supported)")
            idx[s, v] = hit[0]
    return idx


def _b64(arr):
    raw = np.ascontiguousarray(arr).tobytes()
    return (base64.b64encode(np.array([len(raw)], dtype="<u4").tobytes()) + base64.b64encode(raw)).decode("ascii")


class VTKFile:
    """``VTKFile("solution.pvd").write(Q, p, ...)`` -- every call appends one time level."""

    def __init__(self, filename, mode="w"):
        if not filename.endswith(".pvd"):
            raise ValueError("VTKFile expects a .pvd file name")
        self.filename = filename
        self._base = filename[:-4]
        self._entries = []
        self._cache = {}

    def _vertex_values(self, f):
        space = f.function_space()
        key = id(space)
        if key not in self._cache:
            self._cache[key] = cell_vertex_nodes(space)
        vn = self._cache[key]
        ncells = space.mesh().num_cells()
        data = np.asarray(f.dat.data, dtype=float)
        data = data.reshape(ncells, -1, space.value_size) if space.value_size > 1 else data.reshape(ncells, -1)
        out = np.empty((ncells, 3) + data.shape[2:])
        for s in range(2):
            out[s::2] = data[s::2][:, vn[s]]
        return out

    def write(self, *functions, time=None):
        if not functions:
            raise ValueError("nothing to write")
        mesh = functions[0].function_space().mesh()
        ncells = mesh.num_cells()
        pts = np.zeros((ncells * 3, 3), dtype="<f8")
        pts[:, :2] = _mesh_cell_vertices(mesh).reshape(-1, 2)
        conn = np.arange(ncells * 3, dtype="<i4")
        offs = np.arange(3, 3 * ncells + 1, 3, dtype="<i4")
        types = np.full(ncells, 5, dtype="<u1")  # VTK_TRIANGLE
        pdata = []
        for n, f in enumerate(functions):
            if f.function_space().mesh() is not mesh:
                raise ValueError("all functions must live on the same mesh")
            name = f.name() or f"function_{n}"
            v = self._vertex_values(f)
            if v.ndim == 3:  # vector field: VTK wants three components
                v3 = np.zeros((ncells * 3, 3), dtype="<f8")
                v3[:, : v.shape[2]] = v.reshape(ncells * 3, -1)
                pdata.append(f'<DataArray type="Float64" Name="{name}" NumberOfComponents="3" format="binary">'
                             f"{_b64(v3)}</DataArray>")
            else:
                pdata.append(f'<DataArray type="Float64" Name="{name}" format="binary">'
                             f'{_b64(v.reshape(-1).astype("<f8"))}</DataArray>')
        index = len(self._entries)
        vtu = f"{self._base}_{index}.vtu"
        with open(vtu, "w") as fh:
            fh.write('<?xml version="1.0"?>\n<VTKFile type="UnstructuredGrid" version="0.1" byte_order="LittleEndian" '
                     'header_type="UInt32">\n<UnstructuredGrid>\n')
            fh.write(f'<Piece NumberOfPoints="{ncells * 3}" NumberOfCells="{ncells}">\n')
            fh.write(f'<Points><DataArray type="Float64" NumberOfComponents="3" format="binary">{_b64(pts)}'
                     "</DataArray></Points>\n")
            fh.write(f'<Cells><DataArray type="Int32" Name="connectivity" format="binary">{_b64(conn)}</DataArray>'
                     f'<DataArray type="Int32" Name="offsets" format="binary">{_b64(offs)}</DataArray>'
                     f'<DataArray type="UInt8" Name="types" format="binary">{_b64(types)}</DataArray></Cells>\n')
            fh.write("<PointData>\n" + "\n".join(pdata) + "\n</PointData>\n</Piece>\n</UnstructuredGrid>\n</VTKFile>\n")
        self._entries.append((float(index if time is None else time), os.path.basename(vtu)))
        with open(self.filename, "w") as fh:
            fh.write('<?xml version="1.0"?>\n<VTKFile type="Collection" version="0.1" byte_order="LittleEndian">\n'
                     "<Collection>\n")
            for t, name in self._entries:
                fh.write(f'<DataSet timestep="{t}" part="0" file="{name}"/>\n')
            fh.write("</Collection>\n</VTKFile>\n")
        return vtu
