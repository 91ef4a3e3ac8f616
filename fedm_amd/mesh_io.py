"""Mesh ingestion and result output in the reference's wire formats (cold path).

* legacy DOLFIN XML meshes (``Mesh('mesh.xml')``, examples/streamer_discharge/fedm-streamer.py:116)
* ASCII ``.vtu`` + ``.pvd`` as DOLFIN's ``File("x.pvd") << (function, t)`` writes them -- the
  format the reference's tests read back (tests/integrated_tests/testing_utils.py:15-19)
* ``file_output``: linear interpolation of the solution to the output times
  (fedm/file_io.py:538-616)
"""
import xml.etree.ElementTree as ET
from pathlib import Path

import numpy as np

from .mesh import Mesh


_VERTEX = r'<vertex\s+index="(\d+)"\s+x="([^"]+)"\s+y="([^"]+)"'
_TRIANGLE = r'<triangle\s+index="(\d+)"\s+v0="(\d+)"\s+v1="(\d+)"\s+v2="(\d+)"'


def read_dolfin_xml(path):
    """<dolfin><mesh celltype="triangle" dim="2"><vertices><vertex index x y/>...<cells><triangle index v0 v1 v2/>

    Files in DOLFIN's own attribute order (what DOLFIN and `write_dolfin_xml` write) are scanned with two
    regular expressions -- a 340 k-vertex mesh in a second instead of ten; anything else goes through
    the XML parser."""
    import re
    text = Path(path).read_text()
    head = re.search(r'<mesh\s[^>]*>', text)
    if head is None or 'celltype="triangle"' not in head.group(0):
        raise ValueError(f"{path}: not a DOLFIN XML triangle mesh")
    sizes = re.search(r'<vertices\s+size="(\d+)"', text), re.search(r'<cells\s+size="(\d+)"', text)
    verts, cells = re.findall(_VERTEX, text), re.findall(_TRIANGLE, text)
    if all(sizes) and len(verts) == int(sizes[0].group(1)) and len(cells) == int(sizes[1].group(1)):
        v = np.array(verts, dtype=np.float64)
        c = np.array(cells, dtype=np.int64)
        coords = np.empty((len(verts), 2))
        coords[v[:, 0].astype(np.int64)] = v[:, 1:]
        tri = np.empty((len(cells), 3), dtype=np.int32)
        tri[c[:, 0]] = c[:, 1:]
        return Mesh(coords, tri)
    root = ET.fromstring(text)
    mesh = root.find("mesh") if root.tag != "mesh" else root
    if mesh is None or mesh.get("celltype") != "triangle":
        raise ValueError(f"{path}: not a DOLFIN XML triangle mesh")
    verts = mesh.find("vertices")
    cells = mesh.find("cells")
    nv, nc = int(verts.get("size")), int(cells.get("size"))
    coords = np.empty((nv, 2))
    for v in verts.iter("vertex"):
        coords[int(v.get("index"))] = (float(v.get("x")), float(v.get("y")))
    tri = np.empty((nc, 3), dtype=np.int32)
    for c in cells.iter("triangle"):
        tri[int(c.get("index"))] = (int(c.get("v0")), int(c.get("v1")), int(c.get("v2")))
    return Mesh(coords, tri)


def write_dolfin_xml(mesh, path):
    lines = ['<?xml version="1.0"?>', '<dolfin xmlns:dolfin="http://fenicsproject.org">',
             '  <mesh celltype="triangle" dim="2">', f'    <vertices size="{mesh.num_vertices()}">']
    lines += [f'      <vertex index="{i}" x="{x!r}" y="{y!r}" />' for i, (x, y) in enumerate(mesh.coords.tolist())]
    lines += ['    </vertices>', f'    <cells size="{mesh.num_cells()}">']
    lines += [f'      <triangle index="{i}" v0="{a}" v1="{b}" v2="{c}" />'
              for i, (a, b, c) in enumerate(mesh.cells.tolist())]
    lines += ['    </cells>', '  </mesh>', '</dolfin>']
    Path(path).write_text("\n".join(lines) + "\n")


class PVDFile:
    """``File("dir/name.pvd")``: each ``write(values, name, t)`` adds ``name%06d.vtu``."""

    def __init__(self, path, mesh=None):
        self.path, self.mesh = Path(path), mesh
        self.path.parent.mkdir(parents=True, exist_ok=True)
        self.entries = []

    def __lshift__(self, item):
        """``vtkfile << (function, t)`` (fedm-gd.py:306)."""
        f, t = item if isinstance(item, tuple) else (item, 0.0)
        if self.mesh is None:
            self.mesh = f.space.mesh
        self.write(f.vector(), getattr(f, "name", None) or self.path.stem, t)
        return self

    def write(self, values, field_name, t):
        if self.mesh is None and hasattr(values, "space"):
            self.mesh, values = values.space.mesh, values.vector()
        idx = len(self.entries)
        vtu = self.path.with_name(f"{self.path.stem}{idx:06d}.vtu")
        write_vtu(vtu, self.mesh, np.asarray(values, dtype=np.float64), field_name)
        self.entries.append((t, vtu.name))
        body = "".join(f'<DataSet timestep="{tt!r}" part="0" file="{name}" />\n' for tt, name in self.entries)
        self.path.write_text('<?xml version="1.0"?>\n<VTKFile type="Collection" version="0.1">\n<Collection>\n'
                             + body + '</Collection>\n</VTKFile>\n')
        return vtu


def write_vtu(path, mesh, values, field_name):
    nv, nc = mesh.num_vertices(), mesh.num_cells()
    fmt = lambda a: " ".join(repr(float(x)) for x in a)
    pts = "  ".join(f"{x!r} {y!r} 0" for x, y in mesh.coords.tolist())
    conn = "  ".join(f"{a} {b} {c}" for a, b, c in mesh.cells.tolist())
    offs = " ".join(str(3 * (i + 1)) for i in range(nc))
    types = " ".join(["5"] * nc)
    Path(path).write_text(
        '<?xml version="1.0"?>\n<VTKFile type="UnstructuredGrid"  version="0.1"  >\n<UnstructuredGrid>\n'
        f'<Piece  NumberOfPoints="{nv}" NumberOfCells="{nc}">\n<Points>\n'
        f'<DataArray  type="Float64"  NumberOfComponents="3"  format="ascii">{pts}  </DataArray>\n</Points>\n'
        f'<Cells>\n<DataArray  type="UInt32"  Name="connectivity"  format="ascii">{conn}  </DataArray>\n'
        f'<DataArray  type="UInt32"  Name="offsets"  format="ascii">{offs} </DataArray>\n'
        f'<DataArray  type="UInt8"  Name="types"  format="ascii">{types} </DataArray>\n</Cells>\n'
        f'<PointData  Scalars="{field_name}"> \n'
        f'<DataArray  type="Float64"  Name="{field_name}"  format="ascii">{fmt(values)}  </DataArray> \n'
        '</PointData> \n</Piece>\n</UnstructuredGrid>\n</VTKFile>')


def read_vtu(path, field_name):
    """What tests/integrated_tests/testing_utils.py:15-19 does with vtk, for ASCII files."""
    root = ET.parse(path).getroot()
    for arr in root.iter("DataArray"):
        if arr.get("Name") == field_name:
            return np.array(arr.text.split(), dtype=np.float64)
    raise KeyError(field_name)


class XDMFFile:
    """``df.XDMFFile(path)`` as ``file_output`` uses it (fedm/file_io.py:597-604):
    ``write_checkpoint(values, name, t, append=True)`` adds one snapshot of a P1 field to
    ``<stem>.h5`` in DOLFIN's checkpoint layout -- ``/<name>/<name>_<k>/{vector, cell_dofs,
    x_cell_dofs, cells, mesh/{geometry, topology}}``, the layout the reference's own tests read
    (tests/integrated_tests/testing_utils.py:21-24) -- and rewrites the light ``.xdmf`` index.
    The dof numbering is the vertex numbering (``cell_dofs`` = flattened topology)."""

    class Encoding:
        HDF5 = "HDF5"

    def __init__(self, path, mesh=None):
        self.path = Path(path)
        self.path.parent.mkdir(parents=True, exist_ok=True)
        self.mesh = mesh
        self.h5path = self.path.with_suffix(".h5")
        self.counts = {}
        self.times = {}
        self.parameters = {}

    def write_checkpoint(self, values, name, t, encoding=None, append=True):
        from . import h5
        if hasattr(values, "space"):                  # a nodal Function: its mesh, its vector
            if self.mesh is None:
                self.mesh = values.space.mesh
            values = values.vector()
        values = np.asarray(values, dtype=np.float64).ravel()
        if values.size != len(self.mesh.coords):
            raise ValueError("write_checkpoint: one value per mesh vertex expected")
        if not append or not self.h5path.exists():   # DOLFIN: append=False starts the file over
            self.counts, self.times, mode = {}, {}, "w"
        else:
            mode = "a"
        cells = np.asarray(self.mesh.cells, dtype=np.int64)
        with h5.File(self.h5path, mode) as f:
            if name not in self.counts:               # appending to a file of an earlier session
                self.counts[name] = len(f.keys(f"/{name}")) if f"/{name}" in f else 0
                self.times.setdefault(name, [float("nan")] * self.counts[name])
            k = self.counts[name]
            g = f"/{name}/{name}_{k}"
            f.write(g + "/vector", values.reshape(-1, 1))
            f.write(g + "/cell_dofs", cells.reshape(-1, 1))
            f.write(g + "/x_cell_dofs", (3 * np.arange(len(cells) + 1, dtype=np.int64)).reshape(-1, 1))
            f.write(g + "/cells", np.arange(len(cells), dtype=np.int64).reshape(-1, 1))
            f.write(g + "/mesh/geometry", np.asarray(self.mesh.coords, dtype=np.float64))
            f.write(g + "/mesh/topology", cells)
        self.counts[name] = k + 1
        self.times[name].append(float(t))
        self._write_index()

    def _write_index(self):
        nv, nc = len(self.mesh.coords), len(self.mesh.cells)
        out = ['<?xml version="1.0"?>', '<Xdmf Version="3.0"><Domain>']
        for name, ts in self.times.items():
            out.append(f'<Grid Name="{name}" GridType="Collection" CollectionType="Temporal">')
            for k, t in enumerate(ts):
                g = f"{self.h5path.name}:/{name}/{name}_{k}"
                out += [f'<Grid Name="{name}_{k}" GridType="Uniform">',
                        f'<Topology TopologyType="Triangle" NumberOfElements="{nc}" NodesPerElement="3">'
                        f'<DataItem Dimensions="{nc} 3" NumberType="Int" Format="HDF">{g}/mesh/topology</DataItem></Topology>',
                        f'<Geometry GeometryType="XY"><DataItem Dimensions="{nv} 2" Format="HDF">{g}/mesh/geometry</DataItem></Geometry>',
                        f'<Time Value="{t!r}" />',
                        f'<Attribute Name="{name}" AttributeType="Scalar" Center="Node">'
                        f'<DataItem Dimensions="{nv} 1" Format="HDF">{g}/vector</DataItem></Attribute>',
                        '</Grid>']
            out.append('</Grid>')
        out.append('</Domain></Xdmf>')
        self.path.with_suffix(".xdmf").write_text("\n".join(out))


def read_h5(path, key):
    """tests/integrated_tests/testing_utils.py:21-24: the ``vector`` of every snapshot under
    ``/<key>``, in file order."""
    from . import h5
    with h5.File(path, "r") as f:
        return [f.read(f"/{key}/{sub}/vector") for sub in f.keys(f"/{key}")]


def read_checkpoint(path, key):
    """Snapshots of ``/<key>`` mapped from the file's dof numbering back to its own vertex
    numbering (works for DOLFIN-written files too): ``(coords, cells, [nodal values, ...])``."""
    from . import h5
    with h5.File(path, "r") as f:
        subs = f.keys(f"/{key}")
        fields, coords, cells = [], None, None
        for sub in subs:
            g = f"/{key}/{sub}"
            vec = f.read(g + "/vector")[:, 0]
            topo = f.read(g + "/mesh/topology", np.int64)
            cd = f.read(g + "/cell_dofs", np.int64)[:, 0].reshape(-1, 3)
            coords = f.read(g + "/mesh/geometry")
            nodal = np.empty(len(coords))
            nodal[topo.ravel()] = vec[cd.ravel()]
            fields.append(nodal)
            cells = topo
    return coords, cells, fields


def file_output(t, t_old, t_out, step, t_out_list, step_list, file_type, output_file_list,
                particle_name, u_old, u_old1, unit="s"):
    """Linear interpolation of the solution to the output times, fedm/file_io.py:538-616: same
    arguments (``u_old`` / ``u_old1`` are the lists of new / old nodal arrays, sic), same
    stepping of ``t_out`` including the reference's stale-``step`` branch."""
    units = {"ns": 1e9, "us": 1e6, "ms": 1e3, "s": 1.0}
    if unit not in units:
        raise ValueError(f"fedm.file_output: unit '{unit}' not valid. "
                         f"Options are {', '.join(repr(u) for u in units)}.'")
    scale = units[unit]
    if t > max(t_out_list):
        index = len(t_out_list) - 1
    else:
        index = next(x for x, val in enumerate(t_out_list) if val > t)
    while t_out <= t:
        for i in range(len(output_file_list)):
            vec = lambda f: np.asarray(f.vector() if hasattr(f, "vector") else f, dtype=np.float64)
            new, old = vec(u_old[i]), vec(u_old1[i])
            if getattr(output_file_list[i], "mesh", 0) is None and hasattr(u_old[i], "space"):
                output_file_list[i].mesh = u_old[i].space.mesh
            temp = old + (t_out - t_old) * (new - old) / (t - t_old)
            if file_type[i] == "pvd":
                output_file_list[i].write(temp, particle_name[i], t_out * scale)
            elif file_type[i] == "xdmf":
                output_file_list[i].write_checkpoint(temp, particle_name[i], t_out * scale, None, True)
            else:
                raise ValueError(f"fedm.file_output: file type '{file_type}' not recognised. "
                                 "Options are 'pvd' and 'xdmf'.")
        if t_out >= 0.999 * t_out_list[index - 1] and t_out < 0.999 * t_out_list[index]:
            step = step_list[index - 1]
        elif t_out >= 0.999 * t_out_list[index]:
            step = step_list[index]
        t_out += step
    return t_out, step
