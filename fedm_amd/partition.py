"""Mesh partitioning and halo plans for the multi-GPU path.

The reference relies on DOLFIN's implicit MPI decomposition (``mpirun -np 8``,
README.md:63-67): DOLFIN partitions the mesh with SCOTCH/ParMETIS and PETSc
scatters ghost values.  Here the *vertex* set is split by recursive coordinate
bisection (METIS is not available; RCB is deterministic, so every rank computes
the same partition without talking).  A rank owns the matrix rows of its
vertices and assembles every cell that touches one of them -- cells on a
partition boundary are evaluated on both sides -- so nothing but ghost *input*
values ever has to be communicated.

Local numbering of a part: owned vertices first, then ghost vertices grouped by
owning rank (ascending rank, ascending global id inside a group); both sides of
a halo link list the shared vertices in ascending global id, so the receive
side needs no unpacking.
"""
from dataclasses import dataclass
from typing import List

import numpy as np


def partition_rcb(coords, n_parts):
    """part[v] in [0, n_parts): recursive coordinate bisection balancing vertex counts."""
    coords = np.asarray(coords, dtype=np.float64)
    part = np.zeros(coords.shape[0], dtype=np.int32)

    def split(idx, first, count):
        if count == 1:
            part[idx] = first
            return
        left = count // 2
        ext = coords[idx].max(axis=0) - coords[idx].min(axis=0)
        axis = int(np.argmax(ext))
        order = np.argsort(coords[idx, axis], kind="stable")
        cut = int(round(idx.size * left / count))
        split(idx[order[:cut]], first, left)
        split(idx[order[cut:]], first + left, count - left)

    split(np.arange(coords.shape[0]), 0, int(n_parts))
    return part


@dataclass
class LocalMesh:
    rank: int
    n_parts: int
    coords: np.ndarray          # (n_local, 2): owned vertices, then ghosts
    cells: np.ndarray           # (n_cells_local, 3) in local numbering
    cell_global: np.ndarray     # global cell ids of the local cells
    vertex_global: np.ndarray   # global vertex id of every local vertex
    n_owned: int
    neighbours: np.ndarray      # ranks exchanged with, ascending
    send_ptr: np.ndarray        # (n_nb+1)
    send_idx: np.ndarray        # local (owned) vertex ids to send, per neighbour
    recv_ptr: np.ndarray        # (n_nb+1), ghost offsets relative to n_owned

    @property
    def n_ghost(self):
        return self.coords.shape[0] - self.n_owned


def local_mesh(coords, cells, part, rank):
    """Sub-mesh of ``rank``: its owned vertices, every cell touching one, the ghosts."""
    coords = np.asarray(coords, dtype=np.float64)
    cells = np.asarray(cells, dtype=np.int64)
    part = np.asarray(part)
    n_parts = int(part.max()) + 1
    cpart = part[cells]                                   # (Nc,3)
    mine = (cpart == rank).any(axis=1)
    cell_global = np.nonzero(mine)[0]
    lc = cells[cell_global]
    verts = np.unique(lc)
    owner = part[verts]
    owned = verts[owner == rank]
    ghosts = verts[owner != rank]
    gorder = np.lexsort((ghosts, part[ghosts]))           # by owner rank, then global id
    ghosts = ghosts[gorder]
    vertex_global = np.concatenate([owned, ghosts])
    lookup = np.full(coords.shape[0], -1, dtype=np.int64)
    lookup[vertex_global] = np.arange(vertex_global.size)
    gowner = part[ghosts]
    neighbours = np.unique(gowner)
    recv_ptr = np.concatenate([[0], np.cumsum([np.count_nonzero(gowner == q) for q in neighbours])])
    # what rank q needs from me: my owned vertices that sit in a cell with a q-owned vertex
    send_lists = []
    for q in neighbours:
        touch = (cpart == q).any(axis=1) & mine
        v = np.unique(cells[touch])
        v = v[part[v] == rank]
        send_lists.append(lookup[np.sort(v)])
    # a rank that needs my vertices also owns ghosts of mine (cells are shared), so the
    # neighbour sets are symmetric by construction
    send_ptr = np.concatenate([[0], np.cumsum([len(s) for s in send_lists])])
    send_idx = np.concatenate(send_lists) if send_lists else np.zeros(0, dtype=np.int64)
    return LocalMesh(rank=rank, n_parts=n_parts, coords=coords[vertex_global],
                     cells=lookup[lc].astype(np.int32), cell_global=cell_global,
                     vertex_global=vertex_global, n_owned=int(owned.size),
                     neighbours=neighbours.astype(np.int32),
                     send_ptr=send_ptr.astype(np.int32), send_idx=send_idx.astype(np.int32),
                     recv_ptr=recv_ptr.astype(np.int32))


def exchange_ghosts(lm: LocalMesh, values, group=None):
    """Fill the ghost rows of ``values`` (n_local, width) from their owners with
    torch.distributed point-to-point messages (any backend; gloo on CPU).  The device
    path does the same with RCCL inside libfedm_hip.so; this host version serves set-up
    code and the tests."""
    import torch
    import torch.distributed as dist
    v = np.ascontiguousarray(values, dtype=np.float64)
    flat = v.reshape(v.shape[0], -1)
    reqs, recv_bufs = [], []
    for k, q in enumerate(lm.neighbours):
        s = flat[lm.send_idx[lm.send_ptr[k]:lm.send_ptr[k + 1]]]
        reqs.append(dist.isend(torch.from_numpy(np.ascontiguousarray(s)), int(q), group=group))
        buf = torch.empty((int(lm.recv_ptr[k + 1] - lm.recv_ptr[k]), flat.shape[1]), dtype=torch.float64)
        recv_bufs.append(buf)
        reqs.append(dist.irecv(buf, int(q), group=group))
    for r in reqs:
        r.wait()
    for k, buf in enumerate(recv_bufs):
        flat[lm.n_owned + lm.recv_ptr[k]:lm.n_owned + lm.recv_ptr[k + 1]] = buf.numpy()
    return flat.reshape(v.shape)
