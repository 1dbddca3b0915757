"""Stand-in for trimesh (absent here): only what vgtk/functional/rotation.py:237-345 touches
(load_mesh of the binary sphere12.ply, fix_normals, face_normals, face_adjacency).
face_adjacency ordering is re-implemented from the documented trimesh behaviour (edges sorted by
the hashed key, pairs ascending) -> the column ORDER of intra_idx is unpinned vs real trimesh."""
import struct

import numpy as np


class _Sample:
    @staticmethod
    def sample_surface(mesh, n):
        raise NotImplementedError


sample = _Sample()


class Trimesh:
    def __init__(self, vertices=None, faces=None, process=False, maintain_order=True):
        self.vertices = np.asarray(vertices)
        self.faces = np.asarray(faces)

    def fix_normals(self):
        tri = self.vertices[self.faces]
        n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
        c = tri.mean(1)
        flip = (n * c).sum(1) < 0
        self.faces[flip] = self.faces[flip][:, ::-1]

    @property
    def face_normals(self):
        tri = self.vertices[self.faces]
        n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
        return n / np.linalg.norm(n, axis=1, keepdims=True)

    @property
    def face_adjacency(self):
        f = self.faces
        edges = f[:, [0, 1, 1, 2, 2, 0]].reshape(-1, 2)
        eface = np.tile(np.arange(len(f)), (3, 1)).T.reshape(-1)
        edges = np.sort(edges, axis=1)
        key = edges[:, 0].astype(np.int64) ^ (edges[:, 1].astype(np.int64) << 32)
        order = np.argsort(key, kind="stable")
        ks = key[order]
        pairs = []
        i = 0
        while i < len(ks) - 1:
            if ks[i] == ks[i + 1]:
                pairs.append(sorted((eface[order[i]], eface[order[i + 1]])))
                i += 2
            else:
                i += 1
        return np.array(pairs)


def load_mesh(path, process=False, maintain_order=True):
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    L = head.decode().split("\n")
    nv = [int(l.split()[-1]) for l in L if l.startswith("element vertex")][0]
    nf = [int(l.split()[-1]) for l in L if l.startswith("element face")][0]
    off = 0
    V = []
    for _ in range(nv):
        V.append(struct.unpack_from("<3f", body, off))
        off += 12 + 4
    F = []
    for _ in range(nf):
        n = body[off]
        off += 1
        F.append(struct.unpack_from("<%di" % n, body, off))
        off += 4 * n
        nt = body[off]
        off += 1
        off += 4 * nt
        off += 4
    return Trimesh(np.array(V, dtype=np.float64), np.array(F, dtype=np.int64))
