"""Stand-in for plyfile (absent here): ASCII-PLY vertex reader, enough for vgtk/pc/io.py:load_ply."""
import numpy as np


class PlyElement:
    pass


class PlyData(dict):
    @staticmethod
    def read(fn):
        raw = open(fn, "rb").read()
        head, body = raw.split(b"end_header\n", 1)
        lines = head.decode().split("\n")
        assert any("format ascii" in l for l in lines), "stub handles ascii only"
        nv = [int(l.split()[-1]) for l in lines if l.startswith("element vertex")][0]
        rows = [r.split() for r in body.decode().strip().split("\n")[:nv]]
        arr = np.array([[float(x) for x in r[:3]] for r in rows], dtype=np.float32)
        d = PlyData()
        d["vertex"] = {"x": arr[:, 0], "y": arr[:, 1], "z": arr[:, 2]}
        return d
