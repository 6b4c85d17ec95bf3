"""Loading of the golden fixtures (tests/golden/*.npz, made by tools/make_golden.py
from the unmodified reference binary)."""
import os

import numpy as np

import rlutil

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Fixture:
    def __init__(self, name, tmp_path, painting=None):
        self.z = np.load(os.path.join(GOLD, name + ".npz"))
        self.N, self.L, self.W, _ = [int(x) for x in self.z["meta"]]
        self.dir = str(tmp_path)
        for k in self.z.files:
            if k.startswith("in/"):
                self.z[k].tofile(os.path.join(self.dir, k[3:]))
        self.chunk = rlutil.read_chunk(self.dir)
        self.painting = painting
        if painting:  # --painting theta,rho as Paint.cpp:38-61 applies it (std::stof)
            th, rho = painting
            self.chunk.theta = float(np.float32(th))
            self.chunk.r = self.chunk.r * float(np.float32(rho))

    def paint_file(self, w):
        return self.z["paint/relate_%d.bin" % w].tobytes()

    def write_paint_files(self, d):
        os.makedirs(d, exist_ok=True)
        for w in range(self.W):
            open(os.path.join(d, "relate_%d.bin" % w), "wb").write(self.paint_file(w))

    def repaint(self, w):
        """-> list over targets of (logscales[D], top[D,N]) from the reference's RePaintSection"""
        buf = self.z["repaint/w%d" % w].tobytes()
        N = int(np.frombuffer(buf, np.int32, 1, 0)[0])
        pos, out = 4, []
        for _ in range(N):
            D = int(np.frombuffer(buf, np.int32, 1, pos)[0]); pos += 4
            ls = np.frombuffer(buf, np.float32, D, pos); pos += 4 * D
            top = np.frombuffer(buf, np.float32, D * N, pos).reshape(D, N); pos += 4 * D * N
            out.append((ls, top))
        return out

    def matrices(self, w):
        """-> list of (snp, d[N,N]) from the reference's GetMatrix"""
        buf = self.z["matrix/w%d" % w].tobytes()
        N = int(np.frombuffer(buf, np.int32, 1, 0)[0])
        pos, out = 4, []
        while pos < len(buf):
            s = int(np.frombuffer(buf, np.int32, 1, pos)[0]); pos += 4
            out.append((s, np.frombuffer(buf, np.float32, N * N, pos).reshape(N, N))); pos += 4 * N * N
        return out

    def dump_windows(self):
        return sorted(int(k.split("w")[1]) for k in self.z.files if k.startswith("repaint/w"))
