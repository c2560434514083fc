"""Minimal PCD v0.7 reader/writer (x y z float32; DATA ascii | binary), the input format of
the reference's entry points (pcl::io::loadPCDFile<PointXYZ>, src/Registration.cpp:87,
252-253; savePCDFileBinary, Registration.cpp:394).  Extra fields are ignored on read."""
import numpy as np


def read_pcd(path):
    with open(path, "rb") as f:
        hdr = {}
        while True:
            line = f.readline()
            if not line:
                raise ValueError("PCD: no DATA line")
            s = line.decode("ascii", "replace").strip()
            if not s or s.startswith("#"):
                continue
            key, _, val = s.partition(" ")
            hdr[key.upper()] = val.split()
            if key.upper() == "DATA":
                break
        fields = hdr["FIELDS"]
        sizes = [int(v) for v in hdr["SIZE"]]
        types = hdr["TYPE"]
        counts = [int(v) for v in hdr.get("COUNT", ["1"] * len(fields))]
        npts = int(hdr["POINTS"][0]) if "POINTS" in hdr else int(hdr["WIDTH"][0]) * int(hdr["HEIGHT"][0])
        mode = hdr["DATA"][0].lower()
        dt = []
        for name, sz, tp, cnt in zip(fields, sizes, types, counts):
            code = {"F": "f", "I": "i", "U": "u"}[tp.upper()] + str(sz)
            dt.append((name, "<" + code) if cnt == 1 else (name, "<" + code, (cnt,)))
        dt = np.dtype(dt)
        if mode == "binary":
            rec = np.frombuffer(f.read(npts * dt.itemsize), dtype=dt, count=npts)
            xyz = np.stack([rec["x"], rec["y"], rec["z"]], axis=1).astype(np.float32)
        elif mode == "ascii":
            arr = np.loadtxt(f, dtype=np.float64, ndmin=2)
            cols, c = {}, 0
            for name, cnt in zip(fields, counts):
                cols[name] = c
                c += cnt
            xyz = arr[:npts, [cols["x"], cols["y"], cols["z"]]].astype(np.float32)
        else:
            raise ValueError("PCD: unsupported DATA mode %r" % mode)
    return np.ascontiguousarray(xyz)


def write_pcd_binary(path, xyz):
    xyz = np.ascontiguousarray(np.asarray(xyz, np.float32)[:, :3])
    n = len(xyz)
    hdr = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\n"
           "TYPE F F F\nCOUNT 1 1 1\nWIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA binary\n" % (n, n))
    with open(path, "wb") as f:
        f.write(hdr.encode("ascii"))
        f.write(xyz.tobytes())
