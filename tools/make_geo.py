#!/usr/bin/env python3
"""Reader / writer for corona-13's `.geo` container (include/prims.h:26-35, include/geo.h:3-4) and a
generator of a finer backdrop for tests of trees that do not fit LDS:

    python3 tools/make_geo.py subdivide scenes/geo/plane.geo scenes/geo/plane_fine.geo 2

splits every quad of the input into k x k quads (bilinear positions in float32; normals and texture coordinates of the
nearest parent corner are reused, so no re-encoding is involved). Deterministic: same input, same bytes out.
"""
import struct
import sys

import numpy as np

MAGIC, VERSION = 0xc01337, 2
VTXIDX = np.dtype([("v", "<u4"), ("uv", "<u4")])
VTX = np.dtype([("p", "<f4", 3), ("n", "<u4")])


def read_geo(fn):
    d = open(fn, "rb").read()
    magic, ver, nprims, vio, vo = struct.unpack("<iiQQQ", d[:32])
    if magic != MAGIC or ver != VERSION:
        raise ValueError(f"{fn}: not a version-2 .geo")
    primid = np.frombuffer(d, dtype="<u8", count=nprims, offset=32).copy()
    vtxidx = np.frombuffer(d, dtype=VTXIDX, count=(vo - vio) // 8, offset=vio).copy()
    vtx = np.frombuffer(d, dtype=VTX, count=(len(d) - vo) // 16, offset=vo).copy()
    return primid, vtxidx, vtx


def write_geo(fn, primid, vtxidx, vtx):
    """primid_t: extra:3 | shapeid:29 | vi:28 | mb:1 | vcnt:3 (include/corona_common.h:45-53); shapeid is 0 on disk"""
    vio = 32 + 8 * len(primid)
    vo = vio + 8 * len(vtxidx)
    with open(fn, "wb") as f:
        f.write(struct.pack("<iiQQQ", MAGIC, VERSION, len(primid), vio, vo))
        f.write(np.ascontiguousarray(primid, dtype="<u8").tobytes())
        f.write(np.ascontiguousarray(vtxidx, dtype=VTXIDX).tobytes())
        f.write(np.ascontiguousarray(vtx, dtype=VTX).tobytes())


def motion_blur(primid, vtxidx, vtx, delta, turn_deg=0.0):
    """every primitive gets the motion-blur bit; vertices are stored interleaved: 2i = shutter open, 2i+1 = shutter close
    (include/geo.h:108-138). Shutter close = shutter open turned by turn_deg about the z axis through the centroid, then moved by delta."""
    f32 = np.float32
    P0 = vtx["p"].astype(np.float32)
    c = P0.mean(axis=0, dtype=np.float64).astype(np.float32)
    a = np.radians(turn_deg)
    R = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], dtype=np.float32)
    P1 = ((P0 - c) @ R.T + c + np.asarray(delta, dtype=np.float32)).astype(np.float32)
    out = np.zeros(2 * len(vtx), dtype=VTX)
    out["p"][0::2] = P0
    out["p"][1::2] = P1
    out["n"][0::2] = vtx["n"]
    out["n"][1::2] = vtx["n"]          # the encoded normals are kept (exact for a pure translation)
    return primid | (np.uint64(1) << np.uint64(60)), vtxidx, out


def subdivide_quads(primid, vtxidx, vtx, k):
    vcnt = (primid >> np.uint64(61)) & np.uint64(7)
    if not (vcnt == 4).all():
        raise ValueError("only quad meshes")
    vi0 = ((primid >> np.uint64(32)) & np.uint64(0x0fffffff)).astype(np.int64)
    corners = vtxidx[vi0[:, None] + np.arange(4)[None, :]]                      # [nq, 4] records
    P = vtx["p"][corners["v"]]                                                   # [nq, 4, 3] float32
    N = vtx["n"][corners["v"]]
    UV = corners["uv"]
    out_v, out_vi, out_p = [], [], []
    f32 = np.float32
    for a in range(k):
        for b in range(k):
            quad_p, quad_n, quad_uv = [], [], []
            # corner order of a corona quad: v0 -> v1 -> v2 -> v3 around the face; (s,t) in [0,1]^2 with v0=(0,0), v1=(1,0), v2=(1,1), v3=(0,1)
            for s, t in ((a, b), (a + 1, b), (a + 1, b + 1), (a, b + 1)):
                fs, ft = f32(s) / f32(k), f32(t) / f32(k)
                w = [(f32(1) - fs) * (f32(1) - ft), fs * (f32(1) - ft), fs * ft, (f32(1) - fs) * ft]
                p = (w[0] * P[:, 0] + w[1] * P[:, 1]) + (w[2] * P[:, 2] + w[3] * P[:, 3])
                near = int(np.argmax(w))
                quad_p.append(p.astype(np.float32)); quad_n.append(N[:, near]); quad_uv.append(UV[:, near])
            out_p.append((np.stack(quad_p, 1), np.stack(quad_n, 1), np.stack(quad_uv, 1)))
    nq = len(primid)
    pos = np.concatenate([x[0] for x in out_p], 0).reshape(-1, 3)               # [k*k*nq*4, 3]
    nrm = np.concatenate([x[1] for x in out_p], 0).reshape(-1)
    uv = np.concatenate([x[2] for x in out_p], 0).reshape(-1)
    nv = len(pos)
    new_vtx = np.zeros(nv, dtype=VTX); new_vtx["p"] = pos; new_vtx["n"] = nrm
    new_vi = np.zeros(nv, dtype=VTXIDX); new_vi["v"] = np.arange(nv, dtype=np.uint32); new_vi["uv"] = uv
    new_prim = (np.uint64(4) << np.uint64(61)) | (np.arange(0, nv, 4, dtype=np.uint64) << np.uint64(32))
    assert len(new_prim) == k * k * nq
    return new_prim, new_vi, new_vtx


if __name__ == "__main__":
    if len(sys.argv) == 5 and sys.argv[1] == "subdivide":
        write_geo(sys.argv[3], *subdivide_quads(*read_geo(sys.argv[2]), int(sys.argv[4])))
    elif len(sys.argv) in (7, 8) and sys.argv[1] == "mb":
        # python3 tools/make_geo.py mb in.geo out.geo dx dy dz [turn degrees about z]
        write_geo(sys.argv[3], *motion_blur(*read_geo(sys.argv[2]), [float(x) for x in sys.argv[4:7]], float(sys.argv[7]) if len(sys.argv) == 8 else 0.0))
    elif len(sys.argv) == 3 and sys.argv[1] == "info":
        p, vi, v = read_geo(sys.argv[2])
        print(len(p), "prims", len(vi), "vtxidx", len(v), "vertices; kinds", sorted(set(int(x) for x in (p >> np.uint64(61)) & np.uint64(7))))
    else:
        print(__doc__)
        sys.exit(1)
