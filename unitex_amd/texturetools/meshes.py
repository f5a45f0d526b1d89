"""Synthetic UV-mapped test meshes + minimal OBJ / GLB I/O (host-side, numpy).

The reference's shipped meshes are missing from /root/reference (.MISSING_LARGE_BLOBS) and trimesh /
open3d / xatlas are not installed, so benchmarks and tests run on generated meshes that carry their
own UV atlas (SURVEY 8d "Synthetic inputs"): a bumpy lat-long sphere (self-occluding, single chart with
a seam) at any face count.  OBJ I/O mirrors what the reference round-trips through
cache/processed_mesh.obj (pipeline.py:171-179, 330)."""
import json
import struct

import numpy as np


def make_bumpy_sphere(n_lon=64, n_lat=32, bump=0.18, scale=0.95):
    """Returns verts [V,3] f32, faces [F,3] i32, uvs [V,2] f32 in [0,1] (one uv per vertex: seam and pole
    vertices are duplicated), normalised so the largest bbox side is 2*scale and centred (A7)."""
    lon = np.linspace(0.0, 2.0 * np.pi, n_lon + 1)
    lat = np.linspace(0.0, np.pi, n_lat + 1)
    LON, LAT = np.meshgrid(lon, lat, indexing="xy")      # [n_lat+1, n_lon+1]
    r = 1.0 + bump * np.sin(5.0 * LON) * np.sin(4.0 * LAT) + 0.5 * bump * np.cos(3.0 * LON + 1.0) * np.sin(LAT) ** 2
    x = r * np.sin(LAT) * np.cos(LON)
    y = r * np.cos(LAT)
    z = r * np.sin(LAT) * np.sin(LON)
    verts = np.stack([x, y, z], -1).reshape(-1, 3)
    uvs = np.stack([LON / (2.0 * np.pi), 1.0 - LAT / np.pi], -1).reshape(-1, 2)
    # keep a 4-texel gutter inside [0,1]
    uvs = 0.01 + 0.98 * uvs
    faces = []
    W = n_lon + 1
    for j in range(n_lat):
        for i in range(n_lon):
            a, b, c, d = j * W + i, j * W + i + 1, (j + 1) * W + i, (j + 1) * W + i + 1
            if j != 0:
                faces.append((a, b, c))
            if j != n_lat - 1:
                faces.append((b, d, c))
    faces = np.asarray(faces, dtype=np.int32)
    lo, hi = verts.min(0), verts.max(0)
    s = (hi - lo).max() / (2.0 * scale)
    verts = (verts - 0.5 * (lo + hi)) / s
    return verts.astype(np.float32), faces, uvs.astype(np.float32)


def sphere_with_faces(n_faces, **kw):
    """choose the grid so that the face count is ~n_faces (F = 2*n_lon*(n_lat-1))."""
    n_lat = max(4, int(round(np.sqrt(n_faces / 4.0))))
    n_lon = max(8, int(round(n_faces / (2.0 * (n_lat - 1)))))
    return make_bumpy_sphere(n_lon, n_lat, **kw)


def save_obj(path, verts, faces, uvs=None, faces_uv=None, mtl=None):
    with open(path, "w") as f:
        if mtl:
            f.write("mtllib %s\nusemtl material_0\n" % mtl)
        for v in verts:
            f.write("v %.8g %.8g %.8g\n" % (v[0], v[1], v[2]))
        if uvs is not None:
            for t in uvs:
                f.write("vt %.8g %.8g\n" % (t[0], t[1]))
            fu = faces if faces_uv is None else faces_uv
            for a, b in zip(faces, fu):
                f.write("f %d/%d %d/%d %d/%d\n" % (a[0] + 1, b[0] + 1, a[1] + 1, b[1] + 1, a[2] + 1, b[2] + 1))
        else:
            for a in faces:
                f.write("f %d %d %d\n" % (a[0] + 1, a[1] + 1, a[2] + 1))


def load_obj(path):
    """Minimal OBJ reader: v / vt / f (triangles or fan-triangulated polygons).
    Returns verts [V,3], faces [F,3], uvs [Vt,2] | None, faces_uv [F,3] | None."""
    vs, vts, fv, ft = [], [], [], []
    with open(path) as f:
        for line in f:
            if line.startswith("v "):
                p = line.split()
                vs.append((float(p[1]), float(p[2]), float(p[3])))
            elif line.startswith("vt "):
                p = line.split()
                vts.append((float(p[1]), float(p[2])))
            elif line.startswith("f "):
                toks = line.split()[1:]
                vi, ti = [], []
                for t in toks:
                    q = t.split("/")
                    vi.append(int(q[0]))
                    ti.append(int(q[1]) if len(q) > 1 and q[1] else 0)
                nv, nt = len(vs), len(vts)
                vi = [i - 1 if i > 0 else nv + i for i in vi]
                ti = [i - 1 if i > 0 else (nt + i if i < 0 else -1) for i in ti]
                for k in range(1, len(vi) - 1):
                    fv.append((vi[0], vi[k], vi[k + 1]))
                    ft.append((ti[0], ti[k], ti[k + 1]))
    verts = np.asarray(vs, dtype=np.float32)
    faces = np.asarray(fv, dtype=np.int32)
    if vts and all(t[0] >= 0 for t in ft):
        return verts, faces, np.asarray(vts, dtype=np.float32), np.asarray(ft, dtype=np.int32)
    return verts, faces, None, None


def unify_uv_indexing(verts, faces, uvs, faces_uv):
    """One vertex per (position index, uv index) pair -> faces index both arrays (like the reference's
    merge_vertices(merge_tex=False) view used for faces_2d: structure_v2.py:272-288)."""
    key = faces.astype(np.int64) * (uvs.shape[0] + 1) + faces_uv.astype(np.int64)
    uniq, inv = np.unique(key.reshape(-1), return_inverse=True)
    vi = (uniq // (uvs.shape[0] + 1)).astype(np.int64)
    ti = (uniq % (uvs.shape[0] + 1)).astype(np.int64)
    return verts[vi], inv.reshape(-1, 3).astype(np.int32), uvs[ti]


def save_glb(path, verts, faces, uvs, texture_rgb_u8):
    """Minimal glTF 2.0 binary with one textured primitive (PBR metallic 0 / roughness 1, as
    link_rgb_to_mesh sets them: io/link_pbr_to_mesh.py:16-23).  texture row 0 = top of the image."""
    import io
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(texture_rgb_u8).save(buf, format="PNG")
    png = buf.getvalue()
    v = np.ascontiguousarray(verts, dtype=np.float32)
    t = np.ascontiguousarray(np.stack([uvs[:, 0], 1.0 - uvs[:, 1]], -1), dtype=np.float32)  # glTF v is top-down
    idx = np.ascontiguousarray(faces, dtype=np.uint32).reshape(-1)
    chunks, offs = [], []
    pos = 0
    for b in (v.tobytes(), t.tobytes(), idx.tobytes(), png):
        pad = (-len(b)) % 4
        offs.append((pos, len(b)))
        chunks.append(b + b"\x00" * pad)
        pos += len(b) + pad
    bin_blob = b"".join(chunks)
    gltf = {
        "asset": {"version": "2.0", "generator": "unitex_amd"},
        "scene": 0, "scenes": [{"nodes": [0]}], "nodes": [{"mesh": 0}],
        "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "TEXCOORD_0": 1}, "indices": 2, "material": 0}]}],
        "materials": [{"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}, "metallicFactor": 0.0,
                                                "roughnessFactor": 1.0}}],
        "textures": [{"source": 0, "sampler": 0}], "samplers": [{"magFilter": 9729, "minFilter": 9729}],
        "images": [{"bufferView": 3, "mimeType": "image/png"}],
        "buffers": [{"byteLength": len(bin_blob)}],
        "bufferViews": [{"buffer": 0, "byteOffset": offs[0][0], "byteLength": offs[0][1], "target": 34962},
                        {"buffer": 0, "byteOffset": offs[1][0], "byteLength": offs[1][1], "target": 34962},
                        {"buffer": 0, "byteOffset": offs[2][0], "byteLength": offs[2][1], "target": 34963},
                        {"buffer": 0, "byteOffset": offs[3][0], "byteLength": offs[3][1]}],
        "accessors": [{"bufferView": 0, "componentType": 5126, "count": int(v.shape[0]), "type": "VEC3",
                       "min": v.min(0).tolist(), "max": v.max(0).tolist()},
                      {"bufferView": 1, "componentType": 5126, "count": int(t.shape[0]), "type": "VEC2"},
                      {"bufferView": 2, "componentType": 5125, "count": int(idx.shape[0]), "type": "SCALAR"}],
    }
    js = json.dumps(gltf, separators=(",", ":")).encode()
    js += b" " * ((-len(js)) % 4)
    with open(path, "wb") as f:
        f.write(struct.pack("<III", 0x46546C67, 2, 12 + 8 + len(js) + 8 + len(bin_blob)))
        f.write(struct.pack("<II", len(js), 0x4E4F534A))
        f.write(js)
        f.write(struct.pack("<II", len(bin_blob), 0x004E4942))
        f.write(bin_blob)
