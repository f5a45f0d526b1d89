"""Synthetic UV-mapped test meshes + minimal OBJ / GLB I/O (host-side, numpy).

The reference's shipped meshes are missing from /root/reference (.MISSING_LARGE_BLOBS) and trimesh /
open3d / xatlas are not installed, so benchmarks and tests run on generated meshes that carry their
own UV atlas (SURVEY 8d "Synthetic inputs"): a bumpy lat-long sphere (self-occluding, single chart with
a seam) at any face count.  OBJ I/O mirrors what the reference round-trips through
cache/processed_mesh.obj (pipeline.py:171-179, 330)."""
import json
import os
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


def closed_sphere(n_lon, n_lat):
    """closed unit lat-long sphere WITHOUT UVs (shared pole vertices, no seam duplicates): the UV-less input case of
    prepare_blank_mesh and the analytic case of the vertex-normal tests."""
    lon = np.linspace(0, 2 * np.pi, n_lon, endpoint=False)
    lat = np.linspace(0, np.pi, n_lat + 1)[1:-1]
    v = [[0.0, 1.0, 0.0]] + [[np.sin(a) * np.cos(o), np.cos(a), np.sin(a) * np.sin(o)] for a in lat for o in lon] + [[0.0, -1.0, 0.0]]
    R, f = len(lat), []
    f += [[0, 1 + (i + 1) % n_lon, 1 + i] for i in range(n_lon)]
    for r in range(R - 1):
        for i in range(n_lon):
            a, b = 1 + r * n_lon + i, 1 + r * n_lon + (i + 1) % n_lon
            f += [[a, b, a + n_lon], [b, b + n_lon, a + n_lon]]
    last = len(v) - 1
    f += [[last, 1 + (R - 1) * n_lon + i, 1 + (R - 1) * n_lon + (i + 1) % n_lon] for i in range(n_lon)]
    return np.asarray(v, np.float32), np.asarray(f, np.int32)


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
    """OBJ reader: v / vt / f (triangles or fan-triangulated polygons).
    Returns verts [V,3], faces [F,3], uvs [Vt,2] | None, faces_uv [F,3] | None.
    The pipeline re-reads processed_mesh.obj in every stage (as the reference does), so the plain all-triangle file this package writes is
    parsed with a few whole-file numpy conversions; anything else (polygons, negative indices, `a//c` references, extra vertex columns) goes
    through the line-by-line reader.  Same values either way: decimal text -> float64 -> float32."""
    with open(path, "rb") as f:
        lines = f.read().split(b"\n")
    vl = [l[2:] for l in lines if l[:2] == b"v "]
    tl = [l[3:] for l in lines if l[:3] == b"vt "]
    fl = [l[2:] for l in lines if l[:2] == b"f "]
    try:
        if not vl or not fl:
            raise ValueError
        vt_ = np.fromstring(b" ".join(vl), dtype=np.float64, sep=" ")
        if vt_.size != 3 * len(vl):
            raise ValueError
        verts = vt_.astype(np.float32).reshape(-1, 3)
        ref0 = fl[0].split()[0]
        nfield = ref0.count(b"/") + 1
        joined = b" ".join(fl)
        if b"//" in joined or b"-" in joined:
            raise ValueError
        ft_ = np.fromstring(joined.replace(b"/", b" "), dtype=np.int64, sep=" ")
        if ft_.size != 3 * nfield * len(fl):
            raise ValueError
        idx = ft_.reshape(len(fl), 3, nfield) - 1
        if idx.min() < 0:
            raise ValueError
        faces = np.ascontiguousarray(idx[:, :, 0]).astype(np.int32)
        if tl and nfield >= 2:
            tt_ = np.fromstring(b" ".join(tl), dtype=np.float64, sep=" ")
            if tt_.size % len(tl) or tt_.size // len(tl) < 2:
                raise ValueError
            uvs = tt_.astype(np.float32).reshape(len(tl), -1)[:, :2]
            return verts, faces, np.ascontiguousarray(uvs), np.ascontiguousarray(idx[:, :, 1]).astype(np.int32)
        return verts, faces, None, None
    except ValueError:
        return _load_obj_generic(path)


def _load_obj_generic(path):
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
    Image.fromarray(texture_rgb_u8).save(buf, format="PNG", compress_level=int(os.environ.get("UTX_PNG_LEVEL", "1")))      # lossless either way; level 6 costs 5x the time
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


# ---------------------------------------------------------------------------------------------------------------
# Mesh input for real user meshes.  The reference goes through open3d / trimesh / UVAtlas
# (geometry/uv/uv_atlas.py:131-194, io/mesh_loader.py:22-30) -- none of which exist in this image and whose outputs
# cannot be pinned -- so the functions below are builder-defined host-side equivalents with the same role and the
# same knobs (min_faces / max_faces / scale / atlas size / gutter); they are numpy data preparation, not hot path.

_GLTF_DTYPES = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_GLTF_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


def load_glb(path):
    """glTF 2.0 binary reader: all triangle primitives of all meshes (node transforms applied), merged.
    Returns verts [V,3] f32, faces [F,3] i32, uvs [V,2] f32 in [0,1] with v bottom-up | None, texture u8 [H,W,3] | None."""
    with open(path, "rb") as f:
        blob = f.read()
    magic, version, total = struct.unpack_from("<III", blob, 0)
    if magic != 0x46546C67:
        raise ValueError("%s is not a .glb file" % path)
    off, js, binc = 12, None, b""
    while off < total:
        clen, ctype = struct.unpack_from("<II", blob, off)
        data = blob[off + 8: off + 8 + clen]
        if ctype == 0x4E4F534A:
            js = json.loads(data.decode("utf-8"))
        elif ctype == 0x004E4942:
            binc = data
        off += 8 + clen + ((-clen) % 4)
    return _gltf_scene_to_mesh(js, path, binc)


def _gltf_uri_bytes(uri, path):
    """a glTF `uri`: an RFC 2397 data URI (base64) or a file beside the .gltf"""
    if uri.startswith("data:"):
        import base64
        return base64.b64decode(uri.split(",", 1)[1])
    from urllib.parse import unquote
    base = os.path.dirname(os.path.abspath(path))
    full = os.path.normpath(os.path.join(base, unquote(uri)))
    if os.path.commonpath([base, full]) != base:      # a mesh file names its side files; it does not get to name files outside its own directory
        raise ValueError("%s: uri %r points outside the file's directory" % (path, uri))
    with open(full, "rb") as f:
        return f.read()


def load_gltf(path):
    """glTF 2.0 JSON form (the reference's header reader knows both containers: io/mesh_header_loader.py:49-55): buffers and images as data URIs or as files beside
    the .gltf.  Same outputs as load_glb."""
    with open(path, "r", encoding="utf-8") as f:
        js = json.load(f)
    return _gltf_scene_to_mesh(js, path, None)


def _gltf_scene_to_mesh(js, path, glb_bin):
    import io
    bufs = []
    for i, b in enumerate(js.get("buffers", [])):
        if "uri" in b:
            bufs.append(_gltf_uri_bytes(b["uri"], path))
        elif i == 0 and glb_bin is not None:
            bufs.append(glb_bin)       # the GLB's BIN chunk is buffer 0 when that buffer has no uri
        else:
            raise ValueError("%s: buffer %d has neither a uri nor a BIN chunk" % (path, i))

    def accessor(i):
        a = js["accessors"][i]
        bv = js["bufferViews"][a["bufferView"]]
        dt, nc = _GLTF_DTYPES[a["componentType"]], _GLTF_NCOMP[a["type"]]
        start = bv.get("byteOffset", 0) + a.get("byteOffset", 0)
        stride = bv.get("byteStride", 0)
        item = np.dtype(dt).itemsize * nc
        if stride and stride != item:
            # interleaved attributes: element i sits at start + i*stride; the last element ends `item` bytes in, NOT a full
            # stride -- reading count*stride bytes runs past the bufferView when accessor.byteOffset > 0
            span = np.frombuffer(bufs[bv.get("buffer", 0)], dtype=np.uint8, count=stride * (a["count"] - 1) + item, offset=start)
            raw = np.lib.stride_tricks.as_strided(span, shape=(a["count"], item), strides=(stride, 1))
            arr = np.frombuffer(np.ascontiguousarray(raw).tobytes(), dtype=dt).reshape(a["count"], nc)
        else:
            arr = np.frombuffer(bufs[bv.get("buffer", 0)], dtype=dt, count=a["count"] * nc, offset=start).reshape(a["count"], nc)
        if a.get("normalized") and dt != np.float32:
            arr = arr.astype(np.float32) / float(np.iinfo(dt).max)
        return arr

    def node_matrix(n):
        if "matrix" in n:
            return np.asarray(n["matrix"], dtype=np.float64).reshape(4, 4).T
        m = np.eye(4)
        if "scale" in n:
            m = np.diag(list(n["scale"]) + [1.0]) @ m
        if "rotation" in n:
            x, y, z, w = n["rotation"]
            r = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                          [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                          [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
            rm = np.eye(4); rm[:3, :3] = r
            m = rm @ m
        if "translation" in n:
            tm = np.eye(4); tm[:3, 3] = n["translation"]
            m = tm @ m
        return m

    vs, fs, ts, base, tex, have_uv = [], [], [], 0, None, True

    def visit(ni, parent):
        nonlocal base, tex, have_uv
        n = js["nodes"][ni]
        m = parent @ node_matrix(n)
        if "mesh" in n:
            for prim in js["meshes"][n["mesh"]]["primitives"]:
                if prim.get("mode", 4) != 4:
                    continue
                pos = accessor(prim["attributes"]["POSITION"]).astype(np.float64)
                pos = (np.concatenate([pos, np.ones((len(pos), 1))], 1) @ m.T)[:, :3]
                idx = accessor(prim["indices"]).reshape(-1, 3).astype(np.int64) if "indices" in prim else np.arange(len(pos)).reshape(-1, 3)
                vs.append(pos.astype(np.float32)); fs.append(idx + base); base += len(pos)
                if "TEXCOORD_0" in prim["attributes"]:
                    t = accessor(prim["attributes"]["TEXCOORD_0"]).astype(np.float32)
                    ts.append(np.stack([t[:, 0], 1.0 - t[:, 1]], -1))        # glTF v runs top-down
                else:
                    have_uv = False
                mat = prim.get("material")
                if tex is None and mat is not None:
                    bct = js["materials"][mat].get("pbrMetallicRoughness", {}).get("baseColorTexture")
                    if bct is not None:
                        img = js["images"][js["textures"][bct["index"]]["source"]]
                        from PIL import Image
                        if "bufferView" in img:
                            bv = js["bufferViews"][img["bufferView"]]
                            tex = np.asarray(Image.open(io.BytesIO(bufs[bv.get("buffer", 0)][bv.get("byteOffset", 0): bv.get("byteOffset", 0) + bv["byteLength"]])).convert("RGB"))
                        elif "uri" in img:
                            tex = np.asarray(Image.open(io.BytesIO(_gltf_uri_bytes(img["uri"], path))).convert("RGB"))
        for c in n.get("children", []):
            visit(c, m)

    scene = js["scenes"][js.get("scene", 0)]
    for ni in scene["nodes"]:
        visit(ni, np.eye(4))
    if not vs:
        raise ValueError("no triangle primitives in %s" % path)
    verts, faces = np.concatenate(vs), np.concatenate(fs).astype(np.int32)
    uvs = np.concatenate(ts) if (have_uv and ts) else None
    return verts, faces, uvs, tex


_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2",
              "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}


def _fan(polys):
    """polygons (lists of vertex indices) -> triangles by a fan around the first corner (what trimesh does with a convex polygon)"""
    out = []
    for p_ in polys:
        for i in range(1, len(p_) - 1):
            out.append((p_[0], p_[i], p_[i + 1]))
    return np.asarray(out, dtype=np.int32).reshape(-1, 3)


def load_ply(path):
    """Stanford PLY, ascii / binary_little_endian / binary_big_endian: element `vertex` (x y z, optional s t | u v | texture_u texture_v; any other property is
    skipped) and element `face` (a list property vertex_indices | vertex_index; other properties skipped; polygons fanned).  Other elements before / between are
    skipped by their declared layout.  Returns (verts f32 [V,3], faces i32 [F,3], uvs f32 [V,2] | None)."""
    with open(path, "rb") as f:
        blob = f.read()
    end = blob.find(b"end_header")
    if not blob.startswith(b"ply") or end < 0:
        raise ValueError("%s is not a PLY file" % path)
    body = blob.index(b"\n", end) + 1
    fmt, elements = None, []
    for line in blob[:end].decode("ascii", "replace").splitlines():
        t = line.split()
        if not t or t[0] in ("ply", "comment", "obj_info"):
            continue
        if t[0] == "format":
            fmt = t[1]
        elif t[0] == "element":
            elements.append([t[1], int(t[2]), []])
        elif t[0] == "property":
            if t[1] == "list":
                elements[-1][2].append((t[4], _PLY_TYPES[t[2]], _PLY_TYPES[t[3]]))
            else:
                elements[-1][2].append((t[2], _PLY_TYPES[t[1]], None))
    if fmt not in ("ascii", "binary_little_endian", "binary_big_endian"):
        raise ValueError("%s: PLY format %r" % (path, fmt))
    verts = uvs = None
    polys = []
    if fmt == "ascii":
        tok = blob[body:].split()
        pos = 0
        for name, count, props in elements:
            rows = []
            for _ in range(count):
                row = {}
                for pn, pt, lt in props:
                    if lt is None:
                        row[pn] = float(tok[pos]); pos += 1
                    else:
                        n = int(tok[pos]); pos += 1
                        row[pn] = [int(float(x)) for x in tok[pos:pos + n]]; pos += n
                rows.append(row)
            if name == "vertex":
                verts, uvs = _ply_vertex_arrays(lambda k: np.asarray([r[k] for r in rows], dtype=np.float64), [p_[0] for p_ in props], count)
            elif name == "face":
                key = next((k for k in ("vertex_indices", "vertex_index") if rows and k in rows[0]), None)
                polys = [r[key] for r in rows] if key else []
    else:
        e = "<" if fmt == "binary_little_endian" else ">"
        pos = body
        for name, count, props in elements:
            if all(lt is None for _, _, lt in props):
                dt = np.dtype([(pn, e + pt) for pn, pt, _ in props])
                arr = np.frombuffer(blob, dtype=dt, count=count, offset=pos)
                pos += dt.itemsize * count
                if name == "vertex":
                    verts, uvs = _ply_vertex_arrays(lambda k: arr[k].astype(np.float64), [p_[0] for p_ in props], count)
                continue
            # an element with list properties: fixed-size fast path when every list of the element has the same length (all-triangle / all-quad files), else row by row
            got = None
            if count:
                lens = []
                q = pos
                for pn, pt, lt in props:
                    if lt is None:
                        q += np.dtype(pt).itemsize; lens.append(None)
                    else:
                        n = int(np.frombuffer(blob, dtype=e + pt, count=1, offset=q)[0]); lens.append(n)
                        q += np.dtype(pt).itemsize + n * np.dtype(lt).itemsize
                fields = []
                for (pn, pt, lt), n in zip(props, lens):
                    if lt is None:
                        fields.append((pn, e + pt))
                    else:
                        fields.append((pn + "__n", e + pt)); fields.append((pn, e + lt, (n,)))
                dt = np.dtype(fields)
                if pos + dt.itemsize * count <= len(blob):
                    arr = np.frombuffer(blob, dtype=dt, count=count, offset=pos)
                    if all(n is None or np.all(arr[pn + "__n"] == n) for (pn, _, _), n in zip(props, lens)):
                        got = arr
                        pos += dt.itemsize * count
            if got is not None:
                if name == "face":
                    key = next((k for k in ("vertex_indices", "vertex_index") if k in got.dtype.names), None)
                    if key:
                        polys = got[key].astype(np.int64)
                continue
            rows = []
            for _ in range(count):
                row = {}
                for pn, pt, lt in props:
                    if lt is None:
                        row[pn] = np.frombuffer(blob, dtype=e + pt, count=1, offset=pos)[0]; pos += np.dtype(pt).itemsize
                    else:
                        n = int(np.frombuffer(blob, dtype=e + pt, count=1, offset=pos)[0]); pos += np.dtype(pt).itemsize
                        row[pn] = np.frombuffer(blob, dtype=e + lt, count=n, offset=pos).astype(np.int64).tolist(); pos += n * np.dtype(lt).itemsize
                rows.append(row)
            if name == "face":
                key = next((k for k in ("vertex_indices", "vertex_index") if rows and k in rows[0]), None)
                polys = [r[key] for r in rows] if key else []
    if verts is None:
        raise ValueError("%s: no vertex element" % path)
    if isinstance(polys, np.ndarray) and polys.ndim == 2 and polys.shape[1] == 3:
        faces = polys.astype(np.int32)
    else:
        faces = _fan([list(p_) for p_ in polys])
    if len(faces) == 0:
        raise ValueError("%s holds no faces (a point cloud cannot be textured)" % path)
    return verts, faces, uvs


def _ply_vertex_arrays(col, names, count):
    v = np.stack([col("x"), col("y"), col("z")], -1).astype(np.float32) if count else np.zeros((0, 3), np.float32)
    for a, b in (("s", "t"), ("u", "v"), ("texture_u", "texture_v")):
        if a in names and b in names:
            return v, np.stack([col(a), col(b)], -1).astype(np.float32)
    return v, None


def load_stl(path):
    """STL, binary or ascii: a triangle soup; bit-equal corners are merged into one vertex (first occurrence order), as trimesh's STL loader followed by
    merge_vertices does (io/mesh_loader.py:17)."""
    with open(path, "rb") as f:
        blob = f.read()
    tri = None
    if len(blob) >= 84:
        n = struct.unpack_from("<I", blob, 80)[0]
        if 84 + 50 * n == len(blob):         # the binary form's length is exact; an ascii file that starts with 80 bytes of text never satisfies this
            rec = np.frombuffer(blob, dtype=np.dtype([("n", "<f4", (3,)), ("v", "<f4", (9,)), ("a", "<u2")]), count=n, offset=84)
            tri = rec["v"].reshape(-1, 3).astype(np.float32)
    if tri is None:
        pts = []
        for line in blob.decode("ascii", "replace").splitlines():
            t = line.split()
            if len(t) == 4 and t[0] == "vertex":
                pts.append((float(t[1]), float(t[2]), float(t[3])))
        if not pts or len(pts) % 3:
            raise ValueError("%s is not an STL file" % path)
        tri = np.asarray(pts, dtype=np.float32)
    uniq, first, inv = np.unique(tri.view(np.uint32).reshape(-1, 3), axis=0, return_index=True, return_inverse=True)
    order = np.argsort(first)                 # keep the file's vertex order
    rank = np.empty_like(order); rank[order] = np.arange(len(order))
    return tri[first[order]], rank[inv.reshape(-1)].reshape(-1, 3).astype(np.int32)


def load_off(path):
    """Object File Format (OFF / COFF headers; the counts on the header line or on the next; polygons fanned; per-face colours behind the indices ignored)."""
    with open(path, "r") as f:
        lines = [t for t in (line.split("#", 1)[0].split() for line in f) if t]
    if not lines or not lines[0][0].endswith("OFF"):
        raise ValueError("%s is not an OFF file" % path)
    head = lines[0][1:]
    at = 1
    if len(head) < 2:
        head, at = lines[1], 2
    nv, nf = int(head[0]), int(head[1])
    if len(lines) < at + nv + nf:
        raise ValueError("%s: OFF header promises %d vertices + %d faces, the file holds %d lines" % (path, nv, nf, len(lines) - at))
    verts = np.asarray([t[:3] for t in lines[at:at + nv]], dtype=np.float64).astype(np.float32).reshape(nv, 3)      # COFF: colours behind x y z, cut
    polys = []
    for t in lines[at + nv:at + nv + nf]:
        n = int(t[0])
        polys.append([int(x) for x in t[1:1 + n]])
    return verts, _fan(polys)


def load_mesh(path):
    """.obj / .glb / .gltf / .ply / .stl / .off -> (verts, faces, uvs | None [one per vertex], texture | None).  The reference hands any path to trimesh.load
    (io/mesh_loader.py:22-30); these are the formats its own tools name (mesh/structure.py:73-83, render/blender/render_blender.py:59) plus the
    interchange formats mesh generators write."""
    ext = path.lower().rsplit(".", 1)[-1]
    if ext == "obj":
        v, f, uv, fuv = load_obj(path)
        if uv is not None:
            v, f, uv = unify_uv_indexing(v, f, uv, fuv)
        return v, f, uv, None
    if ext == "glb":
        return load_glb(path)
    if ext == "gltf":
        return load_gltf(path)
    if ext == "ply":
        v, f, uv = load_ply(path)
        return v, f, uv, None
    if ext == "stl":
        v, f = load_stl(path)
        return v, f, None, None
    if ext == "off":
        v, f = load_off(path)
        return v, f, None, None
    raise NotImplementedError("mesh format .%s is not read natively (supported: .obj, .glb, .gltf, .ply, .stl, .off)" % ext)


def clean_mesh(verts, faces, merge_eps=1e-8):
    """merge_close_vertices + remove_degenerate_triangles + remove_unreferenced_vertices (uv_atlas.py:150-152,169)."""
    key = np.round(verts.astype(np.float64) / max(merge_eps, 1e-12)).astype(np.int64) if merge_eps > 0 else None
    if key is not None and np.abs(key).max() < 2 ** 62:
        _, first, inv = np.unique(key, axis=0, return_index=True, return_inverse=True)
        verts, faces = verts[first], inv.reshape(-1)[faces]
    a, b, c = verts[faces[:, 0]].astype(np.float64), verts[faces[:, 1]].astype(np.float64), verts[faces[:, 2]].astype(np.float64)
    area2 = np.linalg.norm(np.cross(b - a, c - a), axis=1)
    keep = (faces[:, 0] != faces[:, 1]) & (faces[:, 1] != faces[:, 2]) & (faces[:, 0] != faces[:, 2]) & (area2 > 0)
    faces = faces[keep]
    used = np.unique(faces)
    remap = -np.ones(len(verts), dtype=np.int64); remap[used] = np.arange(len(used))
    return verts[used].astype(np.float32), remap[faces].astype(np.int32)


def subdivide_midpoint(verts, faces):
    """one 1:4 midpoint subdivision (the reference calls open3d subdivide_loop x2 for meshes under min_faces,
    uv_atlas.py:164-165; positions are not smoothed here)."""
    e = np.sort(np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]]), axis=1)
    ue, inv = np.unique(e, axis=0, return_inverse=True)
    mid = 0.5 * (verts[ue[:, 0]] + verts[ue[:, 1]])
    F, V = len(faces), len(verts)
    m01, m12, m20 = inv[:F] + V, inv[F:2 * F] + V, inv[2 * F:] + V
    a, b, c = faces[:, 0], faces[:, 1], faces[:, 2]
    nf = np.concatenate([np.stack([a, m01, m20], 1), np.stack([m01, b, m12], 1), np.stack([m20, m12, c], 1), np.stack([m01, m12, m20], 1)])
    return np.concatenate([verts, mid]).astype(np.float32), nf.astype(np.int32)


def subdivide_loop(verts, faces, iterations=1):
    """Loop subdivision (C. Loop 1987), what open3d's TriangleMesh.subdivide_loop computes [3p] and the reference applies twice to meshes under
    min_faces (uv_atlas.py:164-165): every triangle is split 1:4; a new vertex on an INTERIOR edge (a, b) with opposite corners (c, d) is
    3/8 (a + b) + 1/8 (c + d), on a boundary edge the midpoint; an old interior vertex of valence n moves to (1 - n beta) v + beta sum(neighbours),
    beta = 3/16 for n = 3 and 3/(8 n) otherwise; an old boundary vertex to 3/4 v + 1/8 (its two boundary neighbours).  Vectorised, float64."""
    v = np.asarray(verts, np.float64)
    f = np.asarray(faces, np.int64)
    for _ in range(int(iterations)):
        F, V = len(f), len(v)
        e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])                    # edge k of face i at row k * F + i
        opp = np.concatenate([f[:, 2], f[:, 0], f[:, 1]])                                   # the corner opposite to that edge
        es = np.sort(e, axis=1)
        ue, inv, cnt = np.unique(es, axis=0, return_inverse=True, return_counts=True)
        inv = inv.reshape(-1)
        interior = cnt == 2                                                                 # (non-manifold edges with > 2 faces are treated as boundary)
        osum = np.zeros((len(ue), 3)); np.add.at(osum, inv, v[opp])
        mid = v[ue[:, 0]] + v[ue[:, 1]]
        newv = np.where(interior[:, None], 0.375 * mid + 0.125 * osum, 0.5 * mid)
        # old vertices
        nsum = np.zeros((V, 3)); val = np.zeros(V)
        np.add.at(nsum, ue[:, 0], v[ue[:, 1]]); np.add.at(nsum, ue[:, 1], v[ue[:, 0]])
        np.add.at(val, ue[:, 0], 1.0); np.add.at(val, ue[:, 1], 1.0)
        bedge = ue[~interior]
        on_b = np.zeros(V, bool); on_b[bedge.reshape(-1)] = True
        bsum = np.zeros((V, 3)); np.add.at(bsum, bedge[:, 0], v[bedge[:, 1]]); np.add.at(bsum, bedge[:, 1], v[bedge[:, 0]])
        beta = np.where(val == 3, 3.0 / 16.0, 3.0 / (8.0 * np.maximum(val, 1.0)))
        moved = (1.0 - val * beta)[:, None] * v + beta[:, None] * nsum
        moved = np.where(on_b[:, None], 0.75 * v + 0.125 * bsum, moved)
        moved = np.where((val == 0)[:, None], v, moved)                                     # unreferenced vertices stay
        m01, m12, m20 = inv[:F] + V, inv[F:2 * F] + V, inv[2 * F:] + V
        a, b, c = f[:, 0], f[:, 1], f[:, 2]
        f = np.concatenate([np.stack([a, m01, m20], 1), np.stack([m01, b, m12], 1), np.stack([m20, m12, c], 1), np.stack([m01, m12, m20], 1)])
        v = np.concatenate([moved, newv])
    return v.astype(np.float32), f.astype(np.int32)


def smooth_simple(verts, faces, iterations=3):
    """open3d filter_smooth_simple(FilterScope.Vertex) [3p] (uv_atlas.py:169: three iterations on the COPY of the mesh that is unwrapped -- the UVs
    are computed on the smoothed surface and carried by the original one): v <- (v + sum of its edge neighbours) / (1 + valence) per iteration."""
    v = np.asarray(verts, np.float64)
    f = np.asarray(faces, np.int64)
    ue = np.unique(np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]), axis=1), axis=0)
    val = np.zeros(len(v)); np.add.at(val, ue[:, 0], 1.0); np.add.at(val, ue[:, 1], 1.0)
    for _ in range(int(iterations)):
        acc = v.copy()
        np.add.at(acc, ue[:, 0], v[ue[:, 1]]); np.add.at(acc, ue[:, 1], v[ue[:, 0]])
        v = acc / (1.0 + val)[:, None]
    return v.astype(np.float32)


def decimate_qem(verts, faces, max_faces, boundary_weight=1.0):
    """quadric-error-metric edge-collapse decimation to <= max_faces triangles (utx_mesh_decimate_qem: host C++ in libunitex_hip.so, the published
    Garland-Heckbert algorithm; the reference calls open3d's simplify_quadric_decimation, uv_atlas.py:155-163 [3p])."""
    import ctypes as C
    from .._lib import load_library
    lib = load_library()
    v = np.ascontiguousarray(verts, np.float32); f = np.ascontiguousarray(faces, np.int32)
    vo = np.empty_like(v); fo = np.empty_like(f)
    nv, nf = C.c_int(0), C.c_int(0)
    rc = lib.utx_mesh_decimate_qem(v.ctypes.data_as(C.c_void_p), len(v), f.ctypes.data_as(C.c_void_p), len(f), int(max_faces), float(boundary_weight),
                                   vo.ctypes.data_as(C.c_void_p), fo.ctypes.data_as(C.c_void_p), C.byref(nv), C.byref(nf))
    if rc < 0:
        raise RuntimeError("utx_mesh_decimate_qem -> %d" % rc)
    if rc == 1:      # every remaining collapse was rejected (flip / non-manifold fan / link condition): a valid mesh, but larger than asked
        import warnings
        warnings.warn("decimate_qem: stopped at %d faces, above the requested %d (no admissible edge collapse left)" % (nf.value, int(max_faces)), RuntimeWarning)
    return vo[: nv.value].copy(), fo[: nf.value].copy()


def decimate_cluster(verts, faces, max_faces):
    """vertex-clustering decimation until F <= max_faces (the reference uses open3d quadric decimation,
    uv_atlas.py:154-160): vertices are snapped to a uniform grid, cells collapse to their mean."""
    lo, hi = verts.min(0), verts.max(0)
    res = int(np.ceil(np.sqrt(max_faces / 2.0))) * 2
    while len(faces) > max_faces and res >= 4:
        cell = np.floor((verts - lo) / np.maximum(hi - lo, 1e-12) * (res - 1e-6)).astype(np.int64)
        key = (cell[:, 0] * res + cell[:, 1]) * res + cell[:, 2]
        uk, inv = np.unique(key, return_inverse=True)
        cnt = np.bincount(inv, minlength=len(uk)).astype(np.float64)
        nv = np.stack([np.bincount(inv, weights=verts[:, k].astype(np.float64), minlength=len(uk)) / cnt for k in range(3)], 1)
        v2, f2 = clean_mesh(nv.astype(np.float32), inv[faces].astype(np.int32), merge_eps=0)
        f2 = f2[np.sort(np.unique(np.sort(f2, 1), axis=0, return_index=True)[1])]      # drop duplicate faces
        if len(f2) <= max_faces:
            return v2, f2
        res = int(res * 0.85)
    return verts, faces


def unwrap_grid(verts, faces, atlas=2048, gutter=4.0):
    """Fallback UV atlas: every triangle gets its own right-triangle slot, two slots per square cell of a regular grid, each
    inset by `gutter`/2 texels so that no two triangles share a texel.  Bijective by construction, no distortion control.
    Positions stay SHARED (smooth vertex normals for the condition render, as with the reference's per-corner
    `triangle_uvs`, uv_atlas.py:171-175): returns verts [V,3], faces [F,3], uvs [3F,2] in [0,1], faces_uv [F,3]."""
    F = len(faces)
    ncell = (F + 1) // 2
    n = int(np.ceil(np.sqrt(ncell)))
    cs = atlas / float(n)                         # cell size in texels
    g = gutter * 0.5
    if cs < 4.0:
        raise ValueError("atlas %d too small for %d faces (%.1f texels per cell)" % (atlas, F, cs))
    if cs < 3.5 * gutter:          # dense meshes: shrink the gutter rather than fail (a 200k-face mesh has 6.5-texel cells at 2048^2)
        gutter = cs / 3.5
        g = gutter * 0.5
    fi = np.arange(F)
    cx, cy = (fi // 2) % n, (fi // 2) // n
    upper = (fi % 2) == 1
    x0, y0 = cx * cs, cy * cs
    # lower triangle: (g, g), (cs-2g-g.., g), (g, cs-...) ; upper: mirrored, with a full gutter across the diagonal
    d = cs - g * (2.0 + np.sqrt(2.0))             # leg length that keeps `gutter` texels between the two hypotenuses
    lo = np.stack([np.stack([x0 + g, y0 + g], 1), np.stack([x0 + g + d, y0 + g], 1), np.stack([x0 + g, y0 + g + d], 1)], 1)
    up = np.stack([np.stack([x0 + cs - g, y0 + cs - g], 1), np.stack([x0 + cs - g - d, y0 + cs - g], 1),
                   np.stack([x0 + cs - g, y0 + cs - g - d], 1)], 1)
    tri_uv = np.where(upper[:, None, None], up, lo) / float(atlas)          # [F,3,2]
    f_uv = np.arange(3 * F, dtype=np.int32).reshape(F, 3)
    return verts.astype(np.float32), faces.astype(np.int32), tri_uv.reshape(-1, 2).astype(np.float32), f_uv


# ---- chart-based unwrap (the reference: open3d compute_uvatlas(size=2048, gutter=4), uv_atlas.py:171-175 [3p]) ----
# axis of projection bucket b and the in-plane basis (u, v) with u x v = axis, so that front-facing triangles keep positive area
_BUCKET_AXES = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], np.float64)
_BUCKET_UV = [(1, 2), (2, 1), (2, 0), (0, 2), (0, 1), (1, 0)]


def projection_frames(n_dirs=26):
    """projection directions of the chart unwrap and an orthonormal in-plane basis (u, v) with u x v = d for each: [n,3] x 3.
    6  = the coordinate axes (round 2): a face may lie up to 54.7 degrees off its axis -> up to 73 % area stretch in projection;
    26 = axes + the 12 edge and 8 corner diagonals of the cube: every unit normal is within 27.6 degrees of one of them, i.e. the planar
         projection stretches a face's area by at most 1 / cos(27.6 deg) - 1 = 12.8 % -- inside the reference's UVAtlas bound max_stretch = 0.1667
         (uv_atlas.py:171) BY CONSTRUCTION (UVAtlas itself is [3p]: its own stretch metric is not restated, the bound is)."""
    if n_dirs == 6:
        d = _BUCKET_AXES.copy()
    elif n_dirs == 26:
        d = np.array([(x, y, z) for x in (-1, 0, 1) for y in (-1, 0, 1) for z in (-1, 0, 1) if (x, y, z) != (0, 0, 0)], np.float64)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
    else:
        raise ValueError("projection direction sets exist for 6 and 26 directions")
    ref = np.where(np.abs(d[:, 2:3]) < 0.9, np.array([[0.0, 0.0, 1.0]]), np.array([[1.0, 0.0, 0.0]]))
    u = np.cross(ref, d); u /= np.linalg.norm(u, axis=1, keepdims=True)
    v = np.cross(d, u)
    return d, u, v


def face_adjacency(faces):
    """adj [F,3] int32: the face across edge e = (v_e, v_{e+1}) of each face, -1 on a border; a non-manifold edge links its first
    two faces only."""
    F = len(faces)
    e = np.stack([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]], 1).reshape(-1, 2).astype(np.int64)      # [3F,2], edge id = 3f + e
    key = np.minimum(e[:, 0], e[:, 1]) * (int(faces.max()) + 1) + np.maximum(e[:, 0], e[:, 1])
    order = np.argsort(key, kind="stable")
    ks = key[order]
    adj = -np.ones(3 * F, dtype=np.int32)
    same = ks[1:] == ks[:-1]
    first = np.ones(len(ks), bool); first[1:] = ~same                      # first edge of every run of equal keys
    second = np.zeros(len(ks), bool); second[1:] = same & first[:-1]      # its immediate successor
    i2 = np.nonzero(second)[0]
    a, b = order[i2 - 1], order[i2]
    adj[a] = (b // 3).astype(np.int32); adj[b] = (a // 3).astype(np.int32)
    return adj.reshape(F, 3)


def chart_buckets(verts, faces, adj, smooth_iters=2, cos_accept=6.0 / 7.0, dirs=None):
    """projection bucket per face: the projection direction closest to the face normal, then a few majority-vote sweeps that move a face
    into the bucket of >= 2 of its neighbours when its normal still faces that direction by more than acos(cos_accept) -- removes the
    one-face islands along bucket borders (fewer, larger charts = less seam).  cos_accept = 6/7: a face is never put into a bucket that
    would stretch it by more than 1/6 (the reference's max_stretch, uv_atlas.py:171)."""
    v = verts.astype(np.float64)
    n = np.cross(v[faces[:, 1]] - v[faces[:, 0]], v[faces[:, 2]] - v[faces[:, 0]])
    n /= np.maximum(np.linalg.norm(n, axis=1, keepdims=True), 1e-30)
    dirs = _BUCKET_AXES if dirs is None else dirs
    dots = n @ dirs.T                                            # [F, n_dirs]
    bucket = np.argmax(dots, axis=1).astype(np.int32)
    for _ in range(smooth_iters):
        nb = np.where(adj >= 0, bucket[np.maximum(adj, 0)], -1)            # [F,3]
        for b in range(len(dirs)):
            votes = (nb == b).sum(1)
            move = (votes >= 2) & (bucket != b) & (dots[:, b] > cos_accept)
            bucket = np.where(move, b, bucket).astype(np.int32)
    return bucket


def _shelf_pack(sizes, T):
    """sizes [C,2] (w, h) in texels, sorted by the caller; returns offsets [C,2] or None when they do not fit a T x T atlas."""
    x = y = shelf = 0.0
    out = np.zeros_like(sizes)
    for i, (w, h) in enumerate(sizes):
        if w > T or h > T:
            return None
        if x + w > T:
            x, y, shelf = 0.0, y + shelf, 0.0
        if y + h > T:
            return None
        out[i] = (x, y)
        x += w
        shelf = max(shelf, h)
    return out


def unwrap_charts(verts, faces, atlas=2048, gutter=4.0, device="cuda:0", max_rounds=4, n_dirs=26):
    """Chart unwrap: faces are bucketed by the projection direction their normal faces (26 directions: area stretch <= 12.8 %, inside the
    reference's max_stretch = 1/6; `projection_frames`), charts are the connected same-bucket components (labelled on
    the GPU: utx_chart_flood), every chart is projected orthographically along its direction (isometric up to the cosine of the facing
    angle) and the charts are shelf-packed at ONE texel density with `gutter` texels between them.  A
    chart that folds over itself in projection is detected by rasterising the atlas with the product's own rasteriser and counting
    texels per face; faces that lose their texels to another face of the same chart are split off into further charts (a few
    rounds); faces that are STILL overlapped after `max_rounds` become one-triangle charts of their own in a final round (a single
    triangle projected along its dominant axis cannot fold), so the atlas that is returned is bijective for every face the check can
    see (texel-centre resolution: a face that covers no texel centre at all has nothing to lose).  Should even the final round
    leave overlapped faces (it cannot by construction) a RuntimeWarning says so instead of returning the atlas silently.
    Returns verts [V,3] (shared positions), faces [F,3], uvs [3F,2] in [0,1], faces_uv [F,3]."""
    import torch
    from . import ops
    F = len(faces)
    faces = np.asarray(faces, np.int32)
    adj = face_adjacency(faces)
    dirs, bu, bv = projection_frames(n_dirs)
    bucket = chart_buckets(verts, faces, adj, dirs=dirs)
    v64 = verts.astype(np.float64)
    layer = np.zeros(F, np.int32)                     # fold-over layer: faces split off a chart get the next layer
    dev = torch.device(device)
    adj_d = torch.from_numpy(adj).to(dev)
    for rnd in range(max_rounds + 1):
        key = (bucket + len(dirs) * layer).astype(np.int32)
        chart = ops.chart_flood(adj_d, torch.from_numpy(key).to(dev)).cpu().numpy()
        ids, cidx = np.unique(chart, return_inverse=True)
        C = len(ids)
        # per-face 2-D coordinates in its chart's plane
        P = v64[faces]                                                  # [F,3,3]
        uv3 = np.stack([(P * bu[bucket][:, None, :]).sum(-1), (P * bv[bucket][:, None, :]).sum(-1)], -1)               # [F,3,2]: corners in the chart's plane
        lo = np.full((C, 2), np.inf); hi = np.full((C, 2), -np.inf)
        for k in range(3):
            np.minimum.at(lo, cidx, uv3[:, k]); np.maximum.at(hi, cidx, uv3[:, k])
        ext = hi - lo                                                   # chart extents in world units
        order = np.argsort(-ext[:, 1], kind="stable")
        # largest texel density at which the shelf packing fits
        d_lo, d_hi, best = 0.0, atlas / max(float(ext.max()), 1e-9), None
        for _ in range(40):
            d = 0.5 * (d_lo + d_hi)
            offs = _shelf_pack(ext[order] * d + gutter, atlas)
            if offs is None:
                d_hi = d
            else:
                d_lo, best = d, offs
        if best is None:
            raise ValueError("atlas %d cannot hold %d charts with a %g-texel gutter" % (atlas, C, gutter))
        off = np.zeros((C, 2)); off[order] = best
        uv = (off[cidx][:, None, :] + 0.5 * gutter + (uv3 - lo[cidx][:, None, :]) * d_lo) / float(atlas)                 # [F,3,2]
        uvs = uv.reshape(-1, 2).astype(np.float32)
        f_uv = np.arange(3 * F, dtype=np.int32).reshape(F, 3)
        # bijectivity check on the GPU, exact at texel-centre resolution: the atlas is rasterised twice with the product's rasteriser, once with
        # the faces in order and once reversed.  All UV triangles lie at z = 0, so a texel centre covered by two faces goes to the smaller face
        # id -- the other one in the reversed pass: any texel whose two winners differ is claimed twice, and the faces that lose it in the first
        # pass are the ones split off.  (Round 2 compared texels owned with the UV area, which also flagged slivers that overlap nothing.)
        uvclip = np.concatenate([uvs * 2 - 1, np.zeros((3 * F, 1), np.float32), np.ones((3 * F, 1), np.float32)], -1)
        uvc_d = torch.from_numpy(uvclip).to(dev)
        ida = ops.rasterize(uvc_d, torch.from_numpy(f_uv).to(dev), atlas, atlas)[..., 3].long()
        idb = ops.rasterize(uvc_d, torch.from_numpy(np.ascontiguousarray(f_uv[::-1])).to(dev), atlas, atlas)[..., 3].long()
        idb = torch.where(idb > 0, F + 1 - idb, idb)
        twice = ida != idb
        lost = np.zeros(F, bool)
        lost[(idb[twice] - 1).unique().cpu().numpy()] = True
        if not lost.any():
            break
        if rnd == max_rounds:
            import warnings
            warnings.warn("unwrap_charts: %d faces still share texels with another face after the one-triangle fallback" % int(lost.sum()), RuntimeWarning)
            break
        if rnd == max_rounds - 1:       # last resort: every face that is still overlapped becomes a chart of its own (unique key -> no same-key neighbour)
            layer = layer.copy()
            layer[lost] = max_rounds + 1 + np.arange(int(lost.sum()), dtype=np.int32)
        else:
            layer = np.where(lost, layer + 1, layer).astype(np.int32)
    return verts.astype(np.float32), faces, uvs, f_uv


def normalise_to_bbox(verts, scale):
    lo, hi = verts.min(0), verts.max(0)
    return ((verts - 0.5 * (lo + hi)) / ((hi - lo).max() / (2.0 * scale))).astype(np.float32)


def prepare_blank_mesh(path, min_faces=20_000, max_faces=200_000, scale=1.0, atlas=2048, gutter=4.0, unwrap="grid", device="cuda:0"):
    """preprocess_blank_mesh_o3d (uv_atlas.py:131-175) with the host-side equivalents above: rescale to the bbox;
    a mesh that already has UVs passes through; otherwise clean, bring the face count into [min_faces, max_faces],
    and unwrap.  Returns verts [V,3], faces [F,3], uvs [Vt,2], faces_uv [F,3]: positions stay shared, UVs are per corner
    (write with save_obj(..., faces_uv=faces_uv); load_mesh splits per (v, vt) pair for the inverse renderer)."""
    ext = path.lower().rsplit(".", 1)[-1]
    if ext == "obj":
        verts, faces, uvs, faces_uv = load_obj(path)
    else:
        verts, faces, uvs, _ = load_mesh(path)
        faces_uv = faces if uvs is not None else None
    verts = normalise_to_bbox(verts, scale)
    if uvs is not None:
        return verts, faces, uvs, faces_uv
    verts, faces = clean_mesh(verts, faces)
    # the reference's branches (uv_atlas.py:154-168): quadric decimation above max_faces, exactly TWO Loop subdivisions below min_faces, clean again
    if len(faces) > max_faces:
        verts, faces = decimate_qem(verts, faces, max_faces)
        verts, faces = clean_mesh(verts, faces)
    elif len(faces) < min_faces:
        verts, faces = subdivide_loop(verts, faces, iterations=2)
        verts, faces = clean_mesh(verts, faces)
    if unwrap == "grid":
        return unwrap_grid(verts, faces, atlas=atlas, gutter=gutter)
    if unwrap == "charts":
        # the atlas is computed on a COPY smoothed three times (uv_atlas.py:169-175) and carried by the unsmoothed mesh
        sm = smooth_simple(verts, faces, iterations=3)
        _, f2, uvs, f_uv = unwrap_charts(sm, faces, atlas=atlas, gutter=gutter, device=device)
        return verts.astype(np.float32), f2, uvs, f_uv
    raise ValueError("unknown unwrap method %r" % (unwrap,))
