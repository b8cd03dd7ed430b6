"""scene_pack.py -- TEST INFRASTRUCTURE ONLY (oracle side of the scene loader).

CPU restatement, in numpy/Python, of how the reference turns a scene text file
plus Wavefront OBJ/MTL files into its packed scene buffer:

* ``parse_obj``      restates tinyobj::LoadObj / LoadMtl v0.9.17 as vendored by the
                     reference (Include/3rdParty/tiny_obj_loader.h:266-383 float parser,
                     :421-452 parseTriple, :516-585 exportFaceGroupToShape, :587-813 LoadMtl,
                     :863-1135 LoadObj).  PINNED: tests compare it field by field with
                     oracle/_ref/tinyobj_dump, which is that very header compiled from
                     /root/reference (see oracle/ref_tinyobj_dump.cpp), and the dumps' SHA-256
                     are committed under tests/golden/.
* ``pack_scene``     restates Scene::load (src/CLRayTracerEngine/Scene.cpp:231-444) with
                     processMaterials (:129-167), materialEquals (CLData/Primitives/Material.h:69-81)
                     and the buffer layout of CLData/SceneBufferParser.h:55-160.
                     "parity unpinned": Scene.cpp needs <Windows.h> and Boost and cannot be built here.

Bytes the reference leaves uninitialised (the .w lane of CL_FLOAT3 temporaries copied into
the buffer: vertices, material colours, model/scene bounding boxes) are defined as 0 here.
"""
from __future__ import annotations

import math
import os
import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

FLT_MAX = np.float32(3.4028234663852886e38)
FLT_MIN = np.float32(1.1754943508222875e-38)
FLT_EPSILON = np.float32(1.1920928955078125e-07)

SCENE_HEADER_SIZE = 96
LIGHT_SIZE = 16
SPHERE_SIZE = 16
MATERIAL_SIZE = 96
MODEL_HEADER_SIZE = 64
MESH_HEADER_SIZE = 48
VERTEX_SIZE = 16
INDEX_SIZE = 2


# ----------------------------------------------------------------------------
# tinyobj restatement
# ----------------------------------------------------------------------------
def try_parse_double(tok: str) -> Tuple[bool, float]:
    """tryParseDouble, tiny_obj_loader.h:266-369 (NOT a correctly rounded strtod)."""
    n = len(tok)
    if n == 0:
        return False, 0.0
    i = 0
    mantissa = 0.0
    exponent = 0
    sign = '+'
    exp_sign = '+'
    c = tok[0]
    if c == '+' or c == '-':
        sign = c
        i = 1
    elif c.isdigit() and c.isascii():
        pass
    else:
        return False, 0.0
    read = 0
    end_not_reached = False
    while True:
        end_not_reached = i != n
        if not (end_not_reached and tok[i].isascii() and tok[i].isdigit()):
            break
        mantissa *= 10
        mantissa += ord(tok[i]) - 0x30
        i += 1
        read += 1
    if read == 0:
        return False, 0.0
    assemble = not end_not_reached
    if not assemble:
        if tok[i] == '.':
            i += 1
            read = 1
            while True:
                end_not_reached = i != n
                if not (end_not_reached and tok[i].isascii() and tok[i].isdigit()):
                    break
                mantissa += (ord(tok[i]) - 0x30) * math.pow(10.0, -read)
                read += 1
                i += 1
        elif tok[i] == 'e' or tok[i] == 'E':
            pass
        else:
            assemble = True
    if not assemble and not end_not_reached:
        assemble = True
    if not assemble and (tok[i] == 'e' or tok[i] == 'E'):
        i += 1
        end_not_reached = i != n
        if end_not_reached and (tok[i] == '+' or tok[i] == '-'):
            exp_sign = tok[i]
            i += 1
        elif i < n and tok[i].isascii() and tok[i].isdigit():
            pass
        else:
            return False, 0.0
        read = 0
        while True:
            end_not_reached = i != n
            if not (end_not_reached and tok[i].isascii() and tok[i].isdigit()):
                break
            exponent *= 10
            exponent += ord(tok[i]) - 0x30
            i += 1
            read += 1
        exponent *= (1 if exp_sign == '+' else -1)
        if read == 0:
            return False, 0.0
    val = (1 if sign == '+' else -1) * math.ldexp(mantissa * math.pow(5.0, exponent), exponent)
    return True, val


class _Tok:
    """Cursor over one line, mirroring the `const char*& token` idiom."""

    def __init__(self, s: str, pos: int = 0):
        self.s = s
        self.p = pos

    def skip_ws(self):  # token += strspn(token, " \t")
        s, p = self.s, self.p
        while p < len(s) and s[p] in ' \t':
            p += 1
        self.p = p

    def span_until(self, stops: str) -> str:  # strcspn
        s, p = self.s, self.p
        q = p
        while q < len(s) and s[q] not in stops:
            q += 1
        self.p = q
        return s[p:q]

    def peek(self) -> str:
        return self.s[self.p] if self.p < len(self.s) else '\0'


def _parse_float(t: _Tok) -> np.float32:
    """parseFloat, tiny_obj_loader.h:370-383."""
    t.skip_ws()
    tok = t.span_until(' \t\r')
    ok, val = try_parse_double(tok)
    return np.float32(val if ok else 0.0)


def _atoi(s: str) -> int:
    i = 0
    n = len(s)
    while i < n and s[i] in ' \t\n\v\f\r':
        i += 1
    sign = 1
    if i < n and s[i] in '+-':
        sign = -1 if s[i] == '-' else 1
        i += 1
    v = 0
    while i < n and s[i].isascii() and s[i].isdigit():
        v = v * 10 + (ord(s[i]) - 48)
        i += 1
    return sign * v


def _sscanf_s(s: str) -> str:
    """sscanf(token, "%s", buf): skip white space, read up to next white space."""
    i = 0
    while i < len(s) and s[i] in ' \t\n\v\f\r':
        i += 1
    j = i
    while j < len(s) and s[j] not in ' \t\n\v\f\r':
        j += 1
    return s[i:j]


def _fix_index(idx: int, n: int) -> int:  # tiny_obj_loader.h:215-221
    if idx > 0:
        return idx - 1
    if idx == 0:
        return 0
    return n + idx


def _parse_triple(t: _Tok, vsize: int, vnsize: int, vtsize: int) -> Tuple[int, int, int]:
    """parseTriple, tiny_obj_loader.h:421-452 -> (v, vt, vn)."""
    v = vt = vn = -1
    v = _fix_index(_atoi(t.s[t.p:]), vsize)
    t.span_until('/ \t\r')
    if t.peek() != '/':
        return (v, vt, vn)
    t.p += 1
    if t.peek() == '/':
        t.p += 1
        vn = _fix_index(_atoi(t.s[t.p:]), vnsize)
        t.span_until('/ \t\r')
        return (v, vt, vn)
    vt = _fix_index(_atoi(t.s[t.p:]), vtsize)
    t.span_until('/ \t\r')
    if t.peek() != '/':
        return (v, vt, vn)
    t.p += 1
    vn = _fix_index(_atoi(t.s[t.p:]), vnsize)
    t.span_until('/ \t\r')
    return (v, vt, vn)


@dataclass
class ObjMaterial:
    name: str = ""
    ambient: List[np.float32] = field(default_factory=lambda: [np.float32(0)] * 3)
    diffuse: List[np.float32] = field(default_factory=lambda: [np.float32(0)] * 3)
    specular: List[np.float32] = field(default_factory=lambda: [np.float32(0)] * 3)
    transmittance: List[np.float32] = field(default_factory=lambda: [np.float32(0)] * 3)
    emission: List[np.float32] = field(default_factory=lambda: [np.float32(0)] * 3)
    shininess: np.float32 = np.float32(1.0)
    ior: np.float32 = np.float32(1.0)
    dissolve: np.float32 = np.float32(1.0)
    illum: int = 0

    def as_floats(self) -> np.ndarray:
        return np.array(list(self.ambient) + list(self.diffuse) + list(self.specular)
                        + list(self.transmittance) + list(self.emission)
                        + [self.shininess, self.ior, self.dissolve, np.float32(self.illum)], dtype=np.float32)


@dataclass
class ObjShape:
    positions: np.ndarray   # float32 [3*nv]
    indices: np.ndarray     # uint32 [3*ntri]
    material_ids: np.ndarray  # int32 [ntri]
    name: str = ""


def _read_lines(path: str) -> List[str]:
    # std::istream::getline on a text file; '\r' is trimmed by the caller logic below.
    with open(path, 'rb') as f:
        data = f.read()
    lines = data.decode('latin-1').split('\n')
    if lines and lines[-1] == '':
        lines.pop()
    return lines


def _trim(line: str) -> str:
    if line.endswith('\n'):
        line = line[:-1]
    if line.endswith('\r'):
        line = line[:-1]
    return line


def load_mtl(path: str, material_map: Dict[str, int], materials: List[ObjMaterial]) -> None:
    """LoadMtl, tiny_obj_loader.h:587-813 (+ MaterialFileReader :815-836)."""
    material = ObjMaterial()
    lines = _read_lines(path) if os.path.exists(path) else []
    for raw in lines:
        line = _trim(raw)
        if not line:
            continue
        t = _Tok(line)
        t.skip_ws()
        s = line[t.p:]
        if not s or s[0] == '#':
            continue

        def sp(k):  # isSpace(token[k])
            return len(s) > k and s[k] in ' \t'

        if s.startswith('newmtl') and sp(6):
            if material.name:
                material_map.setdefault(material.name, len(materials))
                materials.append(material)
            material = ObjMaterial()
            material.name = _sscanf_s(s[7:])
            continue
        two = s[:2]
        if two in ('Ka', 'Kd', 'Ks', 'Kt', 'Ke') and sp(2):
            tt = _Tok(s, 2)
            rgb = [_parse_float(tt), _parse_float(tt), _parse_float(tt)]
            attr = {'Ka': 'ambient', 'Kd': 'diffuse', 'Ks': 'specular', 'Kt': 'transmittance', 'Ke': 'emission'}[two]
            setattr(material, attr, rgb)
            continue
        if two == 'Ni' and sp(2):
            material.ior = _parse_float(_Tok(s, 2))
            continue
        if two == 'Ns' and sp(2):
            material.shininess = _parse_float(_Tok(s, 2))
            continue
        if s.startswith('illum') and sp(5):
            tt = _Tok(s, 6)
            tt.skip_ws()
            material.illum = _atoi(s[tt.p:])
            continue
        if s[0] == 'd' and sp(1):
            material.dissolve = _parse_float(_Tok(s, 1))
            continue
        if two == 'Tr' and sp(2):
            material.dissolve = np.float32(np.float32(1.0) - _parse_float(_Tok(s, 2)))
            continue
        # texture names / unknown parameters carry no data the path uses
    material_map.setdefault(material.name, len(materials))
    materials.append(material)


def parse_obj(path: str, mtl_base: Optional[str] = None) -> Tuple[List[ObjShape], List[ObjMaterial]]:
    """tinyobj::LoadObj(shapes, materials, err, filename, mtl_basepath, triangulate=true)."""
    if mtl_base is None:
        mtl_base = mtl_base_path(path)
    v: List[np.float32] = []
    vn_count = 0
    vt_count = 0
    face_group: List[List[Tuple[int, int, int]]] = []
    shapes: List[ObjShape] = []
    materials: List[ObjMaterial] = []
    material_map: Dict[str, int] = {}
    material = -1
    name = ""

    def flush():
        # exportFaceGroupToShape, :516-585; vertexCache is passed BY VALUE so it starts empty every call
        nonlocal face_group
        if not face_group:
            return
        cache: Dict[Tuple[int, int, int], int] = {}
        pos: List[np.float32] = []
        idx: List[int] = []
        mids: List[int] = []

        def upd(vi):
            k = cache.get(vi)
            if k is not None:
                return k
            b = 3 * vi[0]
            pos.extend((v[b], v[b + 1], v[b + 2]))
            k = len(pos) // 3 - 1
            cache[vi] = k
            return k

        for face in face_group:
            i0 = face[0]
            i2 = face[1]
            for k in range(2, len(face)):
                i1 = i2
                i2 = face[k]
                a = upd(i0)
                b = upd(i1)
                c = upd(i2)
                idx.extend((a, b, c))
                mids.append(material)
        shapes.append(ObjShape(np.array(pos, dtype=np.float32), np.array(idx, dtype=np.uint32),
                               np.array(mids, dtype=np.int32), name))

    for raw in _read_lines(path):
        line = _trim(raw)
        if not line:
            continue
        t = _Tok(line)
        t.skip_ws()
        s = line[t.p:]
        if not s or s[0] == '#':
            continue
        c0 = s[0]
        c1 = s[1] if len(s) > 1 else '\0'
        c2 = s[2] if len(s) > 2 else '\0'
        if c0 == 'v' and c1 in ' \t':
            tt = _Tok(s, 2)
            v.extend((_parse_float(tt), _parse_float(tt), _parse_float(tt)))
            continue
        if c0 == 'v' and c1 == 'n' and c2 in ' \t':
            vn_count += 1
            continue
        if c0 == 'v' and c1 == 't' and c2 in ' \t':
            vt_count += 1
            continue
        if c0 == 'f' and c1 in ' \t':
            tt = _Tok(s, 2)
            tt.skip_ws()
            face = []
            while tt.peek() not in '\r\n\0':
                face.append(_parse_triple(tt, len(v) // 3, vn_count, vt_count))
                while tt.p < len(s) and s[tt.p] in ' \t\r':
                    tt.p += 1
            face_group.append(face)
            continue
        if s.startswith('usemtl') and len(s) > 6 and s[6] in ' \t':
            namebuf = _sscanf_s(s[7:])
            flush()
            face_group = []
            material = material_map.get(namebuf, -1)
            continue
        if s.startswith('mtllib') and len(s) > 6 and s[6] in ' \t':
            namebuf = _sscanf_s(s[7:])
            load_mtl((mtl_base + namebuf) if mtl_base else namebuf, material_map, materials)
            continue
        if c0 == 'g' and c1 in ' \t':
            flush()
            face_group = []
            names = []
            tt = _Tok(s, 0)
            while tt.peek() not in '\r\n\0':
                tt.skip_ws()
                names.append(tt.span_until(' \t\r'))
                while tt.p < len(s) and s[tt.p] in ' \t\r':
                    tt.p += 1
            name = names[1] if len(names) > 1 else ""
            continue
        if c0 == 'o' and c1 in ' \t':
            flush()
            face_group = []
            name = _sscanf_s(s[2:])
            continue
    flush()
    return shapes, materials


def mtl_base_path(file_name: str) -> str:
    """loadMesh, Scene.cpp:58-68: directory prefix including the trailing separator."""
    s = file_name.rfind('\\')
    r = file_name.rfind('/')
    if not (s == -1 or r == -1):
        k = max(s, r)
    else:
        # min over size_t with npos == max  -> the one that is not npos (or npos)
        cands = [x for x in (s, r) if x != -1]
        k = min(cands) if cands else -1
    return file_name[:k + 1]


def load_tinyobj_dump(path: str) -> Tuple[List[ObjShape], List[ObjMaterial]]:
    """Reads the dump written by oracle/_ref/tinyobj_dump (the REFERENCE's parser)."""
    with open(path, 'rb') as f:
        data = f.read()
    assert data[:6] == b'TOBJ1\n'
    off = 6
    nshapes, nmat = struct.unpack_from('<II', data, off)
    off += 8
    materials = []
    for _ in range(nmat):
        fl = np.frombuffer(data, dtype='<f4', count=19, offset=off)
        off += 76
        m = ObjMaterial()
        m.ambient = list(fl[0:3]); m.diffuse = list(fl[3:6]); m.specular = list(fl[6:9])
        m.transmittance = list(fl[9:12]); m.emission = list(fl[12:15])
        m.shininess, m.ior, m.dissolve = fl[15], fl[16], fl[17]
        m.illum = int(fl[18])
        materials.append(m)
    shapes = []
    for _ in range(nshapes):
        npos, nidx, nmid = struct.unpack_from('<III', data, off)
        off += 12
        pos = np.frombuffer(data, dtype='<f4', count=npos, offset=off).copy(); off += 4 * npos
        idx = np.frombuffer(data, dtype='<u4', count=nidx, offset=off).copy(); off += 4 * nidx
        mid = np.frombuffer(data, dtype='<i4', count=nmid, offset=off).copy(); off += 4 * nmid
        shapes.append(ObjShape(pos, idx, mid))
    assert off == len(data)
    return shapes, materials


# ----------------------------------------------------------------------------
# Scene::load restatement
# ----------------------------------------------------------------------------
def _float_equals(a: np.float32, b: np.float32) -> bool:
    """floatEquals, RTKernelUtils.h:125-129."""
    diff = np.float32(a) - np.float32(b)
    return bool(diff <= FLT_EPSILON and diff >= -FLT_EPSILON)


def _material_equals(a: np.ndarray, b: np.ndarray) -> bool:
    """materialEquals, Material.h:69-81 over the 19 packed floats."""
    return all(_float_equals(x, y) for x, y in zip(a, b))


def _parse_number(val: str) -> Optional[np.float32]:
    """parseNumber, Scene.cpp:169-182 (atof, then reject non-numeric strings that gave 0)."""
    i = 0
    n = len(val)
    while i < n and val[i] in ' \t\n\v\f\r':
        i += 1
    j = i
    if j < n and val[j] in '+-':
        j += 1
    k = j
    while k < n and val[k].isdigit():
        k += 1
    if k < n and val[k] == '.':
        k += 1
        while k < n and val[k].isdigit():
            k += 1
    digits = k > j and any(ch.isdigit() for ch in val[j:k])
    if digits and k < n and val[k] in 'eE':
        e = k + 1
        if e < n and val[e] in '+-':
            e += 1
        if e < n and val[e].isdigit():
            while e < n and val[e].isdigit():
                e += 1
            k = e
    f = float(val[i:k]) if digits else 0.0
    if f == 0.0:
        for ch in val:
            if ch != '0' and ch != '.':
                return None
    return np.float32(f)


def _string_to_floats(s: str, size: int) -> List[np.float32]:
    """stringToFloats, Scene.cpp:184-200: split on ' ' (token_compress_on), first `size` tokens."""
    toks: List[str] = []
    cur = ''
    prev_sep = False
    for ch in s:
        if ch == ' ':
            if not prev_sep:
                toks.append(cur)
                cur = ''
            prev_sep = True
        else:
            cur += ch
            prev_sep = False
    toks.append(cur)
    out = []
    for tok in toks[:size]:
        v = _parse_number(tok)
        if v is not None:
            out.append(v)
    return out


def parse_scene_file(path: str) -> Dict[str, List[str]]:
    """Scene::load, Scene.cpp:233-261: first word is the key (std::map => sorted keys)."""
    raw: Dict[str, List[str]] = {}
    with open(path, 'rb') as f:
        text = f.read().decode('latin-1')
    lines = text.split('\n')
    if lines and lines[-1] == '':
        lines.pop()
    for line in lines:
        if line.endswith('\r'):          # Windows text-mode ifstream folds CRLF
            line = line[:-1]
        sp = line.find(' ')
        if sp == -1:
            continue
        raw.setdefault(line[:sp], []).append(line[sp + 1:])
    return raw


def resolve_mesh_path(mesh: str, scene_path: str) -> str:
    p = mesh.replace('\\', '/')
    if os.path.exists(p):
        return p
    q = os.path.join(os.path.dirname(os.path.abspath(scene_path)), p)
    return q if os.path.exists(q) else p


def pack_scene(raw: Dict[str, List[str]], models: List[Tuple[List[ObjShape], List[ObjMaterial]]]) -> bytes:
    """Scene::load, Scene.cpp:263-444.  `models` holds one (shapes, materials) per MESH line."""
    unique: List[np.ndarray] = []
    size = SCENE_HEADER_SIZE
    total_prims = 0
    n_lights = n_spheres = n_mesh = 0
    total_model_size = 0
    model_sizes: List[int] = []
    remapped: List[List[np.ndarray]] = []
    for key in sorted(raw.keys()):
        if key == 'LIGHT':
            n_lights = len(raw[key])
            size += n_lights * LIGHT_SIZE
        elif key == 'SPHERE':
            n_spheres = len(raw[key])
            size += n_spheres * SPHERE_SIZE
        elif key == 'MESH':
            n_mesh = len(raw[key])
            for shapes, mats in models:
                msz = MODEL_HEADER_SIZE
                for sh in shapes:
                    msz += len(sh.indices) * INDEX_SIZE + (len(sh.positions) // 3) * VERTEX_SIZE + MESH_HEADER_SIZE
                model_sizes.append(msz)
                total_model_size += msz
                # processMaterials, Scene.cpp:129-167
                index_transform: Dict[int, int] = {}
                for i, m in enumerate(mats):
                    mf = m.as_floats()
                    found = None
                    for j, u in enumerate(unique):
                        if _material_equals(mf, u):
                            found = j
                            break
                    if found is None:
                        found = len(unique)
                        unique.append(mf)
                    index_transform[i] = found
                ids = []
                for sh in shapes:
                    # CL_UINT originalIndex = material_ids[j]; indexTransform[originalIndex] (missing key -> 0)
                    ids.append(np.array([index_transform.get(int(np.uint32(x)), 0) for x in sh.material_ids],
                                        dtype=np.uint32))
                remapped.append(ids)
            size += total_model_size
        else:
            continue
        total_prims += len(raw[key])
        size += len(unique) * MATERIAL_SIZE   # inside the key loop, as written (Scene.cpp:305)

    buf = bytearray(size)
    struct.pack_into('<8Q', buf, 0, size, total_prims, n_lights, n_spheres, len(unique),
                     total_model_size, len(models), 0)
    lights_off = SCENE_HEADER_SIZE
    spheres_off = lights_off + n_lights * LIGHT_SIZE
    mats_off = spheres_off + n_spheres * SPHERE_SIZE
    models_off = mats_off + len(unique) * MATERIAL_SIZE
    for i, s in enumerate(raw.get('LIGHT', [])):
        vals = _string_to_floats(s, 4)
        if len(vals) != 4:
            continue
        struct.pack_into('<4f', buf, lights_off + i * LIGHT_SIZE, *[float(x) for x in vals])
    for i, s in enumerate(raw.get('SPHERE', [])):
        vals = _string_to_floats(s, 4)
        if len(vals) != 4:
            continue
        struct.pack_into('<4f', buf, spheres_off + i * SPHERE_SIZE, *[float(x) for x in vals])
    for i, u in enumerate(unique):
        o = mats_off + i * MATERIAL_SIZE
        for k in range(5):
            struct.pack_into('<3f', buf, o + 16 * k, *[float(x) for x in u[3 * k:3 * k + 3]])
        struct.pack_into('<4f', buf, o + 80, *[float(x) for x in u[15:19]])

    scene_lo = np.array([FLT_MAX] * 3, dtype=np.float32)
    scene_hi = np.array([FLT_MIN] * 3, dtype=np.float32)
    scene_tris = 0
    off = models_off
    for mi, (shapes, _m) in enumerate(models):
        mbase = off
        lo = np.array([FLT_MAX] * 3, dtype=np.float32)
        hi = np.array([FLT_MIN] * 3, dtype=np.float32)
        model_tris = 0
        moff = mbase + MODEL_HEADER_SIZE
        for si, sh in enumerate(shapes):
            nv = len(sh.positions) // 3
            ni = len(sh.indices)
            msize = ni * INDEX_SIZE + nv * VERTEX_SIZE + MESH_HEADER_SIZE
            mat_index = int(remapped[mi][si][0])
            struct.pack_into('<6Q', buf, moff, msize, ni // 3, nv, ni, mat_index, 0)
            model_tris += ni // 3
            if nv:
                verts = np.zeros((nv, 4), dtype='<f4')
                verts[:, :3] = sh.positions.reshape(nv, 3)
                buf[moff + MESH_HEADER_SIZE: moff + MESH_HEADER_SIZE + nv * VERTEX_SIZE] = verts.tobytes()
                lo = _macro_min_fold(lo, sh.positions.reshape(nv, 3))
                hi = _macro_max_fold(hi, sh.positions.reshape(nv, 3))
            ioff = moff + MESH_HEADER_SIZE + nv * VERTEX_SIZE
            buf[ioff: ioff + ni * INDEX_SIZE] = sh.indices.astype('<u2').tobytes()   # ushort truncation (Scene.cpp:430)
            # hdr->boundingBox = (minBounds,maxBounds); scene bbox = merge(model, scene)   (Scene.cpp:432-434)
            scene_lo = np.where(lo < scene_lo, lo, scene_lo)       # MIN4(model, scene) = a < b ? a : b
            scene_hi = np.where(hi > scene_hi, hi, scene_hi)
            moff += msize
        struct.pack_into('<4Q', buf, mbase, model_sizes[mi], len(shapes), model_tris, 0)
        struct.pack_into('<4f', buf, mbase + 32, float(lo[0]), float(lo[1]), float(lo[2]), 0.0)
        struct.pack_into('<4f', buf, mbase + 48, float(hi[0]), float(hi[1]), float(hi[2]), 0.0)
        scene_tris += model_tris
        off = mbase + model_sizes[mi]
    struct.pack_into('<Q', buf, 56, scene_tris)
    struct.pack_into('<4f', buf, 64, float(scene_lo[0]), float(scene_lo[1]), float(scene_lo[2]), 0.0)
    struct.pack_into('<4f', buf, 80, float(scene_hi[0]), float(scene_hi[1]), float(scene_hi[2]), 0.0)
    return bytes(buf)


def _macro_min_fold(acc: np.ndarray, pts: np.ndarray) -> np.ndarray:
    """acc = min(acc, v) for every vertex in order with min(a,b) = a < b ? a : b (CLPortability.h:147):
    among equal values (+0 / -0) the LAST one met wins."""
    out = acc.copy()
    for k in range(3):
        col = pts[:, k]
        m = col.min()
        if m < out[k] or not (out[k] < m):          # macro takes v whenever !(acc < v)
            eq = np.nonzero(col == m)[0]
            cand = col[eq[-1]]
            # acc survives only if acc < every later value; it equals m only when acc == m came first
            out[k] = cand if not (out[k] < m) else out[k]
    return out.astype(np.float32)


def _macro_max_fold(acc: np.ndarray, pts: np.ndarray) -> np.ndarray:
    out = acc.copy()
    for k in range(3):
        col = pts[:, k]
        m = col.max()
        if not (out[k] > m):
            eq = np.nonzero(col == m)[0]
            out[k] = col[eq[-1]]
    return out.astype(np.float32)


def load_scene(scene_path: str, obj_loader=parse_obj) -> bytes:
    """Scene::load end to end with the given OBJ loader (own restatement by default)."""
    raw = parse_scene_file(scene_path)
    models = []
    for mesh in raw.get('MESH', []):
        p = resolve_mesh_path(mesh.rstrip('\r'), scene_path)
        models.append(obj_loader(p))
    return pack_scene(raw, models)


def scene_header(buf: bytes) -> dict:
    f = struct.unpack_from('<8Q8f', buf, 0)
    keys = ['totalDataSize', 'numberOfPrimitives', 'numberOfLights', 'numberOfSpheres', 'numberOfMaterials',
            'modelBufferSize', 'numberOfModels', 'totalNumberOfTriangles']
    d = dict(zip(keys, f[:8]))
    d['bbox_min'] = f[8:11]
    d['bbox_max'] = f[12:15]
    return d
