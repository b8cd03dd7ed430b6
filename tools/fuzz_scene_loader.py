#!/usr/bin/env python3
"""Mutation fuzz of the scene / OBJ / MTL parsers and the packer (Scene::load, Scene.cpp:231-444 grammar; the vendored tinyobj's) through
scene_pack_tool -- meant for the sanitizer build (tools/san): untrusted text must end in a packed buffer or an error message, never in a
sanitizer report, a crash or a hang.

    python tools/fuzz_scene_loader.py --cases 10000 --tool tools/san/_build/scene_pack_tool [--seed 1] [--jobs 8]

A case = a scene file + its OBJ + its MTL (seeds: the bundled Cube and Gardenlight scenes and two hand-written edge-case scenes), one of
them mutated: bytes flipped / inserted / deleted, lines deleted / duplicated / swapped, numbers replaced by extremes (huge indices,
negative indices, nan, 1e999, empty), keywords truncated, the file truncated.  Failure = exit code other than 0 / 1 / 2, a signal, a
timeout, or a sanitizer report on stderr.  Failing cases are kept under /tmp/rtmi_fuzz_failures/.
"""
import argparse
import concurrent.futures as cf
import os
import random
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

EDGE_OBJ = b"""# edge cases
mtllib e.mtl
o first
v 0 0 0
v 1 0 0
v 0 1 0
v 1 1 0
vn 0 0 1
vt 0 0
usemtl a
f 1/1/1 2/1/1 3/1/1
f -1 -2 -3
f 1//1 2//1 4//1 3//1
g second
usemtl b
f 1 2 3 4
"""
EDGE_MTL = b"""newmtl a
Ka 0.1 0.1 0.1
Kd 0.5 0.4 0.3
Ks 1 1 1
Ns 10
newmtl b
Kd 1 0 0
d 0.5
illum 2
"""
EXTREMES = [b'0', b'-1', b'-0', b'65535', b'65536', b'65537', b'4294967295', b'4294967296', b'-4294967296', b'99999999999999999999', b'1e999', b'-1e999',
            b'nan', b'inf', b'1e-999', b'', b'.', b'-', b'+', b'1e', b'0x10', b'1/2/3/4', b'//', b'1//', b'/1', b'1e+', b'3.4028235e38', b'1.17549435e-38']
KEYWORDS = [b'LIGHT', b'MESH', b'SPHERE', b'v ', b'vn ', b'vt ', b'f ', b'g ', b'o ', b'usemtl ', b'mtllib ', b'newmtl ', b'Ka ', b'Kd ', b'Ks ', b'Ns ', b'd ']


def seeds():
    out = []
    for scene in ('cube', 'gardenlight'):
        txt = open(os.path.join(ROOT, 'assets', 'scenes', scene + '.txt'), 'rb').read()
        mesh = [l for l in txt.split(b'\n') if l.startswith(b'MESH')][0].split()[1].decode()
        obj = open(os.path.join(ROOT, mesh), 'rb').read()
        mtl_name = [l for l in obj.split(b'\n') if l.startswith(b'mtllib')]
        mtl = b''
        if mtl_name:
            mp = os.path.join(os.path.dirname(os.path.join(ROOT, mesh)), mtl_name[0].split()[1].decode())
            if os.path.exists(mp):
                mtl = open(mp, 'rb').read()
        out.append((txt, obj, mtl))
    out.append((b'LIGHT 0.0 10.0 -330.0 1000.0\nLIGHT 1 2 3\nSPHERE 0 0 0 1 0\nMESH m.obj\nMESH m.obj', EDGE_OBJ, EDGE_MTL))
    out.append((b'MESH m.obj', b'v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n', b''))
    return out


def mutate(data: bytes, rng: random.Random) -> bytes:
    b = bytearray(data)
    for _ in range(rng.choice((1, 1, 1, 2, 3, 8))):
        kind = rng.randrange(11)
        if kind == 0 and b:                                   # flip a byte
            b[rng.randrange(len(b))] = rng.randrange(256)
        elif kind == 1:                                       # insert bytes
            at = rng.randrange(len(b) + 1)
            b[at:at] = bytes(rng.randrange(256) for _ in range(rng.choice((1, 2, 7, 64))))
        elif kind == 2 and b:                                 # delete a span
            at = rng.randrange(len(b))
            del b[at:at + rng.choice((1, 3, 16, 200))]
        elif kind in (3, 4, 5):                               # line surgery
            lines = bytes(b).split(b'\n')
            if lines:
                i = rng.randrange(len(lines))
                if kind == 3:
                    del lines[i]
                elif kind == 4:
                    lines[i:i] = [lines[i]] * rng.choice((1, 3, 50))
                else:
                    j = rng.randrange(len(lines)); lines[i], lines[j] = lines[j], lines[i]
            b = bytearray(b'\n'.join(lines))
        elif kind in (6, 7):                                  # a number becomes an extreme
            toks = bytes(b).split(b' ')
            if len(toks) > 1:
                toks[rng.randrange(1, len(toks))] = rng.choice(EXTREMES)
            b = bytearray(b' '.join(toks))
        elif kind == 8 and b:                                 # truncate
            del b[rng.randrange(len(b)):]
        elif kind == 9:                                       # a keyword in a random place
            at = rng.randrange(len(b) + 1)
            b[at:at] = b'\n' + rng.choice(KEYWORDS)
        elif kind == 10:                                      # line endings / whitespace
            b = bytearray(bytes(b).replace(b'\n', rng.choice((b'\r\n', b'\r', b'\n\n', b' \n\t'))))
    return bytes(b)


def run_case(args):
    tool, case, seed, base = args
    rng = random.Random(seed * 1000003 + case)
    txt, obj, mtl = base[rng.randrange(len(base))]
    which = rng.randrange(4)
    d = tempfile.mkdtemp(prefix='rtmi_fuzz_')
    try:
        # the scene file names its mesh relative to the working directory; the OBJ names its MTL relative to itself
        txt = b'\n'.join((b'MESH m.obj' if l.startswith(b'MESH') else l) for l in txt.split(b'\n'))
        obj = b'\n'.join((b'mtllib m.mtl' if l.startswith(b'mtllib') else l) for l in obj.split(b'\n'))
        if which == 0: txt = mutate(txt, rng)
        elif which == 1 or which == 3: obj = mutate(obj, rng)
        else: mtl = mutate(mtl, rng)
        open(os.path.join(d, 's.txt'), 'wb').write(txt)
        open(os.path.join(d, 'm.obj'), 'wb').write(obj)
        open(os.path.join(d, 'm.mtl'), 'wb').write(mtl)
        try:
            r = subprocess.run([tool, 's.txt', 'out.bin'] + (['--index32'] if rng.randrange(5) == 0 else []), cwd=d, capture_output=True, timeout=60)
        except subprocess.TimeoutExpired:
            return case, 'timeout', d
        err = r.stderr[-4000:]
        bad = r.returncode not in (0, 1, 2) or b'Sanitizer' in err or b'runtime error' in err
        if bad:
            return case, 'exit %d: %s' % (r.returncode, err.decode('latin1')[-1500:]), d
        shutil.rmtree(d, ignore_errors=True)
        return case, None, None
    except Exception as e:          # noqa: BLE001
        return case, 'harness: %r' % (e,), d


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', type=int, default=10000)
    ap.add_argument('--seed', type=int, default=1)
    ap.add_argument('--jobs', type=int, default=min(8, os.cpu_count() or 1))
    ap.add_argument('--tool', default=os.path.join(ROOT, 'apps', 'bin', 'scene_pack_tool'))
    a = ap.parse_args()
    tool = os.path.abspath(a.tool)
    base = seeds()
    failures, ok, errs = [], 0, 0
    keep = '/tmp/rtmi_fuzz_failures'
    with cf.ThreadPoolExecutor(max_workers=a.jobs) as ex:
        for case, why, d in ex.map(run_case, [(tool, c, a.seed, base) for c in range(a.cases)]):
            if why is None:
                ok += 1
                continue
            failures.append((case, why))
            os.makedirs(keep, exist_ok=True)
            if d and len(failures) <= 20:
                shutil.move(d, os.path.join(keep, 'case_%d' % case))
            elif d:
                shutil.rmtree(d, ignore_errors=True)
    print('scene-loader fuzz: %d cases (seed %d) through %s: %d clean, %d FAILED' % (a.cases, a.seed, os.path.relpath(tool, ROOT), ok, len(failures)))
    for case, why in failures[:10]:
        print('  case %d: %s' % (case, why.replace('\n', '\n      ')))
    sys.exit(1 if failures else 0)


if __name__ == '__main__':
    main()
