"""The grid walk's three code paths against each other over many cameras (no oracle: the CPU restatement takes seconds per full-size frame of the
atrium; the parity tests compare each path with it at test sizes).  For `--poses` random poses inside and around the scene -- positions anywhere in
the box, orientations anywhere on the sphere, so that waves of every direction octant, of mixed signs and of axis-parallel rows occur -- the contacts of
    (a) the library as shipped: packet walk where a wave qualifies, else lane by lane on the walk's records,
    (b) RTMI_TLG_TRICAM=0:       every wave lane by lane on the walk's records and the scene's triangle table,
    (c) RTMI_TLG_WALK_RECORDS=0: the walk of round 4 with the reference's divisions
are hashed frame by frame; the three lists of digests must be equal.  Each variant is a process of its own (the knobs are read by rt_ctx_create).
    python tools/tlg_soak.py [--scene synth_atrium] [--res 960x540] [--poses 48] > profiles/r05_tlg_soak.txt"""
import argparse, hashlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'assets', 'synth'))


def child(packed, res, poses):
    import numpy as np
    import bench
    from openclraytracer_amd import capi
    buf = open(packed, 'rb').read()
    ctx = capi.Context(0)
    sc = capi.DeviceScene(ctx, buf)
    g = capi.DeviceTLG(ctx, sc).build()
    base = bench.demo_camera(buf, *res)
    f = np.frombuffer(buf, dtype='<f4', count=8, offset=64)
    lo, hi = f[0:3].astype(np.float64), f[4:7].astype(np.float64)
    rng = np.random.default_rng(20261005)
    out = []
    npx = res[0] * res[1]
    dc = ctx.alloc(npx * 32)
    hits = 0
    for k in range(poses):
        cam = base.copy()
        m = cam['viewTransform'][0]
        if k:
            q = rng.normal(size=4); q /= np.linalg.norm(q)
            w, x, y, z = q
            rot = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                            [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
            if k % 7 == 3:
                rot = np.eye(3)[[(k // 7) % 3, (k // 7 + 1) % 3, (k // 7 + 2) % 3]] * np.array([1.0, -1.0, 1.0])[:, None]     # axis-parallel frames: rows / columns of zero components
            pos = lo + (hi - lo) * rng.uniform(-0.1, 1.1, size=3)
            for r in range(3):
                m[4 * r:4 * r + 3] = rot[r].astype(np.float32)
                m[4 * r + 3] = np.float32(pos[r])
        g.generate_contacts_primary_d(cam, dc)
        ctx.synchronize()
        c = ctx.download(dc, npx, capi.CONTACT)
        hits += int((c['normalAndDistance'][:, 3] > 0).sum())
        out.append(hashlib.sha256(c.tobytes()).hexdigest())
    print('RESULT ' + json.dumps(dict(walk_records=bool(g.walk_records()), hits=hits, digests=out)))


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--scene', default='synth_atrium'); ap.add_argument('--res', default='960x540'); ap.add_argument('--poses', type=int, default=48)
    ap.add_argument('--child', default=None)
    a = ap.parse_args()
    res = tuple(int(v) for v in a.res.split('x'))
    if a.child:
        child(a.child, res, a.poses)
        sys.exit(0)
    import gen_scenes
    txt = gen_scenes.ensure()[a.scene] if a.scene.startswith('synth') else os.path.join(ROOT, 'assets', 'scenes', a.scene + '.txt')
    packed = '/tmp/_soak_%s_%d.bin' % (a.scene, os.getpid())
    subprocess.check_call([os.path.join(ROOT, 'apps', 'bin', 'scene_pack_tool'), os.path.relpath(txt, ROOT), packed], cwd=ROOT)
    got = {}
    for name, env in (('as shipped (packet walk)', {}), ('RTMI_TLG_TRICAM=0 (lane by lane, records)', {'RTMI_TLG_TRICAM': '0'}), ('RTMI_TLG_WALK_RECORDS=0 (the reference\'s divisions)', {'RTMI_TLG_WALK_RECORDS': '0'})):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', packed, '--res', a.res, '--poses', str(a.poses)], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
        if r.returncode:
            raise SystemExit('variant %s failed:\n%s' % (name, r.stderr[-3000:]))
        got[name] = json.loads([l for l in r.stdout.splitlines() if l.startswith('RESULT ')][0][7:])
    os.unlink(packed)
    names = list(got)
    print('grid walk soak: %s, %s, %d poses' % (a.scene, a.res, a.poses))
    for n in names:
        print('  %-55s walk records %s, %d hits over all frames, digest of digests %s' % (n, got[n]['walk_records'], got[n]['hits'], hashlib.sha256(''.join(got[n]['digests']).encode()).hexdigest()[:16]))
    bad = [k for k in range(a.poses) if len({got[n]['digests'][k] for n in names}) != 1]
    print('  frames that differ between the variants: %s' % (bad if bad else 'none'))
    sys.exit(1 if bad else 0)
