#!/usr/bin/env python3
"""Ray-buffer entry point (rt_bvh_generate_contacts_rays): every strategy on every leg of bench.py's ray_buffer measurement.

    python tools/rays_sweep.py [--out profiles/r04_raybuffer_sweep.json] [--strategies lane:16,lane:64,p4,p4s:64,...]

One child process per strategy (RTMI_RAYS is read once per process); 'calibrated' = no override: the library's own first-buffer
calibration, whose choice is reported.  Every leg of every run is checked against the oracle's digests (bench.ray_buffer_leg).
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

DEFAULT = 'calibrated,lane:8,lane:16,lane:32,lane:64,lane:256,lane:1024,p4,p4s:32,p4s:64,p4s:256'


def child(with_atrium):
    import bench
    from openclraytracer_amd import capi
    fx = bench.bench_fixture()
    key = '%dx%d' % bench.RES
    ctx = capi.Context(0)
    out = {}
    scenes = [(n, bench.load_scene_bytes(n), fx['scenes'][n][key]['ray_buffer']) for n in bench.SCENES]
    if with_atrium:
        buf = bench.load_atrium_bytes()
        if bench.sha256_hex(__import__('numpy').frombuffer(buf, dtype='u1')) == fx['atrium']['scene_sha256']:
            scenes.append(('atrium', buf, fx['atrium'][key]['ray_buffer']))
    for name, buf, rec in scenes:
        scene = capi.DeviceScene(ctx, buf)
        bvh = capi.DeviceBVH(ctx, scene).build()
        legs = bench.ray_buffer_leg(ctx, bvh, bench.demo_camera(buf, *bench.RES), rec)
        out[name] = {k: dict(ms=v['ms'], mrays_per_s=v['mrays_per_s'], rays=v['rays'], strategy=v['strategy']) for k, v in legs.items()}
        bvh.close(); scene.close()
    ctx.close()
    print('RAYS_SWEEP ' + json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--child', action='store_true')
    ap.add_argument('--no-atrium', action='store_true')
    ap.add_argument('--strategies', default=DEFAULT)
    ap.add_argument('--out', default=None)
    a = ap.parse_args()
    if a.child:
        return child(not a.no_atrium)
    import bench
    if not a.no_atrium:
        bench.load_atrium_bytes()               # generated once, before the children
    res = {}
    for st in a.strategies.split(','):
        env = dict(os.environ)
        env.pop('RTMI_RAYS', None)
        if st != 'calibrated':
            env['RTMI_RAYS'] = st
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child'] + (['--no-atrium'] if a.no_atrium else []), cwd=ROOT, env=env,
                           capture_output=True, text=True, timeout=900)
        line = [l for l in r.stdout.splitlines() if l.startswith('RAYS_SWEEP ')]
        if r.returncode != 0 or not line:
            res[st] = dict(error=(r.stderr or r.stdout)[-600:])
            print('%-10s FAILED: %s' % (st, res[st]['error'][-300:]), flush=True)
            continue
        res[st] = json.loads(line[-1][len('RAYS_SWEEP '):])
        print('%-10s %s' % (st, '  '.join('%s %s' % (n, '/'.join('%.3f' % v[leg]['ms'] for leg in ('primary', 'reflect', 'shadow'))) for n, v in res[st].items())), flush=True)
    # per scene and leg: the fastest fixed strategy next to what the calibration chose
    summary = {}
    names = [n for n in next(v for v in res.values() if 'error' not in v)]
    for n in names:
        summary[n] = {}
        for leg in ('primary', 'reflect', 'shadow'):
            fixed = {st: v[n][leg]['ms'] for st, v in res.items() if 'error' not in v and st != 'calibrated'}
            best = min(fixed, key=fixed.get) if fixed else None
            cal = res.get('calibrated', {}).get(n)
            summary[n][leg] = dict(best_fixed=best, best_fixed_ms=fixed.get(best), calibrated_ms=cal[leg]['ms'] if cal else None,
                                   calibrated_strategy=cal[leg]['strategy'] if cal else None)
    out = dict(what='rt_bvh_generate_contacts_rays at 1920x1080, ms per call (HIP events), every strategy; all legs verified against the oracle digests',
               results=res, summary=summary)
    if a.out:
        json.dump(out, open(a.out, 'w'), indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == '__main__':
    main()
