"""BASELINE.json configs[3] on one GPU: Two-Level Grid build (densities 2.0 / 2.0) + trace + shade of the atrium-class
scene at 1920x1080.  Prints one JSON object with HIP-event times and the algorithmic-byte rate of the trace kernel
(16 B per top-level cell visited + 8 B per leaf cell + 48 B per triangle test + 32 B contact, counts from the oracle
fixture tests/golden/tlg_frames.json).  Used under rocprofv3 by tools/profile_round.sh."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'assets', 'synth'))
import gen_scenes
import bench
from openclraytracer_amd import capi

name, res = 'synth_atrium', (1920, 1080)
txt = gen_scenes.ensure()[name]
tmp = '/tmp/_%s.bin' % name
if os.environ.get('RTMI_PACKED') == '1':      # under rocprofv3 the scene comes packed and NO child process is started (a process whose GPU is initialised must not spawn)
    if not os.path.exists(tmp):
        raise SystemExit('RTMI_PACKED=1 but %s is missing: pack the scenes before the profiled pass (tools/profile_round.sh)' % tmp)
else:
    subprocess.check_call([os.path.join(ROOT, 'apps', 'bin', 'scene_pack_tool'), os.path.relpath(txt, ROOT), tmp], cwd=ROOT)
buf = open(tmp, 'rb').read()
ctx = capi.Context(0)
sc = capi.DeviceScene(ctx, buf)
g = capi.DeviceTLG(ctx, sc)
tm = capi.Timer(ctx)
g.build(); ctx.synchronize()
B = 5
tm.start()
for _ in range(B):
    g.build()
tm.stop(); build_ms = tm.elapsed_ms() / B
cam = bench.demo_camera(buf, *res)
npx = res[0] * res[1]
dc, dr = ctx.alloc(npx * 32), ctx.alloc(npx * 16)
g.generate_contacts_primary_d(cam, dc); ctx.synchronize()
K = 10
tm.start()
for _ in range(K):
    g.generate_contacts_primary_d(cam, dc)
tm.stop(); trace_ms = tm.elapsed_ms() / K
tm.start()
for _ in range(K):
    g.generate_contacts_primary_d(cam, dc)
    capi.shade_blinn_phong_d(ctx, sc, cam, dc, npx, dr)
tm.stop(); two_ms = tm.elapsed_ms() / K
g.generate_frame_d(cam, dc, dr); ctx.synchronize()
tm.start()
for _ in range(K):
    g.generate_frame_d(cam, dc, dr)          # rt_tlg_generate_frame: contacts + colours in one launch
tm.stop(); frame_ms = tm.elapsed_ms() / K
# the frame that was timed, against the oracle's digest (tests/golden/bench_frames.json; skipped when this host packed other scene bytes)
import hashlib, numpy as np
bfx = bench.bench_fixture()['atrium']
verified = None
if hashlib.sha256(buf).hexdigest() == bfx['scene_sha256']:
    got = hashlib.sha256(ctx.download(dc, npx * 32, np.uint8).tobytes()).hexdigest()
    if got != bfx['%dx%d' % res]['tlg']['contacts_sha256']:
        raise SystemExit('VERIFICATION FAILED: Two-Level-Grid contacts differ from the oracle\'s')
    verified = 'contacts equal the oracle\'s'
fx = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'tlg_frames.json')))[name]['%dx%d' % res]
alg = 16 * fx['top_visits'] + 8 * fx['leaf_visits'] + 48 * fx['tri_tests'] + 32 * fx['rays']
print(json.dumps(dict(config='configs[3]: Two-Level Grid build + trace, %s (%d triangles), %dx%d' % (name, sc.n_tris, res[0], res[1]),
                      build_ms=round(build_ms, 3), trace_ms=round(trace_ms, 3), verified=verified, trace_plus_shade_ms=round(frame_ms, 3), trace_then_shade_two_launches_ms=round(two_ms, 3),
                      mrays_per_s=round(npx / frame_ms / 1e3, 1), algorithmic_bytes_per_launch=alg,
                      reference_algorithm_gbps=round(alg / trace_ms / 1e6, 1),
                      note='the reference algorithm\'s bytes per second exceed the 8 TB/s HBM peak: cells, ranges and triangles are served from L2 / L1 and 61 % of the triangle tests are skipped by the mailbox, so this is no fraction of the HBM roof',
                      per_ray=dict(top_cells=round(fx['top_visits'] / fx['rays'], 1), leaf_cells=round(fx['leaf_visits'] / fx['rays'], 1),
                                   triangle_tests=round(fx['tri_tests'] / fx['rays'], 1)), info=g.info())))
