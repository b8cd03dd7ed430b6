#!/usr/bin/env python3
"""bench.py -- Mrays/s of the BVH primary-ray path (trace + Blinn-Phong shade) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

One STEP = one frame of EACH of the five bundled TestModels scenes (Cube, Gardenlight, Tenryuu, ManFigure,
ClumsyDragon; BASELINE.json configs[1]) at 1920x1080, demo default camera: BVH traversal + shading.  Scene buffers, BVHs
and output buffers are resident in HBM before the timed region.  The five frames of a step are independent and are issued
on five HIP streams (one library context per scene), so one frame's tail is filled by the next frame's blocks.

N > 1 (one process per GPU): every frame is cut into 64x64-pixel tiles dealt over the ranks; each rank traces and shades
its tiles, ONE gather per frame -- RCCL point-to-point into rank 0, issued by the library itself (rt_multi_frame_*,
openclraytracer_amd/csrc/rtmi_multi.hip: no Python and no torch in the data path) -- assembles it; total work per step is
fixed, so scaling is "strong".  torch.distributed (gloo) only carries the barrier, the 128-byte communicator id and the
max over ranks of the elapsed time.

The line printed by rank 0 also carries
  roofline            HBM: bytes the traversal launch really moves (rocprofv3 FETCH_SIZE / WRITE_SIZE, collected by THIS run in
                      child processes before the GPU is touched, corrected as MI355X_MICROARCH.md prescribes) over the launch
                      duration measured live with HIP events, vs 8 TB/s.  The reference algorithm's bytes (SURVEY.md 8(d):
                      56 I + 40 L + 32 per ray) over the same time are reported beside it: the records are cache resident and one
                      fetch serves up to 64 rays of a packet, so that figure exceeds the HBM peak and is not a fraction of it.
  roofline_valu_issue the resource that binds: vector instructions issued (SQ_INSTS_VALU x 2 cycles per wave64 instruction on a
                      SIMD) over the SIMD cycles of the launch (1024 SIMDs x GRBM_GUI_ACTIVE / 8), per scene and for the
                      kernel that takes the most time, with the scalar-unit counters beside it.
  cpu_baseline        the CPU oracle (a restatement of the reference's host path, pinned on the reference's own kernels) timed
                      on this box's host cores on the same five frames (rank 0, N = 1 only).  A reported baseline, not the target.
  config5             BASELINE.json configs[4]: the Sponza-class stand-in at 3840x2160, BVH, tiled over the N ranks.
  per_scene[..].rebuild_plus_trace_ms   the dynamic-scene frame (OpenCLRayTracerDemo.cpp:230-234): initializeFrame + construct +
                      generateContacts + shade (rt_render_frame: one eager sequence, the trace kernels shade what they store), every frame.
  moving_camera       the Demo's RTFrame loop (OpenCLRayTracerDemo.cpp:271-326: the camera is updated before every frame, :275-276):
                      every frame a new camera position and orientation, the per-camera preparation (k_prepare_camera) inside
                      the timed region; graph path (rt_frame_graph_set_camera + launch) and eager path, per scene and per step of
                      five frames, next to the same poses rendered with a camera that stands still.
  per_scene[..].first_frame_ms          the first generateContacts(Camera&) on a fresh BVH: strategy calibration included.
"""
from __future__ import annotations

import argparse
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# five streams per process: give each its own hardware queue (the HIP runtime multiplexes streams onto 4 by default)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

SCENES = ['cube', 'gardenlight', 'tenryuu', 'man', 'clumsydragon']
RES = (1920, 1080)
RES5 = (3840, 2160)         # config 5
HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
SIMDS = 256 * 4             # 256 CUs x 4 SIMDs
VALU_CYCLES_PER_INST = 2.0  # wave64 vector instruction on a SIMD (MI355X_MICROARCH.md, per-instruction cycle constants)
ISOLATED_FRAMES = 10        # frames per scene of the isolated (one scene at a time) measurement pass
PMC_FRAMES = 3              # frames per scene in a counter pass
PMC_PASSES = [('fetch', ['FETCH_SIZE']), ('write', ['WRITE_SIZE']),
              ('sq', ['SQ_WAVES', 'SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_SMEM', 'SQ_INST_CYCLES_SALU', 'SQ_WAIT_INST_ANY',
                      'SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'GRBM_GUI_ACTIVE'])]


def pack_scene_file(scene_txt):
    """Packed scene buffer through the PRODUCT's loader: the C++ Scene::load packing (apps/bin/scene_pack_tool,
    openclraytracer_amd/csrc/host/src/SceneLoader.cpp)."""
    if os.environ.get('RTMI_PACKED') == '1':
        # under rocprofv3 (tools/profile_round.sh) the scenes come packed and this process starts NO child: the profiler's library may
        # have initialised the GPU before the program started, and a process in that state must not spawn
        packed = '/tmp/_%s.bin' % os.path.splitext(os.path.basename(scene_txt))[0]
        if not os.path.exists(packed):
            raise SystemExit('RTMI_PACKED=1 but %s is missing: pack the scenes before the profiled pass (tools/profile_round.sh)' % packed)
        return open(packed, 'rb').read()
    tool = os.environ.get('RTMI_SCENE_PACK_TOOL') or os.path.join(ROOT, 'apps', 'bin', 'scene_pack_tool')
    if not os.path.exists(tool):
        raise SystemExit('apps/bin/scene_pack_tool is not built: run __graft_entry__.build()')
    with tempfile.NamedTemporaryFile(suffix='.bin') as tmp:
        subprocess.check_call([tool, os.path.relpath(scene_txt, ROOT), tmp.name], cwd=ROOT)
        return open(tmp.name, 'rb').read()


def load_scene_bytes(name):
    return pack_scene_file(os.path.join(ROOT, 'assets', 'scenes', name + '.txt'))


def load_atrium_bytes():
    """config 5's scene: Sponza is not shipped with the reference; seeded procedural stand-in of the same size (assets/synth)"""
    sys.path.insert(0, os.path.join(ROOT, 'assets', 'synth'))
    import gen_scenes
    return pack_scene_file(gen_scenes.ensure()['synth_atrium'])


def load_bunny_bytes():
    """config 3's scene: the Stanford Bunny is not shipped with the reference; seeded procedural stand-in of the same size (assets/synth)"""
    sys.path.insert(0, os.path.join(ROOT, 'assets', 'synth'))
    import gen_scenes
    return pack_scene_file(gen_scenes.ensure()['synth_bunny'])


def demo_camera(scene_bytes, resx, resy):
    """The Demo's default camera (OpenCLRayTracerDemo.cpp:202-217) as host code of the benchmark."""
    from openclraytracer_amd import capi
    f = np.frombuffer(scene_bytes, dtype='<f4', count=8, offset=64)
    lo, hi = f[0:3].astype(np.float32), f[4:7].astype(np.float32)
    half = (hi - lo) * np.float32(0.5)
    cam = np.zeros(1, dtype=capi.CAMERA)
    PI = np.float32(3.14159265359)
    deg2rad = PI / np.float32(180.0)
    cam['FOVDistance'] = (np.float32(resy) * np.float32(0.5)) / np.tan(np.float32(90.0) * np.float32(0.5) * deg2rad,
                                                                       dtype=np.float32)
    cam['resX'], cam['resY'], cam['supersamplingFactor'] = resx, resy, 1
    m = np.zeros(12, dtype=np.float32)
    m[0] = m[5] = m[10] = 1.0
    m[3], m[7], m[11] = lo[0] + half[0], lo[1] + half[1], lo[2] - half[2]
    cam['viewTransform'][0] = m
    return cam


def moving_cameras(scene_bytes, resx, resy, count):
    """`count` camera poses around the Demo's default one: the position moves on a small closed curve (0.5 % of the model's box),
    the orientation turns by up to +-1 degree about y and x -- what a user dragging the mouse / holding a key does to the camera
    between frames (OpenCLRayTracerDemo.cpp:378-460).  Pose 0 is the default camera itself (for ClumsyDragon: on a box plane)."""
    from openclraytracer_amd import capi
    base = demo_camera(scene_bytes, resx, resy)
    f = np.frombuffer(scene_bytes, dtype='<f4', count=8, offset=64)
    size = (f[4:7] - f[0:3]).astype(np.float64)
    cams = np.zeros(count, dtype=capi.CAMERA)
    for k in range(count):
        cams[k] = base[0]
        if k == 0:
            continue
        t = 2.0 * np.pi * k / count
        a, b = np.radians(1.0) * np.sin(t), np.radians(1.0) * np.sin(2.0 * t + 0.5)
        ca, sa, cb, sb = np.cos(a), np.sin(a), np.cos(b), np.sin(b)
        rot = np.array([[ca, 0, sa], [0, 1, 0], [-sa, 0, ca]]) @ np.array([[1, 0, 0], [0, cb, -sb], [0, sb, cb]])
        m = cams[k]['viewTransform']
        for r in range(3):
            m[4 * r:4 * r + 3] = rot[r].astype(np.float32)
        m[3] += np.float32(0.005 * size[0] * np.sin(t) + 1e-3 * size[0])
        m[7] += np.float32(0.005 * size[1] * np.sin(3.0 * t))
        m[11] += np.float32(0.005 * size[2] * (np.cos(t) - 1.0))
    return cams


# ---------------------------------------------------------------------------------------------------------------------------
# hardware counters of the traversal launches, collected by this run: bench.py starts itself under rocprofv3 (the program
# directly behind `--`), once per counter pass, BEFORE this process touches the GPU
# ---------------------------------------------------------------------------------------------------------------------------
def pmc_child(packed_dir):
    """the profiled child: the five scenes, first frame (calibration), then PMC_FRAMES isolated traversal launches per scene
    whose kernels carry the template tag <1, 2>.  The scenes come packed from the parent (files in packed_dir): this process
    runs under the profiler with the GPU initialised and starts no process of its own."""
    from openclraytracer_amd import capi
    ctx = capi.Context(0)
    info = {}
    for name in SCENES:
        buf = open(os.path.join(packed_dir, name + '.bin'), 'rb').read()
        scene = capi.DeviceScene(ctx, buf)
        bvh = capi.DeviceBVH(ctx, scene).build()
        cam = demo_camera(buf, *RES)
        dc = ctx.alloc(RES[0] * RES[1] * 32)
        bvh.generate_contacts_primary_d(cam, dc)
        ctx.synchronize()
        ctx.set_profiling(2)
        for _ in range(PMC_FRAMES):
            bvh.generate_contacts_primary_d(cam, dc)
        ctx.synchronize()
        ctx.set_profiling(False)
        info[name] = bvh.trace_strategy()
        ctx.free(dc); bvh.close(); scene.close()
    ctx.close()
    print('PMC_CHILD ' + json.dumps(info), flush=True)


def kernel_tag(name):
    """(PRIMARY, CAL) of a traversal kernel's name: k_trace<1, 2>, k_trace_packet4<1, 2, 0> (a third argument: sub-tree sharing)"""
    try:
        a = name[name.index('<') + 1:name.rindex('>')].split(',')
        return int(a[0]), int(a[1])
    except (ValueError, IndexError):
        return None


def run_pmc_passes(scene_bytes, keep_dir=None):
    """-> dict with per-scene and per-kernel counter sums per traversal launch, or None when rocprofv3 is unavailable"""
    rocprof = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(rocprof):
        return None
    base = tempfile.mkdtemp(prefix='rtmi_pmc_', dir='/tmp')
    packed = os.path.join(base, 'packed')
    os.makedirs(packed)
    for name, buf in scene_bytes.items():
        with open(os.path.join(packed, name + '.bin'), 'wb') as f:
            f.write(buf)
    env = dict(os.environ, TMPDIR='/tmp')
    rows = []            # (pass, dispatch order, kernel, counter, value)
    strategies = None
    try:
        for tag, counters in PMC_PASSES:
            out = os.path.join(base, tag)
            cmd = [rocprof, '--pmc'] + counters + ['--output-format', 'csv', '-d', out, '--', sys.executable, os.path.join(ROOT, 'bench.py'), '--pmc-child', packed]
            try:
                r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
            except (OSError, subprocess.TimeoutExpired) as e:
                print('bench: counter pass %s did not run: %s' % (tag, e), file=sys.stderr)
                return None
            line = [l for l in r.stdout.splitlines() if l.startswith('PMC_CHILD ')]
            if r.returncode != 0 or not line:
                print('bench: counter pass %s failed (rc %d): %s' % (tag, r.returncode, (r.stderr or r.stdout)[-400:]), file=sys.stderr)
                return None
            strategies = json.loads(line[-1][len('PMC_CHILD '):])
            files = sorted(glob.glob(os.path.join(out, '**', '*counter_collection.csv'), recursive=True), key=os.path.getmtime)
            if not files:
                print('bench: counter pass %s left no counter_collection.csv' % tag, file=sys.stderr)
                return None
            for row in csv.DictReader(open(files[-1])):
                k = row['Kernel_Name'].split('(')[0].replace('void ', '').strip()
                if k.startswith(('k_trace', 'k_resolve_split')) and kernel_tag(k) == (1, 2):
                    rows.append((tag, int(row['Dispatch_Id']), k, row['Counter_Name'], float(row['Counter_Value'])))
        if keep_dir:
            os.makedirs(keep_dir, exist_ok=True)
            with open(os.path.join(keep_dir, 'bench_pmc_rows.csv'), 'w') as f:
                f.write('pass,dispatch_id,kernel,counter,value\n')
                for r_ in rows:
                    f.write('%s,%d,"%s",%s,%.3f\n' % r_)
    finally:
        shutil.rmtree(base, ignore_errors=True)
    # dispatches in order: every scene contributes PMC_FRAMES launches = PMC_FRAMES first-phase kernels (+ k_trace_long (+ k_resolve_split))
    first = ('k_trace<', 'k_trace_packet<', 'k_trace_packet4<')
    per_scene = {n: collections.defaultdict(float) for n in SCENES}
    per_kernel = collections.defaultdict(lambda: collections.defaultdict(float))
    kernel_dispatches = collections.defaultdict(int)
    for tag, _ in PMC_PASSES:
        disp = sorted({(d, k) for t, d, k, _, _ in rows if t == tag})
        scene_of, launches = {}, 0
        for d, k in disp:
            if k.startswith(first):
                launches += 1
            scene_of[d] = SCENES[min((launches - 1) // PMC_FRAMES, len(SCENES) - 1)] if launches else SCENES[0]
        if launches != PMC_FRAMES * len(SCENES):
            print('bench: counter pass %s saw %d traversal launches, expected %d' % (tag, launches, PMC_FRAMES * len(SCENES)), file=sys.stderr)
            return None
        for t, d, k, c, v in rows:
            if t == tag:
                per_scene[scene_of[d]][c] += v / PMC_FRAMES
                per_kernel[k][c] += v
        if tag == 'sq':
            for d, k in disp:
                kernel_dispatches[k] += 1
    return dict(per_scene={n: dict(v) for n, v in per_scene.items()}, per_kernel={k: dict(v) for k, v in per_kernel.items()},
                kernel_dispatches=dict(kernel_dispatches), strategies=strategies)


def valu_issue(c):
    """fraction of the VALU issue peak: wave64 vector instructions x 2 cycles over SIMDs x cycles (GRBM_GUI_ACTIVE sums the 8 XCDs)"""
    cycles = c.get('GRBM_GUI_ACTIVE', 0.0) / 8.0
    return (c.get('SQ_INSTS_VALU', 0.0) * VALU_CYCLES_PER_INST / (SIMDS * cycles)) if cycles else None


_T0 = time.perf_counter()


def note(msg):
    """progress on stderr (a long run must keep talking: the GPU pool takes seven silent minutes for a hang)"""
    if os.environ.get('RANK', '0') == '0':
        print('bench [%6.1f s] %s' % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


# ---------------------------------------------------------------------------------------------------------------------------
# in-run verification: the buffers the timed region wrote are compared with the CPU oracle's committed digests
# (tests/golden/bench_frames.json, made by tests/golden/make_golden_bench.py) -- contacts bit for bit (SHA-256), shaded RGBA within 1e-5
# relative (per-channel sums in float64 and every sample_stride-th pixel; alpha exact).  A mismatch fails the run.
# ---------------------------------------------------------------------------------------------------------------------------
RGB_TOL = 1e-5
_BENCH_FIXTURE = None


def bench_fixture():
    global _BENCH_FIXTURE
    if _BENCH_FIXTURE is None:
        _BENCH_FIXTURE = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'bench_frames.json')))
    return _BENCH_FIXTURE


def sha256_hex(a):
    import hashlib
    return hashlib.sha256(memoryview(np.ascontiguousarray(a)).cast('B')).hexdigest()


def camera_from_bits(bits):
    from openclraytracer_amd import capi
    return np.array(bits, dtype=np.uint32).view(capi.CAMERA).copy()


def rgba_mismatch(rgba, rec, stride):
    """rgba: float32 [npx, 4] as rendered; rec: fixture record.  -> None or a description of the first difference"""
    got_sum = rgba.astype(np.float64).sum(axis=0)
    want_sum = np.array(rec['rgba_sum'], dtype=np.float64)
    rel = np.abs(got_sum - want_sum) / np.maximum(np.abs(want_sum), 1.0)           # (a channel no material has sums to 0)
    if float(np.where(got_sum == want_sum, 0.0, rel).max()) > RGB_TOL:
        return 'rgba channel sums %s differ from the oracle\'s %s by more than %g relative' % (got_sum.tolist(), want_sum.tolist(), RGB_TOL)
    want = np.array(rec['rgba_samples_bits'], dtype=np.uint32).view(np.float32)
    got = rgba[::stride]
    if got.shape != want.shape:
        return 'rgba sample count %s != %s' % (got.shape, want.shape)
    if not np.array_equal(got[:, 3], want[:, 3]):
        return 'alpha of the sampled pixels differs from the oracle\'s'
    err = np.abs(got.astype(np.float64) - want) / np.maximum(np.abs(want.astype(np.float64)), 1e-30)
    err = np.where(got == want, 0.0, err)
    if float(err.max()) > RGB_TOL:
        k = int(np.argmax(err.max(axis=1)))
        return 'sampled pixel %d: %s vs the oracle\'s %s' % (k * stride, got[k].tolist(), want[k].tolist())
    return None


def verify_frame(ctx, d_contacts, d_rgba, npx, rec, stride, what):
    """downloads one rendered frame (contacts and / or RGBA device buffers) and compares it with a fixture record"""
    if d_contacts is not None:
        got = sha256_hex(ctx.download(d_contacts, npx * 32, np.uint8))
        if got != rec['contacts_sha256']:
            raise SystemExit('VERIFICATION FAILED (%s): contacts sha256 %s, the oracle\'s %s' % (what, got, rec['contacts_sha256']))
    if d_rgba is not None:
        msg = rgba_mismatch(ctx.download(d_rgba, npx * 4, np.float32).reshape(npx, 4), rec, stride)
        if msg:
            raise SystemExit('VERIFICATION FAILED (%s): %s' % (what, msg))


def ray_buffer_leg(ctx, bvh, cam, fx, frames=8):
    """The second entry point of the manager, generateContacts(CLBuffer& rays, CLBuffer& contacts, n) (BVHManager.cpp:292-329, kernel
    generateContacts2 BVHKernels.cl:103-117): (i) the camera's own rays handed over as Ray[], (ii) the reflection bounce and (iii) the
    shadow rays towards the scene's light, both generated on the device from the contacts of (i) (rt_generate_secondary_rays; the use
    the reference plans, KNOWN_ISSUES_AND_TODO.txt:27-30).  Per leg: HIP-event time of `frames` calls, Mrays/s, and the rays and the
    contacts compared with the oracle's digests (fx = fixture entry 'ray_buffer')."""
    from openclraytracer_amd import capi
    npx = int(cam['resX'][0]) * int(cam['resY'][0])
    d_rays0, d_rays1, d_c0, d_c1 = ctx.alloc(npx * 48), ctx.alloc(npx * 48), ctx.alloc(npx * 32), ctx.alloc(npx * 32)
    light, eps = bench_fixture()['light'], bench_fixture()['eps']
    out = {}
    tm = capi.Timer(ctx)
    try:
        capi.generate_primary_rays_d(ctx, cam, d_rays0)
        bvh.generate_contacts_rays_d(d_rays0, npx, d_c0)
        ctx.synchronize()
        for leg in ('primary', 'reflect', 'shadow'):
            if leg == 'primary':
                d_rays, n, d_c = d_rays0, npx, d_c0
            else:
                n = capi.generate_secondary_rays_d(ctx, d_c0, npx, capi_kind(leg), d_rays1, light, eps, d_rays_in=d_rays0)
                d_rays, d_c = d_rays1, d_c1
            want = fx[leg]
            if n != want['rays']:
                raise SystemExit('VERIFICATION FAILED (ray buffer, %s): %d rays generated, the oracle\'s %d' % (leg, n, want['rays']))
            got = sha256_hex(ctx.download(d_rays, n * 48, np.uint8))
            if got != want['rays_sha256']:
                raise SystemExit('VERIFICATION FAILED (ray buffer, %s): generated rays differ from the oracle\'s' % leg)
            for _ in range(2):
                bvh.generate_contacts_rays_d(d_rays, n, d_c)
            ctx.synchronize()
            tm.start()
            for _ in range(frames):
                bvh.generate_contacts_rays_d(d_rays, n, d_c)
            tm.stop()
            ms = tm.elapsed_ms() / frames
            got = sha256_hex(ctx.download(d_c, n * 32, np.uint8))
            if got != want['contacts_sha256']:
                raise SystemExit('VERIFICATION FAILED (ray buffer, %s): contacts sha256 %s, the oracle\'s %s' % (leg, got, want['contacts_sha256']))
            out[leg] = dict(rays=n, ms=round(ms, 4), mrays_per_s=round(n / ms / 1e3, 1), hits=want['hits'], verified=True,
                            reference_inner_visits_per_ray=round(want['inner_visits'] / max(n, 1), 1), strategy=bvh.rays_strategy())
    finally:
        tm.close()
        for d in (d_rays0, d_rays1, d_c0, d_c1):
            ctx.free(d)
    return out


LINE_LIMIT = 4096      # bytes: a driver that keeps the tail of stdout must still find the whole line in it (round 4's 20 KB line was cut)


def compact_line(out, details_path):
    """The ONE line of stdout: the contract's keys, `roofline` and `cpu_baseline` with the numbers only, `verified` as counts, and one
    figure per further leg.  The whole record (per scene, per kernel, counters, the wording of every formula) goes to `--details`."""
    keep = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype')
    line = {k: out[k] for k in keep}
    line['data'] = 'bundled TestModels scenes, demo camera'
    cfg = out['config']
    line['config'] = dict(workload=cfg['workload'], rays_per_step=cfg['rays_per_step'],
                          parallelism=('single GPU, five frames in flight' if out['n_gpus'] == 1 and 'REHEARSAL' not in cfg['parallelism']
                                       else '64x64 tiles over %d ranks, one gather per frame' % cfg.get('ranks', out['n_gpus'])))
    rf = out['roofline']
    line['roofline'] = {k: rf.get(k) for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'avg_launch_ms', 'compulsory_bytes_per_step',
                                                'frac_of_output_floor', 'reference_algorithm_bytes_per_launch', 'walked_bytes_per_launch')}
    line['roofline']['kernel'] = 'traversal launch of a frame (k_trace_packet4 / k_trace + k_trace_long)'
    rv = out.get('roofline_valu_issue')
    if rv:
        line['roofline']['valu_issue_frac'] = rv['frac']
        line['roofline']['valu_issue_kernel'] = rv['kernel'][:48]
    cb = out.get('cpu_baseline')
    if cb:
        line['cpu_baseline'] = {k: cb[k] for k in ('value', 'unit', 'cores', 'kind', 'sample', 'single_thread_value')}
    else:
        line['cpu_baseline'] = None
    v = out.get('verified')
    if v:
        line['verified'] = {k: (x if not isinstance(x, str) else (x.startswith('frame equals'))) for k, x in v.items() if k != 'against'}
    else:
        line['verified'] = None
    legs = {}
    ps = out.get('per_scene') or {}
    if ps:
        legs['frame_ms'] = {n: q['trace_ms'] for n, q in ps.items()}
        rb = [q['rebuild_plus_trace_ms'] / max(q['trace_ms'], 1e-9) for q in ps.values() if 'rebuild_plus_trace_ms' in q]
        if rb:
            legs['rebuild_plus_trace_over_static'] = [round(min(rb), 2), round(max(rb), 2)]
    mv = out.get('moving_camera')
    if mv:
        legs['moving_camera_step_ms'] = mv['step_ms']
    r = out.get('ray_buffer')
    if r:
        legs['ray_buffer_min_mrays'] = {k: r['min_mrays_per_s_' + k] for k in ('primary', 'reflect', 'shadow')}
    if out.get('config3'):
        legs['config3_ms'] = out['config3']['ms_per_frame']
    if out.get('config4'):
        c4 = out['config4']
        legs['config4'] = dict(build_ms=c4['build_ms'], trace_shade_ms=c4['trace_plus_shade_ms'], mrays=c4['mrays_per_s_grid_kept'])
    if out.get('config5'):
        legs['config5'] = dict(ms=out['config5']['ms_per_frame'], mrays=out['config5']['mrays_per_s'])
    line['legs'] = legs
    if out.get('per_rank'):
        line['per_rank'] = dict(columns=['trace_shade_ms_per_step', 'gather_ms_per_step', 'uses_rccl'],
                                rows=[[r['trace_shade_ms_per_step'], r['gather_ms_per_step'], bool(r['uses_rccl'])] for r in out['per_rank']][:8])
    line['details_file'] = (os.path.relpath(details_path, ROOT) if details_path else None)
    # never above the limit: the optional parts go first
    for drop in ('per_rank', 'legs'):
        if len(json.dumps(line)) <= LINE_LIMIT:
            break
        line.pop(drop, None)
    assert len(json.dumps(line)) <= LINE_LIMIT, 'summary line too long'
    return line


def same_launch(a, b):
    """two calibrated strategies that are the same launch as far as its counters go: the same kernel, and either both without a hand-over / sharing
    budget (2^20 = never) or both with one (neighbouring budgets -- 12 or 16 steps, 32 or 64 -- are the same instance doing nearly the same work)"""
    if not a or not b or a['kernel'] != b['kernel']:
        return False
    return (a['budget'] >= 1 << 20) == (b['budget'] >= 1 << 20)


def q_slots(work, name):
    for it in work:
        if it['name'] == name:
            return it['slots']
    return 0


def capi_kind(leg):
    from openclraytracer_amd import capi
    return {'shadow': 0, 'reflect': 1}[leg]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--settle-ms', type=float, default=60.0,
                    help='before the W warm-up steps: keep the device busy with the same steps for this long, untimed, so that the timed region runs on '
                         'settled clocks (tools/tail_probe.py: a device that has idled for 20 ms needs more than a millisecond of work to get back to '
                         'its clocks -- 20 steps behind 5 warm-up steps 0.28-0.29 ms per step, behind a busy 70 ms 0.25-0.26); 0: none')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-pmc', action='store_true', help='skip the rocprofv3 counter passes (roofline.traffic is then null)')
    ap.add_argument('--no-extra-legs', action='store_true', help='skip config 5 and the rebuild-per-frame leg')
    ap.add_argument('--no-graphs', action='store_true', help='N = 1: issue every frame eagerly instead of replaying a HIP graph per frame')
    ap.add_argument('--check-frames', action='store_true',
                    help='tiled: rank 0 also traces every frame untiled and compares it with the assembled one')
    ap.add_argument('--no-verify', action='store_true', help='skip the in-run comparison of the rendered frames with the oracle digests')
    ap.add_argument('--keep-pmc', default=None, help='directory that receives the raw counter rows of this run')
    ap.add_argument('--pmc-child', default=None, metavar='DIR', help=argparse.SUPPRESS)
    ap.add_argument('--details', default=os.path.join(ROOT, 'gpurun_out', 'bench_details.json'),
                    help='file that receives the whole record of the run (every leg, per scene, counters); the ONE line on stdout is its summary. '
                         'Empty string: no file')
    ap.add_argument('--full-line', action='store_true', help='print the whole record on stdout instead of the summary line (round 4\'s output: 20 KB)')
    args = ap.parse_args()
    if args.pmc_child:
        return pmc_child(args.pmc_child)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit('launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)' % (args.gpus, world))
    # tests on a one-GPU box: RTMI_BENCH_FORCE_TILED=1 runs the N > 1 code path with one rank; RTMI_BENCH_GROUP=n runs it with n
    # ranks inside this process, all on one device (the library then gathers by device copies: RCCL refuses two ranks on a device)
    group_ranks = int(os.environ.get('RTMI_BENCH_GROUP', '0')) if world == 1 else 0
    tiled = world > 1 or group_ranks > 0 or os.environ.get('RTMI_BENCH_FORCE_TILED') == '1'
    nranks = world if world > 1 else max(group_ranks, 1)
    # rank 0 gathers, so it owns e + 1 of every N + e tiles (RT_TILING): what it traces itself does not cross a link.  One GPU now
    # shades ~17.5 Grays/s = 210 GB/s of 12-byte colour, more than a link carries, so the peers' shares are what bounds a step of
    # five small frames: e = 2 up to four ranks, 1 from eight on ("What the links allow")
    root_extra = int(os.environ.get('RTMI_BENCH_ROOT_EXTRA', '0' if nranks <= 1 else ('2' if nranks < 8 else '1')))

    # ---- everything that needs child processes happens before this process touches the GPU ------------------------------
    scene_bytes = {name: load_scene_bytes(name) for name in SCENES}
    atrium_bytes = None if args.no_extra_legs else load_atrium_bytes()
    bunny_bytes = None if args.no_extra_legs else load_bunny_bytes()
    pmc = None
    if rank == 0 and not tiled and not args.no_pmc:
        note('counter passes (rocprofv3 --pmc, child processes)')
        pmc = run_pmc_passes(scene_bytes, args.keep_pmc)
    note('scenes, BVHs, first frames')

    # Which HIP runtime serves the process: librtmi.so is linked against the system's (/opt/rocm), torch ships a copy of its own
    # (an older ROCm) that satisfies the same dependency when it is loaded first.  Measured in round 3: under torch's copy a graph
    # instance whose kernel arguments were re-set (the camera that moves) replays three times slower and the five-stream
    # moving-camera loop stalls; under the system runtime neither happens.  N = 1 therefore runs without torch (device-wide
    # fences come from the library).  N > 1 needs torch.distributed (gloo: barrier, communicator id, max of the elapsed time) and
    # loads torch FIRST, as in round 2, so that the library, RCCL and torch share ONE runtime; that path replays captured frames
    # only.
    torch, dist, dev = None, None, None
    if world > 1 or os.environ.get('RTMI_BENCH_TORCH_FIRST') == '1':     # (the knob: rehearse the N > 1 runtime combination with one rank)
        import torch
        torch.cuda.set_device(local_rank)
        dev = torch.device('cuda', local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group('gloo')          # control plane only
    from openclraytracer_amd import capi

    def fence():
        if dist is not None:
            dist.barrier()
        if torch is not None:
            torch.cuda.synchronize(dev)
        capi.device_synchronize(local_rank)

    frames_fixture = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'frames.json')))
    search_fixture = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'search_tree_frames.json')))
    key = '%dx%d' % RES
    npx = RES[0] * RES[1]

    # One context = one HIP stream per scene (and per rank of an in-process group): the five frames of a step are independent
    def new_comm_id():
        box = [capi.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return box[0]
    # One RCCL communicator shared by the five scenes' frames (each on a stream of its own: rt_comm_share), or one communicator per
    # scene (RTMI_BENCH_COMM=per_scene; also what the run falls back to, on every rank together, when the first frames on the shared
    # communicator fail: see `tiled_warm_up`)
    def make_frames(it):
        for c in it['ctxs']:
            c.set_frames_in_flight(len(SCENES))          # five scenes' frames are in flight per step (see use_graphs below)
        it['frames'] = [capi.MultiFrame(c, b, s, it['cam'], root_extra, it['root']) for c, b, s in zip(it['comms'], it['bvhs'], it['scenes'])]
        for c in it['ctxs']:
            c.set_frames_in_flight(1)
        it['slots'] = it['frames'][0].layout()['own_slots']
    comm_mode = os.environ.get('RTMI_BENCH_COMM', 'shared')
    comm_id = new_comm_id() if world > 1 else None
    main_comm = None
    diag = dict(rank=rank, world=world, comm_mode=comm_mode, phase='setup', phase_since=time.perf_counter())      # what a failing or hanging rank reports about itself

    def phase(name):
        diag['phase'] = name; diag['phase_since'] = time.perf_counter()

    def report(why):
        # every rank, on stderr, whatever else happens: enough to tell a transport problem from a wrong frame
        d = dict(diag, why=why, seconds_in_phase=round(time.perf_counter() - diag['phase_since'], 1))
        d.pop('phase_since', None)
        try:
            if tiled and work and 'comms' in work[0]:
                d['uses_rccl'] = [bool(it['comms'][0].uses_rccl()) for it in work if 'comms' in it]
                d['rccl_ranks'] = [it['comms'][0].rccl_ranks() for it in work if 'comms' in it]
                d['roots'] = [it.get('root') for it in work]
                d['own_slots'] = [it.get('slots') for it in work]
        except Exception as e:          # noqa: BLE001 (a diagnostic must not raise)
            d['diag_error'] = repr(e)
        print('BENCH_DIAG ' + json.dumps(d), file=sys.stderr, flush=True)

    def watchdog():
        # a rank that waits for a peer forever says where it stands and ends the process (a hung collective cannot be cancelled)
        limit = float(os.environ.get('RTMI_BENCH_WATCHDOG_S', '300'))
        while diag['phase'] != 'done':
            time.sleep(2.0)
            if time.perf_counter() - diag['phase_since'] > limit:
                report('watchdog: no progress for %.0f s' % limit)
                os._exit(3)
    if tiled:
        import threading
        threading.Thread(target=watchdog, daemon=True).start()
    work, build_ms = [], {}
    for name in SCENES:
        buf = scene_bytes[name]
        cam = demo_camera(buf, *RES)
        it = dict(name=name, bytes=buf, cam=cam)
        ranks_here = range(nranks) if group_ranks else [rank]
        it['ctxs'] = [capi.Context(local_rank) for _ in ranks_here]        # raises without a gfx950 device: no fallback
        it['scenes'] = [capi.DeviceScene(c, buf) for c in it['ctxs']]
        it['bvhs'] = [capi.DeviceBVH(c, s) for c, s in zip(it['ctxs'], it['scenes'])]
        t0 = time.perf_counter()
        for b in it['bvhs']:
            b.build()
        build_ms[name] = (time.perf_counter() - t0) * 1e3 / len(it['bvhs'])
        it['ctx'], it['scene'], it['bvh'] = it['ctxs'][0], it['scenes'][0], it['bvhs'][0]
        if tiled:
            if group_ranks:
                it['comms'] = capi.Comm.create_group(it['ctxs'])
            elif comm_mode == 'per_scene':
                it['comms'] = [capi.Comm.create(it['ctx'], comm_id if not work else new_comm_id(), rank, world)]      # collective: ncclCommInitRank
                main_comm = main_comm or it['comms'][0]
            elif main_comm is None:
                main_comm = capi.Comm.create(it['ctx'], comm_id, rank, world)      # collective: ncclCommInitRank
                it['comms'] = [main_comm]
            else:
                it['comms'] = [main_comm.share(it['ctx'])]
            # every scene's frame is assembled on another rank (scene i -> rank i mod N): what crosses the links into one GPU is one
            # frame's colour per step, not all five; RTMI_BENCH_ONE_ROOT=1 gathers everything into rank 0
            it['root'] = 0 if os.environ.get('RTMI_BENCH_ONE_ROOT') == '1' else len(work) % nranks
            make_frames(it)
        else:
            it['slots'] = npx
            it['contacts'] = it['ctx'].alloc(npx * 32)
            it['rgba'] = it['ctx'].alloc(npx * 16)
        work.append(it)

    # N = 1: one HIP graph per frame (trace with the calibrated strategy, then shade): one launch call instead of five.  Creating
    # a graph traces one frame eagerly first, which is also where the BVH gets calibrated.  (Tiled: the library does the same
    # inside rt_multi_frame_create, one graph per send buffer.)
    # the first generateContacts(Camera&) on a fresh BVH calibrates the strategy for this (resolution, tiling): what it costs
    first_frame_ms = {}
    if not tiled:
        for it in work:
            it['ctx'].synchronize()
            t1 = time.perf_counter()
            it['bvh'].generate_contacts_primary_d(it['cam'], it['contacts'])
            it['ctx'].synchronize()
            first_frame_ms[it['name']] = ((time.perf_counter() - t1) * 1e3, it['bvh'].calibration_launches(), it['bvh'].calibration_gpu_ms())
    use_graphs = not tiled and not args.no_graphs and os.environ.get('RTMI_BENCH_GRAPHS', '1') != '0'
    if use_graphs:
        # a step keeps the five scenes' frames in flight together, and says so (rt_ctx_set_frames_in_flight, include/rtmi.h: workgroups
        # of one wave, sub-tree sharing only where it wins by far -- both measured here).  The isolated pass, the first frames and
        # the 4K leg run one frame at a time, with the library's default policy.
        for it in work:
            it['ctx'].set_frames_in_flight(len(SCENES))
            it['graph'] = capi.FrameGraph(it['bvh'], it['scene'], it['cam'], it['contacts'], it['rgba'])
            it['ctx'].set_frames_in_flight(1)

    def step():
        for it in work:
            if tiled:
                capi.multi_frame_render(it['frames'])
            elif use_graphs:
                it['graph'].launch()
            else:
                it['bvh'].generate_contacts_primary_d(it['cam'], it['contacts'])
                capi.shade_blinn_phong_frame_d(it['ctx'], it['scene'], it['cam'], it['contacts'], it['rgba'])

    def check_errors():
        for it in work:
            if tiled:
                for f in it['frames']:
                    f.wait()                       # + the trace kernels' (sticky) error word
            else:
                it['ctx'].synchronize()

    def check_frames(when):
        # the rank that assembles a scene's frame also traces and shades it untiled, on a context of its own, and compares
        checked = 0
        cctx = None
        for it in work:
            root_frame = it['frames'][it['root']] if group_ranks else (it['frames'][0] if rank == it['root'] else None)
            if root_frame is None:
                continue
            if cctx is None:
                cctx = capi.Context(local_rank)
            cscene = capi.DeviceScene(cctx, it['bytes'])
            cbvh = capi.DeviceBVH(cctx, cscene).build()
            want = capi.shade_blinn_phong(cctx, cscene, it['cam'], cbvh.generate_contacts_primary(it['cam'])).reshape(-1, 4)
            root_frame.wait()
            got = root_frame.download_frame()
            cbvh.close(); cscene.close()
            if got.tobytes() != want.tobytes():
                raise SystemExit('frame check failed for %s (%s): the assembled frame differs from the untiled one' % (it['name'], when))
            checked += 1
        if cctx is not None:
            cctx.close()
        if dist is not None:
            box = [None] * world
            dist.all_gather_object(box, checked)
            checked = sum(box)
        if rank == 0:
            print('frame check ok (%s): %d assembled frames equal the untiled ones' % (when, checked), file=sys.stderr)
        fence()

    bfx = None if args.no_verify else bench_fixture()

    def verify_timed(when):
        # N = 1: every scene's contact and colour buffers as the last frame left them; tiled: the assembled colour frame on the rank that
        # assembles it (contacts stay on the ranks that traced them).  -> frames verified over all ranks
        done = 0
        for it in work:
            rec = bfx['scenes'][it['name']][key]['default']
            if [int(x) for x in np.asarray(it['cam']).view(np.uint32).ravel()] != rec['camera_bits']:
                raise SystemExit('VERIFICATION FAILED (%s): the camera differs from the fixture\'s' % it['name'])
            if tiled:
                root_frame = it['frames'][it['root']] if group_ranks else (it['frames'][0] if rank == it['root'] else None)
                if root_frame is None:
                    continue
                root_frame.wait()
                msg = rgba_mismatch(root_frame.download_frame(), rec, bfx['sample_stride'])
                if msg:
                    raise SystemExit('VERIFICATION FAILED (%s, assembled frame, %s): %s' % (it['name'], when, msg))
            else:
                verify_frame(it['ctx'], it['contacts'], it['rgba'], npx, rec, bfx['sample_stride'], '%s, %s' % (it['name'], when))
            done += 1
        if dist is not None:
            box = [None] * world
            dist.all_gather_object(box, done)
            done = sum(box)
        return done

    def all_ranks_ok(ok):
        if dist is None:
            return bool(ok)
        t = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t.item()))

    def tiled_warm_up():
        # the first frames of a multi-rank run: a rank whose frames fail (an RCCL error surfaces in rt_multi_frame_render / _wait) says so,
        # and all ranks decide together -- a rank that went on alone would wait in a gather nobody else issues
        err = None
        try:
            for _ in range(max(args.warmup, 1)):
                step()
            for it in work:
                for f in it['frames']:
                    f.wait()
        except capi.RtError as e:
            err = str(e)
        return all_ranks_ok(err is None), err

    note('warm-up and timed steps')
    phase('warm-up')
    if tiled:
        ok, err = tiled_warm_up()
        if not ok and world > 1 and comm_mode == 'shared':
            # the five frames' streams share one RCCL communicator: if that combination does not work on this node, every rank takes a
            # communicator per scene instead (collective: new ids over the control plane), and says so in the line
            report('first frames failed on the shared communicator (%s): falling back to one communicator per scene' % (err or 'a peer failed'))
            for it in work:
                for f in it['frames']:
                    f.close()
            comm_mode = diag['comm_mode'] = 'per_scene (fall-back)'
            for it in work:
                it['comms'] = [capi.Comm.create(it['ctx'], new_comm_id(), rank, world)]
                make_frames(it)
            ok, err = tiled_warm_up()
        if not ok:
            report('the first frames failed: %s' % (err or 'on another rank'))
            raise SystemExit(2)
    # device clocks: the set-up above is mostly host work (captures, uploads, digests), and a device that has idled runs its first milliseconds slower
    settled = 0

    def settle():
        nonlocal settled
        if args.settle_ms <= 0:
            return
        phase('settling the device clocks')
        t_settle = time.perf_counter()
        # (several ranks: the same number of steps on every rank -- fence() holds a barrier, and every step a gather the ranks take part in;
        #  at ~0.3 ms per step that is the same order of time)
        batches = max(1, int(args.settle_ms / 3.0)) if dist is not None else None
        while (batches is None and (time.perf_counter() - t_settle) * 1e3 < args.settle_ms) or (batches is not None and settled < batches * 10):
            for _ in range(10):
                step()
            settled += 10
            fence()
    if not tiled:
        settle()
        for _ in range(args.warmup):
            step()
    fence()
    if tiled and (args.check_frames or not args.no_verify):
        if args.warmup == 0:
            step(); fence()
        if not args.no_verify:          # by default: the first assembled frame of every scene against the oracle's digest, before timing
            note('verified before timing: %d assembled frames' % verify_timed('after the warm-up'))
        if args.check_frames:
            check_frames('after the warm-up')
    if tiled:
        settle()                        # (behind the warm-up steps and their verification: the digests are host work)
    phase('timed region')
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    issued = time.perf_counter() - t0            # host time to issue the steps (the floor when the device part shrinks)
    fence()
    elapsed = time.perf_counter() - t0
    check_errors()                               # an invalid frame anywhere in the timed region fails the run
    if args.check_frames and tiled:
        check_frames('after the timed steps')     # the frames of the last, fully overlapped steps
    # ---- what the timed region wrote, against the oracle's digests (outside the timed region; a mismatch ends the run non-zero) ----
    verified = None
    if not args.no_verify:
        verified = dict(against='tests/golden/bench_frames.json (CPU oracle): contacts SHA-256, RGBA sums and samples within %g relative' % RGB_TOL,
                        frames=verify_timed('the last step of the timed region'))
        note('verified: %d frames of the timed region equal the oracle\'s' % verified['frames'])
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    rays_per_step = npx * len(SCENES)
    value = rays_per_step * args.steps / elapsed / 1e6

    # ---- N > 1: where a step's time goes on every rank (untimed extra pass: events around each rank's trace + shade and around its
    # part of the gather; host time to issue a step) -- so that a scaling run explains itself
    per_rank = None
    if tiled:
        K = max(10, min(args.steps, 100))
        for it in work:
            for f in it['frames']:
                f.set_timing(True)
        fence()
        t1 = time.perf_counter()
        for _ in range(K):
            step()
        host_issue = (time.perf_counter() - t1) * 1e3 / K
        fence()
        wall = (time.perf_counter() - t1) * 1e3 / K
        mine = []
        for i, f0 in enumerate(work[0]['frames']):
            tr = ga = 0.0
            for it in work:
                t = it['frames'][i].times()
                tr += t['trace_ms'] / max(t['frames'], 1); ga += t['gather_ms'] / max(t['frames'], 1)
            c = work[0]['comms'][i]
            mine.append(dict(rank=(i if group_ranks else rank), trace_shade_ms_per_step=round(tr, 4), gather_ms_per_step=round(ga, 4),
                             host_issue_ms_per_step=round(host_issue / (len(work[0]['frames'])), 4), step_ms=round(wall, 4),
                             uses_rccl=c.uses_rccl(), rccl_ranks=c.rccl_ranks(), roots_here=[it['name'] for it in work if it['root'] == (i if group_ranks else rank)]))
        for it in work:
            for f in it['frames']:
                f.set_timing(False)
        if dist is not None:
            box = [None] * world
            dist.all_gather_object(box, mine)
            per_rank = [x for b in box for x in b]
        else:
            per_rank = mine

    # ---- per-scene device times (untimed extra pass, HIP events on the stream the kernels run on; one scene at a time: what a
    # frame costs when it has the device to itself) -- the launches rocprofv3 lists as <1, 2>
    per_scene = {}
    tiling = capi.tiling(nranks, root_extra if nranks > 1 else 0)
    phase('isolated pass')
    note('timed region done: %.4f ms per step; isolated pass per scene' % (elapsed / args.steps * 1e3))
    for it in work:
        sctx = it['ctx']
        # (tiled: this rank's place in the scene's frame, counted from the rank that assembles it)
        r0 = ((0 if group_ranks else rank) - it.get('root', 0)) % nranks
        if tiled:
            it['slots'] = capi.tiles_owned_pixels(it['cam'], r0, tiling)
        dc = sctx.alloc(max(it['slots'], 1) * 32) if tiled else it['contacts']
        if tiled:
            trace = lambda it=it, dc=dc: it['bvh'].generate_contacts_primary_tiled_d(it['cam'], r0, tiling, dc)
        else:
            trace = lambda it=it, dc=dc: it['bvh'].generate_contacts_primary_d(it['cam'], dc)
        tm = capi.Timer(sctx)
        # untimed: the device idled during the fence above and its clocks need milliseconds of work to come back -- ten frames of
        # the Cube are 0.4 ms, and its timed launches then ran at a fraction of the clock (0.26 ms instead of 0.035)
        t_warm = time.perf_counter()
        while time.perf_counter() - t_warm < 0.005:
            for _ in range(ISOLATED_FRAMES):
                trace()
            sctx.synchronize()
        sctx.set_profiling(2)
        tm.start()
        for _ in range(ISOLATED_FRAMES):
            trace()
        tm.stop()
        ms = tm.elapsed_ms() / ISOLATED_FRAMES
        k = sctx.trace_kernel_times()
        sctx.set_profiling(False)
        fx = frames_fixture[it['name']][key]
        alg = 56 * fx['inner_visits'] + 40 * fx['leaf_tests'] + 32 * fx['rays']
        st = it['bvh'].search_tree()
        walked = alg
        if st['depth'] and st['used_by_last_camera']:
            sx = search_fixture[it['name']][key]
            walked = 56 * sx['inner_visits'] + 40 * sx['leaf_tests'] + 32 * sx['rays']
        ps = dict(trace_ms=round(ms, 4), first_kernel_ms=round(k['lane_phase_ms'] / ISOLATED_FRAMES, 4),
                  wave_per_ray_ms=round(k['wave_phase_ms'] / ISOLATED_FRAMES, 4), mrays_per_s=round(it['slots'] / ms / 1e3, 1),
                  inner_visits_per_ray=round(fx['inner_visits'] / fx['rays'], 2), leaf_tests_per_ray=round(fx['leaf_tests'] / fx['rays'], 2),
                  algorithmic_bytes=alg, walked_bytes=walked, search_tree_depth=st['depth'],
                  reference_tree_depth=it['bvh'].info()['max_depth'], build_ms=round(build_ms[it['name']], 3),
                  strategy=it['bvh'].trace_strategy())
        if it['name'] in first_frame_ms:
            ps['first_frame_ms'] = round(first_frame_ms[it['name']][0], 3)
            ps['first_frame_over_steady_frame'] = round(first_frame_ms[it['name']][0] / max(ms, 1e-6), 1)
            ps['calibration_launches'] = first_frame_ms[it['name']][1]
            ps['calibration_gpu_ms'] = round(first_frame_ms[it['name']][2], 3)
            ps['calibration_gpu_over_steady_frame'] = round(first_frame_ms[it['name']][2] / max(ms, 1e-6), 1)
        # the dynamic-scene frame (OpenCLRayTracerDemo.cpp:230-234: the structure is rebuilt per frame when the scene moves):
        # initializeFrame + construct + generateContacts + shade, every frame, with the calibrated strategy kept across rebuilds
        if not tiled and not args.no_extra_legs:
            K = 8

            def dyn(it=it, sctx=sctx):            # (rt_render_frame: contacts + colours as one eager sequence -- a frame graph does not survive construct())
                it['bvh'].initialize_frame(); it['bvh'].construct()
                capi.render_frame_d(it['bvh'], it['scene'], it['cam'], it['contacts'], it['rgba'])
            dyn(); dyn(); sctx.synchronize()
            tw = time.perf_counter()
            for _ in range(K):
                dyn()
            sctx.synchronize()
            ps['rebuild_plus_trace_ms'] = round((time.perf_counter() - tw) * 1e3 / K, 4)
            ps['strategy_after_rebuilds'] = it['bvh'].trace_strategy()
            if verified is not None:            # the frame of the BVH rebuilt K + 2 times
                verify_frame(sctx, it['contacts'], it['rgba'], npx, bfx['scenes'][it['name']][key]['default'], bfx['sample_stride'],
                             '%s, rebuild-per-frame leg' % it['name'])
                verified['rebuilt_frames'] = verified.get('rebuilt_frames', 0) + 1
        # ---- the other modes of rt_bvh_create on this scene: RT_MODE_FAST (the build a scene that is rebuilt every frame would ask for) and
        # RT_MODE_DEVICE_ARITH (dot / cross as the reference's device branch computes them on this GPU); each frame verified against what
        # the mode promises (FAST: hit mask and t of FIXED; DEVICE_ARITH: pixelIndex, materialIndex and t of the oracle's device arithmetic)
        if not tiled and not args.no_extra_legs:
            K = 8
            rec = bfx['scenes'][it['name']][key]['default'] if bfx is not None else None
            fast = capi.DeviceBVH(sctx, it['scene'], mode=2)
            for _ in range(3):
                fast.build()

            def dyn_fast():
                fast.initialize_frame(); fast.construct()
                capi.render_frame_d(fast, it['scene'], it['cam'], it['contacts'], it['rgba'])
            dyn_fast(); dyn_fast(); sctx.synchronize()
            tw = time.perf_counter()
            for _ in range(K):
                fast.build()
            sctx.synchronize()
            ps['fast_mode'] = dict(build_ms=round((time.perf_counter() - tw) * 1e3 / K, 4))
            tw = time.perf_counter()
            for _ in range(K):
                dyn_fast()
            sctx.synchronize()
            ps['fast_mode']['rebuild_plus_trace_ms'] = round((time.perf_counter() - tw) * 1e3 / K, 4)
            tw = time.perf_counter()
            for _ in range(K):
                it['bvh'].build()
            sctx.synchronize()
            ps['build_ms_rebuilt'] = round((time.perf_counter() - tw) * 1e3 / K, 4)       # FIXED, builds 2.. of an allocation (graph replays)
            if rec is not None and 't_sha256' in rec:
                dyn_fast(); sctx.synchronize()
                got = sctx.download(it['contacts'], npx, capi.CONTACT)
                if sha256_hex(np.ascontiguousarray(got['normalAndDistance'][:, 3])) != rec['t_sha256']:
                    raise SystemExit('VERIFICATION FAILED (%s, RT_MODE_FAST): hit mask / t differ from FIXED\'s' % it['name'])
                ps['fast_mode']['verified'] = 'hit mask and t equal FIXED\'s'
                verified['fast_mode_frames'] = verified.get('fast_mode_frames', 0) + 1
            fast.close()
            da = capi.DeviceBVH(sctx, it['scene'], mode=0x100).build()
            da.generate_contacts_primary_d(it['cam'], it['contacts']); sctx.synchronize()         # calibrates
            tm2 = capi.Timer(sctx)
            for _ in range(ISOLATED_FRAMES):
                da.generate_contacts_primary_d(it['cam'], it['contacts'])
            tm2.start()
            for _ in range(ISOLATED_FRAMES):
                da.generate_contacts_primary_d(it['cam'], it['contacts'])
            tm2.stop()
            ps['device_arith_mode'] = dict(trace_ms=round(tm2.elapsed_ms() / ISOLATED_FRAMES, 4), strategy=da.trace_strategy())
            tm2.close()
            if rec is not None and 'device_arith' in rec:
                got = sctx.download(it['contacts'], npx, capi.CONTACT)
                t_bits = np.ascontiguousarray(got['normalAndDistance'][:, 3]).view(np.uint32)
                keyb = np.stack([got['pixelIndex'], got['materialIndex'], t_bits], axis=1).astype('<u4')
                if sha256_hex(keyb) != rec['device_arith']['key_sha256']:
                    raise SystemExit('VERIFICATION FAILED (%s, RT_MODE_DEVICE_ARITH): pixelIndex / materialIndex / t differ from the oracle\'s device arithmetic' % it['name'])
                ps['device_arith_mode']['verified'] = 'pixelIndex, materialIndex, t equal the oracle with the device branch\'s dot / cross'
                verified['device_arith_frames'] = verified.get('device_arith_frames', 0) + 1
            da.close()
        tm.close()
        if tiled:
            sctx.free(dc)
        per_scene[it['name']] = ps

    # ---- the camera that moves (RTFrame, OpenCLRayTracerDemo.cpp:271-326; the camera is updated before every frame, :275-276) --------
    moving = None
    if not tiled and use_graphs and not args.no_extra_legs:
        M, R = 32, 4
        note('moving-camera leg')
        pose_bits_replaced = 0
        for it in work:
            it['poses'] = moving_cameras(it['bytes'], RES[0], RES[1], M)
            if verified is not None:
                # the poses that get verified are the fixture's cameras, bit for bit (this host's sin / cos may round another way)
                for kname, rec in bfx['scenes'][it['name']][key].items():
                    if kname.startswith('pose'):
                        k = int(kname[4:])
                        fc = camera_from_bits(rec['camera_bits'])
                        if it['poses'][k:k + 1].tobytes() != fc.tobytes():
                            pose_bits_replaced += 1
                        it['poses'][k] = fc[0]
            it['graph'].close()                             # the rebuild-per-frame leg above has rebuilt the BVH: a graph of its own
            it['ctx'].set_frames_in_flight(len(SCENES))     # five frames in flight, as in the timed region
            it['graph'] = capi.FrameGraph(it['bvh'], it['scene'], it['cam'], it['contacts'], it['rgba'])
            it['ctx'].set_frames_in_flight(1)
        def pose(it, k):
            return it['poses'][k % M:k % M + 1]
        per = {}
        for it in work:
            sctx, g = it['ctx'], it['graph']
            for k in range(M):                              # untimed: host-side plane table, clocks
                g.set_camera(pose(it, k)); g.launch()
            sctx.synchronize()
            kinds = collections.Counter()
            t1 = time.perf_counter()
            for k in range(M):
                g.set_camera(pose(it, k)); g.launch(); kinds[g.last_launch()] += 1
            sctx.synchronize()
            mv = (time.perf_counter() - t1) * 1e3 / M
            still = 0.0
            for k in range(M):                              # the same poses with a camera that stands still: R replays each
                g.set_camera(pose(it, k)); g.launch(); sctx.synchronize()
                t1 = time.perf_counter()
                for _ in range(R):
                    g.launch()
                sctx.synchronize()
                still += (time.perf_counter() - t1) * 1e3 / R
            still /= M
            t1 = time.perf_counter()
            for k in range(M):
                it['bvh'].generate_contacts_primary_d(pose(it, k), it['contacts'])
                capi.shade_blinn_phong_frame_d(sctx, it['scene'], pose(it, k), it['contacts'], it['rgba'])
            sctx.synchronize()
            eager = (time.perf_counter() - t1) * 1e3 / M
            t1 = time.perf_counter()
            for k in range(M):
                it['bvh'].generate_contacts_primary_d(pose(it, 1), it['contacts'])
                capi.shade_blinn_phong_frame_d(sctx, it['scene'], pose(it, 1), it['contacts'], it['rgba'])
            sctx.synchronize()
            eager_still = (time.perf_counter() - t1) * 1e3 / M
            note('  moving camera, %s: %.4f ms per frame moving, %.4f standing still' % (it['name'], mv, still))
            per[it['name']] = dict(moving_graph_ms_per_frame=round(mv, 4), same_poses_still_graph_ms_per_frame=round(still, 4),
                                   moving_over_still=round(mv / max(still, 1e-9), 3), moving_eager_ms_per_frame=round(eager, 4),
                                   still_eager_ms_per_frame=round(eager_still, 4),
                                   launches=dict(with_preparation=kinds[1], steady=kinds[0], eager=kinds[2]))
            g.set_camera(it['cam'])
        # the step of five frames on five streams, every frame a new pose
        note('  moving camera, five streams')
        K = max(64, min(args.steps, 256))
        for k in range(M):
            for it in work:
                it['graph'].set_camera(pose(it, k)); it['graph'].launch()
        fence()
        t1 = time.perf_counter()
        for k in range(K):
            for it in work:
                it['graph'].set_camera(pose(it, k)); it['graph'].launch()
        issued_mv = time.perf_counter() - t1
        fence()
        mv_step = (time.perf_counter() - t1) * 1e3 / K
        still_step = 0.0
        for k in range(M):
            for it in work:
                it['graph'].set_camera(pose(it, k)); it['graph'].launch()
            fence()
            t1 = time.perf_counter()
            for _ in range(R):
                for it in work:
                    it['graph'].launch()
            fence()
            still_step += (time.perf_counter() - t1) * 1e3 / R
        still_step /= M
        t1 = time.perf_counter()
        for k in range(K):
            for it in work:
                it['bvh'].generate_contacts_primary_d(pose(it, k), it['contacts'])
                capi.shade_blinn_phong_frame_d(it['ctx'], it['scene'], pose(it, k), it['contacts'], it['rgba'])
        fence()
        eager_step = (time.perf_counter() - t1) * 1e3 / K
        check_errors()
        moved_verified = 0
        if verified is not None:
            # frames of a camera that moved, rendered through the graph path (set_camera + launch) and through the eager calls,
            # against the oracle's frames of exactly these cameras (pose 0 = the default camera)
            for it in work:
                recs = {0: bfx['scenes'][it['name']][key]['default']}
                recs.update({int(kn[4:]): r for kn, r in bfx['scenes'][it['name']][key].items() if kn.startswith('pose')})
                for k, rec in sorted(recs.items()):
                    it['graph'].set_camera(pose(it, k)); it['graph'].launch(); it['ctx'].synchronize()
                    verify_frame(it['ctx'], it['contacts'], it['rgba'], npx, rec, bfx['sample_stride'], '%s, moving camera pose %d (graph)' % (it['name'], k))
                    it['bvh'].generate_contacts_primary_d(pose(it, k), it['contacts'])
                    capi.shade_blinn_phong_frame_d(it['ctx'], it['scene'], pose(it, k), it['contacts'], it['rgba'])
                    it['ctx'].synchronize()
                    verify_frame(it['ctx'], it['contacts'], it['rgba'], npx, rec, bfx['sample_stride'], '%s, moving camera pose %d (eager)' % (it['name'], k))
                    moved_verified += 2
            verified['moving_camera_frames'] = moved_verified
            verified['moving_camera_poses_taken_from_fixture_bits'] = pose_bits_replaced
        for it in work:
            it['graph'].set_camera(it['cam'])
        moving = dict(what='every frame a new camera position and orientation (%d poses around the default camera, pose 0 = the default camera); '
                           'k_prepare_camera inside the timed region; graph = rt_frame_graph_set_camera + rt_frame_graph_launch, '
                           'eager = generateContacts(Camera&) + shade' % M,
                      step_ms=round(mv_step, 4), mrays_per_s=round(rays_per_step / mv_step / 1e3, 1),
                      same_poses_still_step_ms=round(still_step, 4), moving_over_still_step=round(mv_step / max(still_step, 1e-9), 3),
                      eager_step_ms=round(eager_step, 4), host_issue_ms_per_step=round(issued_mv / K * 1e3, 4), steps=K, per_scene=per)

    # ---- config 5 (BASELINE.json configs[4]): the Sponza-class stand-in at 3840x2160, BVH, the frame tiled over the ranks --------
    config5 = None
    fx5 = None
    if atrium_bytes is not None:
        note('config 5 leg')
        phase('config 5')
        cam5 = demo_camera(atrium_bytes, *RES5)
        ranks_here = range(nranks) if group_ranks else [rank]
        c5 = [capi.Context(local_rank) for _ in ranks_here]
        s5 = [capi.DeviceScene(c, atrium_bytes) for c in c5]
        tb = time.perf_counter()
        b5 = [capi.DeviceBVH(c, s).build() for c, s in zip(c5, s5)]
        build5 = (time.perf_counter() - tb) * 1e3 / len(b5)
        K5 = 10
        if tiled:
            comms5 = capi.Comm.create_group(c5) if group_ranks else [main_comm.share(c5[0])]
            f5 = [capi.MultiFrame(c, b, s, cam5, root_extra) for c, b, s in zip(comms5, b5, s5)]
            render5 = lambda: capi.multi_frame_render(f5)
        else:
            d5c, d5o = c5[0].alloc(RES5[0] * RES5[1] * 32), c5[0].alloc(RES5[0] * RES5[1] * 16)
            c5[0].synchronize()
            tf = time.perf_counter()
            b5[0].generate_contacts_primary_d(cam5, d5c)
            c5[0].synchronize()
            first5 = (time.perf_counter() - tf) * 1e3
            g5 = capi.FrameGraph(b5[0], s5[0], cam5, d5c, d5o)
            render5 = g5.launch
        render5(); render5(); fence()
        t5 = time.perf_counter()
        for _ in range(K5):
            render5()
        fence()
        e5 = time.perf_counter() - t5
        if tiled:
            for f in f5:
                f.wait()
        else:
            c5[0].synchronize()
        if dist is not None:
            t = torch.tensor([e5], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            e5 = float(t.item())
        if verified is not None:
            if sha256_hex(np.frombuffer(atrium_bytes, dtype=np.uint8)) != bfx['atrium']['scene_sha256']:
                verified['config5'] = 'NOT verified: this host generated other bytes for the procedural stand-in scene than the fixture\'s host'
            else:
                fx5 = bfx['atrium']
                rec5 = fx5['%dx%d' % RES5]['default']
                if tiled:
                    root5 = f5[0] if (group_ranks or rank == 0) else None
                    if root5 is not None:
                        root5.wait()
                        msg = rgba_mismatch(root5.download_frame(), rec5, bfx['sample_stride'])
                        if msg:
                            raise SystemExit('VERIFICATION FAILED (config 5, assembled frame): %s' % msg)
                else:
                    verify_frame(c5[0], d5c, d5o, RES5[0] * RES5[1], rec5, bfx['sample_stride'], 'config 5, 3840x2160')
                verified['config5'] = 'frame equals the oracle\'s'
        config5 = dict(workload='Sponza-class stand-in (assets/synth, %d triangles) x 3840x2160 primary rays, BVH trace + shade, frame tiled over %d rank(s), '
                                'one RCCL gather per frame' % (s5[0].n_tris, nranks), frames=K5, ms_per_frame=round(e5 / K5 * 1e3, 3),
                       mrays_per_s=round(RES5[0] * RES5[1] * K5 / e5 / 1e6, 1), build_ms=round(build5, 2), reference_tree_depth=b5[0].info()['max_depth'],
                       search_tree_depth=b5[0].search_tree()['depth'], strategy=b5[0].trace_strategy())
        if not tiled:
            config5['first_frame_ms'] = round(first5, 2)
            config5['calibration_launches'] = b5[0].calibration_launches()
            config5['calibration_gpu_ms'] = round(b5[0].calibration_gpu_ms(), 2)
            config5['calibration_gpu_over_steady_frame'] = round(b5[0].calibration_gpu_ms() / max(e5 / K5 * 1e3, 1e-6), 1)
            config5['first_frame_over_steady_frame'] = round(first5 / max(e5 / K5 * 1e3, 1e-6), 1)

    # ---- the ray-buffer entry point: generateContacts(CLBuffer& rays, CLBuffer& contacts, n), BVHManager.cpp:292-329 -----------------
    ray_buffer = None
    if not tiled and not args.no_extra_legs and not args.no_verify:
        note('ray-buffer leg')
        ray_buffer = dict(what='rt_bvh_generate_contacts_rays at 1920x1080: the camera\'s rays handed over as Ray[] (primary), the reflection bounce and '
                               'the shadow rays towards the light, generated on the device from the primary contacts; rays and contacts equal the oracle\'s',
                          per_scene={})
        for it in work:
            ray_buffer['per_scene'][it['name']] = ray_buffer_leg(it['ctx'], it['bvh'], it['cam'], bfx['scenes'][it['name']][key]['ray_buffer'])
            note('  ray buffer, %s: %s' % (it['name'], {k: v['mrays_per_s'] for k, v in ray_buffer['per_scene'][it['name']].items()}))
        if atrium_bytes is not None and fx5 is not None:
            ray_buffer['per_scene']['atrium'] = ray_buffer_leg(c5[0], b5[0], demo_camera(atrium_bytes, *RES), fx5[key]['ray_buffer'])
            note('  ray buffer, atrium: %s' % {k: v['mrays_per_s'] for k, v in ray_buffer['per_scene']['atrium'].items()})
        for leg in ('primary', 'reflect', 'shadow'):
            ray_buffer['min_mrays_per_s_' + leg] = min(v[leg]['mrays_per_s'] for v in ray_buffer['per_scene'].values())
        verified['ray_buffer_legs'] = 3 * len(ray_buffer['per_scene'])

    # ---- BASELINE.json configs[2] and configs[3] on this GPU (N = 1): the Bunny-class stand-in through the BVH and the Sponza-class stand-in through
    # the Two-Level Grid (build + trace + shade) at 1920x1080, each frame verified like the others
    config3 = config4 = None
    if not tiled and not args.no_extra_legs:
        note('configs 3 and 4')
        phase('configs 3 and 4')
        xctx = capi.Context(local_rank)
        tmx = capi.Timer(xctx)
        d_c, d_o = xctx.alloc(npx * 32), xctx.alloc(npx * 16)
        if bunny_bytes is not None:
            cam3 = demo_camera(bunny_bytes, *RES)
            s3 = capi.DeviceScene(xctx, bunny_bytes)
            t1 = time.perf_counter()
            b3 = capi.DeviceBVH(xctx, s3).build()
            build3 = (time.perf_counter() - t1) * 1e3
            b3.generate_contacts_primary_d(cam3, d_c); xctx.synchronize()
            g3 = capi.FrameGraph(b3, s3, cam3, d_c, d_o)
            for _ in range(5):
                g3.launch()
            tmx.start()
            for _ in range(20):
                g3.launch()
            tmx.stop()
            ms3 = tmx.elapsed_ms() / 20
            config3 = dict(workload='Bunny-class stand-in (assets/synth, %d triangles) x 1920x1080 primary rays, BVH trace + shade' % s3.n_tris,
                           ms_per_frame=round(ms3, 4), mrays_per_s=round(npx / ms3 / 1e3, 1), first_build_ms=round(build3, 3), reference_tree_depth=b3.info()['max_depth'],
                           search_tree_depth=b3.search_tree()['depth'], strategy=b3.trace_strategy())
            if verified is not None:
                if sha256_hex(np.frombuffer(bunny_bytes, dtype=np.uint8)) != bfx.get('bunny', {}).get('scene_sha256'):
                    verified['config3'] = 'NOT verified: no digest for the stand-in scene as generated on this host'
                else:
                    xctx.synchronize()
                    verify_frame(xctx, d_c, d_o, npx, bfx['bunny'][key]['default'], bfx['sample_stride'], 'config 3, Bunny-class BVH frame')
                    verified['config3'] = 'frame equals the oracle\'s'
            g3.close(); b3.close(); s3.close()
        if atrium_bytes is not None:
            cam4 = demo_camera(atrium_bytes, *RES)
            s4 = capi.DeviceScene(xctx, atrium_bytes)
            g4 = capi.DeviceTLG(xctx, s4).build()
            xctx.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                g4.build()
            xctx.synchronize()
            build4 = (time.perf_counter() - t1) * 1e3 / 3
            for _ in range(3):                          # (warm: camera-relative records written, clocks up)
                g4.generate_frame_d(cam4, d_c, d_o)
            xctx.synchronize()
            tmx.start()
            for _ in range(20):
                g4.generate_frame_d(cam4, d_c, d_o)     # rt_tlg_generate_frame: the trace kernel shades what it stores, as the BVH's captured frames do
            tmx.stop()
            ms4 = tmx.elapsed_ms() / 20
            tmx.start()
            for _ in range(5):
                g4.generate_contacts_primary_d(cam4, d_c)
                capi.shade_blinn_phong_frame_d(xctx, s4, cam4, d_c, d_o)
            tmx.stop()
            ms4_two = tmx.elapsed_ms() / 5
            g4.generate_frame_d(cam4, d_c, d_o)         # (the frame that is verified below: the fused one)
            config4 = dict(workload='Sponza-class stand-in (assets/synth, %d triangles) x 1920x1080 primary rays, Two-Level Grid (densities 2.0 / 2.0) build + trace + shade' % s4.n_tris,
                           build_ms=round(build4, 3), trace_plus_shade_ms=round(ms4, 4), mrays_per_s_grid_kept=round(npx / ms4 / 1e3, 1),
                           trace_then_shade_two_launches_ms=round(ms4_two, 4),
                           mrays_per_s_rebuilt_every_frame=round(npx / (ms4 + build4) / 1e3, 1), grid=g4.info())
            if verified is not None and fx5 is not None and 'tlg' in fx5[key]:
                xctx.synchronize()
                verify_frame(xctx, d_c, d_o, npx, fx5[key]['tlg'], bfx['sample_stride'], 'config 4, Two-Level Grid frame')
                verified['config4'] = 'frame equals the oracle\'s'
            g4.close(); s4.close()
        xctx.free(d_c); xctx.free(d_o); tmx.close(); xctx.close()

    phase('done')
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    # ---- rooflines -------------------------------------------------------------------------------------------------------------
    share = (work[0]['slots'] / float(sum(capi.tiles_owned_pixels(work[0]['cam'], r, tiling) for r in range(nranks)))) if tiled else 1.0
    alg_per_launch = sum(per_scene[n]['algorithmic_bytes'] for n in SCENES) / len(SCENES) * share
    walked_per_launch = sum(per_scene[n]['walked_bytes'] for n in SCENES) / len(SCENES) * share
    trace_ms_avg = sum(per_scene[n]['first_kernel_ms'] + per_scene[n]['wave_per_ray_ms'] for n in SCENES) / len(SCENES)
    launch_s = trace_ms_avg * 1e-3
    traffic = None
    rv = None
    if pmc is not None:
        # FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE tallies the 128-byte requests of wide reads at 64 bytes: doubled
        # (MI355X_MICROARCH.md, HBM section).  Average over the five scenes, per traversal launch, like the duration.
        # ... over the scenes whose counter pass ran the launch that is timed here (the child calibrates on its own: ADVICE r4)
        matched = [n for n in SCENES if same_launch(pmc['strategies'].get(n), per_scene[n]['strategy'])] or list(SCENES)
        fetch = sum(pmc['per_scene'][n].get('FETCH_SIZE', 0.0) for n in matched) / len(matched)
        write = sum(pmc['per_scene'][n].get('WRITE_SIZE', 0.0) for n in matched) / len(matched)
        traffic = int((2.0 * fetch + write) * 1024)
        launch_s = sum(per_scene[n]['first_kernel_ms'] + per_scene[n]['wave_per_ray_ms'] for n in matched) / len(matched) * 1e-3
        trace_ms_avg = launch_s * 1e3
        alg_per_launch = sum(per_scene[n]['algorithmic_bytes'] for n in matched) / len(matched) * share
        walked_per_launch = sum(per_scene[n]['walked_bytes'] for n in matched) / len(matched) * share
        for n in SCENES:
            c = pmc['per_scene'][n]
            per_scene[n]['counters_per_launch'] = {k: round(v, 1) for k, v in c.items()}
            per_scene[n]['valu_issue_frac'] = round(valu_issue(c), 4) if valu_issue(c) is not None else None
            per_scene[n]['hbm_bytes_per_launch'] = int((2.0 * c.get('FETCH_SIZE', 0.0) + c.get('WRITE_SIZE', 0.0)) * 1024)
            per_scene[n]['strategy_in_counter_pass'] = pmc['strategies'][n]
        # the kernel that takes the most device time in the isolated launches: by GRBM_GUI_ACTIVE summed over its dispatches
        dom = max(pmc['per_kernel'], key=lambda k: pmc['per_kernel'][k].get('GRBM_GUI_ACTIVE', 0.0))
        dk = pmc['per_kernel'][dom]
        nd = max(pmc['kernel_dispatches'].get(dom, 1), 1)
        rv = dict(bound='valu_issue', kernel=dom, achieved=round(dk.get('SQ_INSTS_VALU', 0.0) * VALU_CYCLES_PER_INST / nd / 1e6, 3),
                  peak=round(SIMDS * dk.get('GRBM_GUI_ACTIVE', 0.0) / 8.0 / nd / 1e6, 3), unit='M SIMD issue cycles per dispatch',
                  frac=round(valu_issue(dk), 4), dispatches=nd,
                  formula='SQ_INSTS_VALU x %.0f cycles / (%d SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)' % (VALU_CYCLES_PER_INST, SIMDS),
                  counters_per_dispatch={k: round(v / nd, 1) for k, v in dk.items()},
                  salu_per_valu=round(dk.get('SQ_INSTS_SALU', 0.0) / max(dk.get('SQ_INSTS_VALU', 1.0), 1.0), 3),
                  wait_inst_any_over_wave_cycles=round(dk.get('SQ_WAIT_INST_ANY', 0.0) / max(dk.get('SQ_WAVE_CYCLES', 1.0), 1.0), 4),
                  all_kernels={k: dict(dispatches=pmc['kernel_dispatches'].get(k, 0), valu_issue_frac=(round(valu_issue(v), 4) if valu_issue(v) else None),
                                       salu_per_valu=round(v.get('SQ_INSTS_SALU', 0.0) / max(v.get('SQ_INSTS_VALU', 1.0), 1.0), 3))
                               for k, v in pmc['per_kernel'].items()},
                  source='rocprofv3 --pmc passes run by this bench invocation (child processes, %d isolated launches per scene)' % PMC_FRAMES)
    # without the counters: the bytes a launch cannot avoid -- one 32-byte contact per ray -- are the stated lower bound
    compulsory = 32.0 * npx * share
    moved = traffic if traffic is not None else compulsory
    roofline = dict(bound='hbm', achieved=round(moved / launch_s / 1e9, 1), peak=HBM_PEAK_GBPS, unit='GB/s',
                    frac=round(moved / launch_s / 1e9 / HBM_PEAK_GBPS, 4), traffic=traffic,
                    achieved_is=('measured HBM bytes per traversal launch (2 x FETCH_SIZE + WRITE_SIZE, rocprofv3 passes of this run) / avg_launch_ms'
                                 if traffic is not None else 'no counters in this run: the 32-byte contacts a launch must write / avg_launch_ms (lower bound)'),
                    kernel='traversal launch of one frame: k_trace_packet4, k_trace_packet or k_trace, then k_trace_long (+ k_resolve_split) '
                           'where work was handed over; <1, 2> in profiles = these isolated launches, <1, 0> = the timed region, <1, 1> = calibration',
                    avg_launch_ms=round(trace_ms_avg, 4), isolated_launches=ISOLATED_FRAMES * len(SCENES),
                    scenes_averaged=(matched if traffic is not None else list(SCENES)),
                    reference_algorithm_bytes_per_launch=int(alg_per_launch), walked_bytes_per_launch=int(walked_per_launch),
                    reference_algorithm_gbps=round(alg_per_launch / launch_s / 1e9, 1), walked_gbps=round(walked_per_launch / launch_s / 1e9, 1),
                    traffic_over_reference_algorithm=(round(traffic / alg_per_launch, 4) if traffic else None),
                    binds='not HBM: the records are L2 / scalar-cache resident and a packet fetches a record once for up to 64 rays, so the '
                          'reference algorithm\'s bytes per second (reference_algorithm_gbps) exceed the HBM peak and are no fraction of it; '
                          'see roofline_valu_issue')

    # the roof that does bound the path: what a step MUST move through HBM whatever the algorithm -- one 32-byte Contact and one 16-byte
    # RGBA per ray (the records are cache resident; SURVEY.md 8(d)'s 56 I + 40 L bytes are no HBM traffic, see `binds`)
    compulsory_step = 48.0 * rays_per_step
    floor_step_ms = compulsory_step / (HBM_PEAK_GBPS * 1e9) * 1e3
    roofline['compulsory_bytes_per_step'] = int(compulsory_step)
    roofline['output_floor_ms_per_step'] = round(floor_step_ms, 5)
    roofline['frac_of_output_floor'] = round(floor_step_ms / (elapsed / args.steps * 1e3), 4)
    roofline['frac_of_output_floor_is'] = ('(48 B per ray x rays per step / 8 TB/s) / ms_per_step of the timed region: 1.0 = the step takes exactly as long '
                                           'as writing its contacts and colours at the HBM peak; no algorithm can exceed it')
    bound_per_scene = {}
    for n in SCENES:
        q = per_scene[n]
        floor_ms = 32.0 * q_slots(work, n) / (HBM_PEAK_GBPS * 1e9) * 1e3          # an isolated traversal launch writes the contacts only
        f_floor = floor_ms / max(q['trace_ms'], 1e-9)
        f_hbm = (q['hbm_bytes_per_launch'] / (q['trace_ms'] * 1e-3) / 1e9 / HBM_PEAK_GBPS) if 'hbm_bytes_per_launch' in q else None
        f_valu = q.get('valu_issue_frac')
        if 'strategy_in_counter_pass' in q and not same_launch(q['strategy_in_counter_pass'], q['strategy']):
            # the profiled child calibrated on its own and settled on another launch than the one timed here: its counters divided by this
            # process's duration would mix two launches -- no derived fraction for this scene
            bound_per_scene[n] = dict(frac_of_output_floor=round(f_floor, 4), hbm_frac_measured=None, valu_issue_frac=None,
                                      binds='not derived: the counter pass ran %s, the timed launch %s' % (q['strategy_in_counter_pass'], q['strategy']))
            q['counters_are_of_another_strategy'] = True
            continue
        cands = [('hbm (output writes)', f_hbm if f_hbm is not None else f_floor), ('valu_issue', f_valu or 0.0)]
        top = max(cands, key=lambda c: c[1])
        bound_per_scene[n] = dict(frac_of_output_floor=round(f_floor, 4), hbm_frac_measured=(round(f_hbm, 4) if f_hbm is not None else None),
                                  valu_issue_frac=f_valu,
                                  binds=(top[0] if top[1] >= 0.3 else 'latency of a wave step x steps of the heaviest 8x8 block (no roof above 30 %%; nearest: %s %.2f)' % top))
    roofline['per_scene'] = bound_per_scene

    cpu_baseline = None
    if world == 1 and not tiled and not args.no_cpu_baseline:
        note('CPU baseline')
        sys.path.insert(0, os.path.join(ROOT, 'oracle'))
        import oracle as orc                     # cpu_baseline leg only: the oracle is the thing timed here
        cores = orc.ncpu()
        t_cpu, rays_cpu = 0.0, 0
        c_buf = np.zeros(npx, dtype=orc.CONTACT)              # output buffers live outside the timed region, pages touched
        rgba_buf = np.zeros((npx, 4), dtype=np.float32)
        for name in SCENES:
            osc = orc.Scene(scene_bytes[name])
            ob = orc.BVH(osc, 0)
            cam = osc.demo_camera(*RES)
            t1 = time.perf_counter()
            c, _ = ob.trace_primary(cam, nthreads=cores, out=c_buf)
            orc.shade_frame(osc, cam, c, nthreads=cores, out=rgba_buf)
            t_cpu += time.perf_counter() - t1
            rays_cpu += len(c)
        # the same path on ONE thread (SURVEY.md 8(d) asks for both), on a bounded sample: 320x180 frames of the five scenes
        t_one, rays_one = 0.0, 0
        for name in SCENES:
            osc = orc.Scene(scene_bytes[name])
            ob = orc.BVH(osc, 0)
            cam = osc.demo_camera(320, 180)
            t1 = time.perf_counter()
            c, _ = ob.trace_primary(cam, nthreads=1)
            orc.shade_frame(osc, cam, c, nthreads=1)
            t_one += time.perf_counter() - t1
            rays_one += len(c)
        model = 'unknown'
        try:
            for ln in open('/proc/cpuinfo'):
                if ln.startswith('model name'):
                    model = ln.split(':', 1)[1].strip()
                    break
        except OSError:
            pass
        cpu_baseline = dict(value=round(rays_cpu / t_cpu / 1e6, 3), unit='Mrays/s', cores=cores, kind='port',
                            sample='all five scenes, one full 1920x1080 frame each (trace + shade), BVH build excluded',
                            single_thread_value=round(rays_one / t_one / 1e6, 4),
                            single_thread_sample='the five scenes at 320x180 (trace + shade), one thread', cpu_model=model)

    par = 'single GPU, the five frames of a step on five HIP streams'
    if tiled:
        par = ('64x64 frame tiles dealt over %d ranks (the rank that assembles a frame owns %d extra per period of %d tiles), ONE gather of shaded colour per '
               'frame (12 B per pixel, grouped RCCL send/recv issued by librtmi) into that rank: scene i is assembled on rank i mod N (roots %s), '
               'the five frames of a step on five HIP streams per rank'
               % (nranks, root_extra, nranks + root_extra, [it['root'] for it in work]))
        par += '; RCCL communicator: %s' % comm_mode
        if group_ranks:
            par += '; REHEARSAL: all ranks in one process on one device, gather by device copies'
    out = dict(metric='Mrays/s (primary rays, BVH trace + shade, bundled TestModels, 1920x1080)', value=round(value, 1),
               unit='Mrays/s', n_gpus=world, steps=args.steps, warmup=args.warmup,
               ms_per_step=round(elapsed / args.steps * 1e3, 4), higher_is_better=True, scaling='strong',
               vs_baseline=None, dtype='f32', data='bundled TestModels OBJ scenes (shipped with the reference), demo default camera',
               config=dict(workload='5 bundled TestModels scenes x 1920x1080 primary rays, BVH traversal + Blinn-Phong shading',
                           rays_per_step=rays_per_step, resolution=list(RES), scenes=SCENES, parallelism=par, ranks=nranks,
                           settle='%d untimed steps (%.0f ms) before the %d warm-up steps: device clocks (--settle-ms)' % (settled, args.settle_ms, args.warmup),
                           policy='timed region and moving-camera leg: rt_ctx_set_frames_in_flight(5); isolated frames, first frames, config5: the default (one frame at a time)'),
               ms_per_frame=round(elapsed / args.steps / len(SCENES) * 1e3, 4),
               host_issue_ms_per_step=round(issued / args.steps * 1e3, 4),
               verified=verified, roofline=roofline, roofline_valu_issue=rv, cpu_baseline=cpu_baseline, moving_camera=moving, ray_buffer=ray_buffer, config3=config3, config4=config4, config5=config5,
               per_rank=per_rank, per_scene=per_scene)
    line = compact_line(out, args.details)
    if args.details:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(args.details)), exist_ok=True)
            with open(args.details, 'w') as f:
                json.dump(out, f)
                f.write('\n')
        except OSError as e:
            print('bench: details file %s not written: %s' % (args.details, e), file=sys.stderr)
            line['details_file'] = None
    print(json.dumps(out if args.full_line else line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
