#!/usr/bin/env python3
"""bench.py -- Mrays/s of the BVH primary-ray path (trace + Blinn-Phong shade) on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

One STEP = one frame of EACH of the five bundled TestModels scenes (Cube, Gardenlight, Tenryuu, ManFigure,
ClumsyDragon; BASELINE.json configs[1]) at 1920x1080, demo default camera: BVH traversal (two HIP kernels) + shading.
Scene buffers, BVHs and output buffers are resident in HBM before the timed region; BVH build time is reported
separately.  The five frames of a step are independent and are issued on five HIP streams (one library context per
scene), so one frame's tail is filled by the next frame's blocks.  With N > 1 every frame is cut into 64x64-pixel tiles dealt round-robin to the ranks (one process per GPU);
each rank traces and shades its tiles into a compact buffer, ONE gather (RCCL over xGMI) brings them to rank 0, which
scatters them back to pixel order -- total work per step is fixed, so scaling is "strong".

The line printed by rank 0 also carries
  roofline      algorithmic bytes of the traversal (56*I + 40*L + 32 per ray; I / L = inner records walked / triangles
                tested, from the committed counter fixtures tests/golden/search_tree_frames.json for the hierarchy the
                kernels walk and tests/golden/frames.json for the reference's own traversal order, both computed by the
                oracle, never taken from the GPU run) divided by the traversal kernels' device time, measured with HIP
                events on the stream they run on, vs 8 TB/s HBM;
  cpu_baseline  the CPU oracle (a restatement of the reference's host path) timed on this box's host cores on the same
                five frames (rank 0, N = 1 only).  A reported baseline, not the target.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# five streams per process: give each its own hardware queue (the HIP runtime multiplexes streams onto 4 by default)
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

SCENES = ['cube', 'gardenlight', 'tenryuu', 'man', 'clumsydragon']
RES = (1920, 1080)
HBM_PEAK_GBPS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def load_scene_bytes(name):
    """Packed scene buffer through the PRODUCT's loader: the C++ Scene::load packing (apps/bin/scene_pack_tool,
    openclraytracer_amd/csrc/host/src/SceneLoader.cpp)."""
    import subprocess
    import tempfile
    tool = os.path.join(ROOT, 'apps', 'bin', 'scene_pack_tool')
    if not os.path.exists(tool):
        raise SystemExit('apps/bin/scene_pack_tool is not built: run __graft_entry__.build()')
    with tempfile.NamedTemporaryFile(suffix='.bin') as tmp:
        subprocess.check_call([tool, os.path.join('assets', 'scenes', name + '.txt'), tmp.name], cwd=ROOT)
        return open(tmp.name, 'rb').read()


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


ISOLATED_FRAMES = 10         # frames per scene of the isolated (one scene at a time) measurement pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-graphs', action='store_true', help='issue every frame eagerly instead of replaying a HIP graph per frame')
    ap.add_argument('--check-frames', action='store_true',
                    help='N > 1: after the warm-up, rank 0 also traces every frame untiled and compares it with the gathered one')
    args = ap.parse_args()

    import torch
    from openclraytracer_amd import capi, frame_tiles

    world =int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit('launch with torch.distributed.run --nproc-per-node %d (WORLD_SIZE=%d)' % (args.gpus, world))
    # RTMI_BENCH_FORCE_TILED=1 (tests): the N > 1 code path -- tiled kernels, RCCL gather issued from the frame streams,
    # untile -- with a process group of ONE rank, i.e. on real RCCL on a one-GPU box
    force_tiled = world == 1 and os.environ.get('RTMI_BENCH_FORCE_TILED') == '1'
    if force_tiled:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29531')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
    tiled = world > 1 or force_tiled
    # RT_TILING word: with N > 1 rank 0, which gathers the frame, owns `root_extra` more tiles per period of N + root_extra
    # tiles (default 1: two tiles of three at N = 2), so less of the frame has to cross its links
    root_extra = int(os.environ.get('RTMI_BENCH_ROOT_EXTRA', '1' if world > 1 else '0'))
    tiling = capi.tiling(world, root_extra)
    dist = None
    if tiled:
        import torch.distributed as dist
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        # RTMI_BENCH_REHEARSAL=1: all ranks share GPU 0 and talk over gloo -- a functional rehearsal of the N > 1 path on a
        # one-GPU box (RCCL refuses two ranks on one device); its timings mean nothing
        rehearsal = os.environ.get('RTMI_BENCH_REHEARSAL') == '1'
        if rehearsal:
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group('gloo')
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)

    # One context = one HIP stream per scene: the five frames of a step are independent, so their kernels run side by
    # side and one frame's tail (a few long rays, a few busy waves) is filled by the next frame's blocks.  torch sees each
    # stream as an ExternalStream, which orders the RCCL gather of a frame behind that frame's kernels.
    ctxs = [capi.Context(local_rank) for _ in SCENES]   # raises without a gfx950 device: no fallback
    ctx = ctxs[0]
    tstreams = [torch.cuda.ExternalStream(c.get_stream(), device=dev) for c in ctxs]

    frames_fixture = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'frames.json')))
    search_fixture = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'search_tree_frames.json')))
    key = '%dx%d' % RES
    npx = RES[0] * RES[1]
    work = []
    build_ms = {}
    for si, name in enumerate(SCENES):
        buf = load_scene_bytes(name)
        sctx = ctxs[si]
        scene = capi.DeviceScene(sctx, buf)
        bvh = capi.DeviceBVH(sctx, scene)
        t0 = time.perf_counter()
        bvh.build()
        bvh.build()
        build_ms[name] = (time.perf_counter() - t0) * 1e3 / 2
        cam = demo_camera(buf, *RES)
        slots = capi.tiles_owned_pixels(cam, rank, tiling) if tiled else npx
        item = dict(name=name, bytes=buf, ctx=sctx, tstream=tstreams[si], scene=scene, bvh=bvh, cam=cam, slots=slots,
                    contacts=torch.empty(slots * 32, dtype=torch.uint8, device=dev),
                    rgba=torch.empty((slots, 4), dtype=torch.float32, device=dev))
        if tiled:
            item['slots0'] = capi.tiles_owned_pixels(cam, 0, tiling)      # rank 0 owns the most tiles: gather slot size
            if rank == 0:
                item['frame'] = torch.empty((npx, 4), dtype=torch.float32, device=dev)
        work.append(item)

    if tiled:
        # The frames of a step travel in ONE gather (fewer, larger collectives: a gather costs ~50 us of host time and the
        # links into rank 0 are the bound).  12-byte wire format: three float planes per frame and rank,
        # hit flag in the sign bit of red (rt_shade_blinn_phong_tiled_packed).  The send buffer is double buffered: step k
        # shades into buffer k & 1 while the gather of step k - 1 may still be reading the other one.
        slots0 = work[0]['slots0']
        assert all(it['slots0'] == slots0 for it in work)
        send = [torch.zeros((len(SCENES), 3, slots0), dtype=torch.float32, device=dev) for _ in range(2)]
        gathered = torch.empty((world, len(SCENES), 3, slots0), dtype=torch.float32, device=dev) if rank == 0 else None
        gctx = capi.Context(local_rank)                    # gather + untile stream, ordered behind the five frame streams
        gstream = torch.cuda.ExternalStream(gctx.get_stream(), device=dev)
        gather_done = [None, None]                         # event behind the gather that last read send[b]
        gather_events = [torch.cuda.Event(), torch.cuda.Event()]
        step_no = [0]
        # per-step Python work is the bound for N >= 4: everything that can be prepared once is
        gather_out = list(gathered.unbind(0)) if rank == 0 else None
        frame_ptrs = [it['frame'].data_ptr() for it in work] if rank == 0 else None
        gathered_ptrs = [gathered[0, i].data_ptr() for i in range(len(SCENES))] if rank == 0 else None
        rank_stride = len(SCENES) * 3 * slots0
        native_gather = dist.get_backend() != 'gloo'      # gloo (rehearsal on one GPU) stages through the host in frame_tiles

    # One HIP graph per frame (trace with the calibrated strategy, then shade): replaying it is one launch call instead of
    # five.  On one GPU the device is the bound and this changes nothing; with the frame tiled over N GPUs the host is
    # (DESIGN.md section 6).  Creating a graph traces one frame eagerly first, which is also where the BVH gets calibrated.
    use_graphs = not args.no_graphs and os.environ.get('RTMI_BENCH_GRAPHS', '1') != '0'
    if use_graphs:
        try:
            for i, it in enumerate(work):
                if not tiled:
                    it['graph'] = [capi.FrameGraph(it['bvh'], it['scene'], it['cam'], it['contacts'].data_ptr(), it['rgba'].data_ptr())]
                else:
                    it['graph'] = [capi.FrameGraph(it['bvh'], it['scene'], it['cam'], it['contacts'].data_ptr(), send[b][i].data_ptr(),
                                                   capi.FRAME_TILED_PACKED, rank, tiling, slots0) for b in range(2)]
        except capi.RtError as e:
            print('frame graphs unavailable (%s): issuing frames eagerly' % e, file=sys.stderr)
            use_graphs = False
    graphs_on = [use_graphs]                      # the measurement passes after the timed region issue eagerly

    def step():
        if not tiled:
            for it in work:
                if graphs_on[0]:
                    it['graph'][0].launch()
                else:
                    it['bvh'].generate_contacts_primary_d(it['cam'], it['contacts'].data_ptr())
                    capi.shade_blinn_phong_frame_d(it['ctx'], it['scene'], it['cam'], it['contacts'].data_ptr(), it['rgba'].data_ptr())
            return
        b = step_no[0] & 1
        step_no[0] += 1
        for i, it in enumerate(work):
            if gather_done[b] is not None:
                it['tstream'].wait_event(gather_done[b])            # send[b] is free again
            if graphs_on[0]:
                it['graph'][b].launch()
            else:
                it['bvh'].generate_contacts_primary_tiled_d(it['cam'], rank, tiling, it['contacts'].data_ptr())
                # shade straight into this frame's planes of the send buffer (ranks own slots <= slots0)
                capi.shade_blinn_phong_tiled_packed_d(it['ctx'], it['scene'], it['cam'], rank, tiling, it['contacts'].data_ptr(),
                                                      send[b][i].data_ptr(), slots0)
            gstream.wait_stream(it['tstream'])
        with torch.cuda.stream(gstream):
            # the one collective of a step: issued behind all five frames, runs on the communicator's stream
            if native_gather:
                w = dist.gather(send[b], gather_out, dst=0, async_op=True)
            else:
                _, w = frame_tiles.gather_to_root(send[b], world, rank, out=gathered, async_op=True)
            w.wait()                                                 # orders the gather stream behind the gather
            gather_events[b].record(gstream)
            gather_done[b] = gather_events[b]
        if rank == 0:
            for i, it in enumerate(work):
                capi.untile_rgba_packed_d(gctx, it['cam'], tiling, gathered_ptrs[i], slots0, frame_ptrs[i], rank_stride_floats=rank_stride)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    def check_frames(when):
        # rank 0 traces and shades every frame untiled and compares it with the one the ranks assembled
        if rank == 0:
            # on a context of its own: the untiled frame needs larger deferred-ray lists than the tiles, and growing them on a
            # frame's context would invalidate that frame's graphs (rt_frame_graph_launch then reports them stale)
            cctx = capi.Context(local_rank)
            for it in work:
                cscene = capi.DeviceScene(cctx, it['bytes'])
                cbvh = capi.DeviceBVH(cctx, cscene).build()
                full_c = torch.empty(npx * 32, dtype=torch.uint8, device=dev)
                full_rgba = torch.empty((npx, 4), dtype=torch.float32, device=dev)
                cbvh.generate_contacts_primary_d(it['cam'], full_c.data_ptr())
                capi.shade_blinn_phong_d(cctx, cscene, it['cam'], full_c.data_ptr(), npx, full_rgba.data_ptr())
                cctx.synchronize()
                torch.cuda.synchronize(dev)
                same = torch.equal(full_rgba, it['frame'])
                cbvh.close(); cscene.close()
                if not same:
                    raise SystemExit('frame check failed for %s (%s): the gathered frame differs from the untiled one' % (it['name'], when))
            cctx.close()
            print('frame check ok (%s): %d gathered frames equal the untiled ones' % (when, len(work)), file=sys.stderr)
        fence()

    if args.check_frames and tiled:
        if args.warmup == 0:
            step(); fence()
        check_frames('after the warm-up')
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    issued = time.perf_counter() - t0            # host time to issue the steps (the floor when the device part shrinks)
    fence()
    elapsed = time.perf_counter() - t0
    if args.check_frames and tiled:
        check_frames('after the timed steps')     # the frames of the last, fully overlapped step
    # HIP-event span of a traversal launch while the five streams overlap: a short eager pass (a graph has no events in it)
    graphs_on[0] = False
    for c in ctxs:
        c.set_profiling(True)
    for _ in range(min(args.steps, 10)):
        step()
    fence()
    kt = dict(launches=0, lane_phase_ms=0.0, wave_phase_ms=0.0)
    for c in ctxs:
        k1 = c.trace_kernel_times()
        for f in kt:
            kt[f] += k1[f]
        c.set_profiling(False)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    rays_per_step = npx * len(SCENES)
    value = rays_per_step * args.steps / elapsed / 1e6

    # ---- per-scene device times (untimed extra pass, HIP events) and the roofline of the traversal kernels
    # (one scene at a time: what a frame costs when it has the device to itself)
    per_scene = {}
    for it in work:
        sctx = it['ctx']
        tm = capi.Timer(sctx)
        for _ in range(ISOLATED_FRAMES):       # untimed: the device idled during the fence above and the clocks need a millisecond of work
            if not tiled:
                it['bvh'].generate_contacts_primary_d(it['cam'], it['contacts'].data_ptr())
            else:
                it['bvh'].generate_contacts_primary_tiled_d(it['cam'], rank, tiling, it['contacts'].data_ptr())
        sctx.set_profiling(2)                  # events + the <1, 2> kernel names: launches that had the device to themselves
        tm.start()
        for _ in range(ISOLATED_FRAMES):
            if not tiled:
                it['bvh'].generate_contacts_primary_d(it['cam'], it['contacts'].data_ptr())
            else:
                it['bvh'].generate_contacts_primary_tiled_d(it['cam'], rank, tiling, it['contacts'].data_ptr())
        tm.stop()
        ms = tm.elapsed_ms() / ISOLATED_FRAMES
        k = sctx.trace_kernel_times()
        sctx.set_profiling(False)
        tm.close()
        fx = frames_fixture[it['name']][key]
        alg = 56 * fx['inner_visits'] + 40 * fx['leaf_tests'] + 32 * fx['rays']
        # bytes of the records actually walked: primary rays use the search tree (same leaves, fewer inner nodes) when the
        # BVH has one; counts from tests/golden/search_tree_frames.json (make_golden_search.py)
        st = it['bvh'].search_tree()
        walked = alg
        if st['depth'] and st['used_by_last_camera']:
            sx = search_fixture[it['name']][key]
            walked = 56 * sx['inner_visits'] + 40 * sx['leaf_tests'] + 32 * sx['rays']
        per_scene[it['name']] = dict(trace_ms=round(ms, 4), first_kernel_ms=round(k['lane_phase_ms'] / ISOLATED_FRAMES, 4),
                                     wave_per_ray_ms=round(k['wave_phase_ms'] / ISOLATED_FRAMES, 4),
                                     mrays_per_s=round(it['slots'] / ms / 1e3, 1),
                                     inner_visits_per_ray=round(fx['inner_visits'] / fx['rays'], 2),
                                     leaf_tests_per_ray=round(fx['leaf_tests'] / fx['rays'], 2),
                                     algorithmic_bytes=alg, walked_bytes=walked, search_tree_depth=st['depth'],
                                     reference_tree_depth=it['bvh'].info()['max_depth'], build_ms=round(build_ms[it['name']], 3),
                                     strategy=it['bvh'].trace_strategy())

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    launches = max(kt['launches'], 1)
    trace_ms_concurrent = (kt['lane_phase_ms'] + kt['wave_phase_ms']) / launches   # timed region: launches overlap on five streams
    alg_total = sum(per_scene[n]['algorithmic_bytes'] for n in SCENES)
    share = (work[0]['slots'] / float(sum(capi.tiles_owned_pixels(work[0]['cam'], r, tiling) for r in range(world)))) if tiled else 1.0
    alg_per_launch = alg_total / len(SCENES) * share                                # this rank's share of an average frame
    walked_per_launch = sum(per_scene[n]['walked_bytes'] for n in SCENES) / len(SCENES) * share
    # The five frames of a step run side by side on five streams, so in the timed region a launch's HIP-event span contains
    # the time it shared the device with four others (and queued behind them): it is not a kernel duration any more and
    # rocprofv3's per-kernel durations of those launches are stretched the same way.  The roofline figure of the kernel is
    # therefore taken from the isolated pass above -- the same launches on the same frames in the same process, one scene at a
    # time, ISOLATED_FRAMES each, timed with HIP events on their stream; rocprofv3 lists exactly these launches as <1, 2>.
    # What the device as a whole moved during the timed region is reported beside it (achieved_all_streams).
    trace_ms_avg = sum(per_scene[n]['first_kernel_ms'] + per_scene[n]['wave_per_ray_ms'] for n in SCENES) / len(SCENES)
    dev_s = elapsed                                                                 # max over ranks, all streams
    achieved = walked_per_launch / (trace_ms_avg * 1e-3) / 1e9
    achieved_ref = alg_per_launch / (trace_ms_avg * 1e-3) / 1e9                     # the reference algorithm's bytes
    achieved_device = walked_per_launch * launches / dev_s / 1e9                    # all concurrent launches together
    traffic = None
    tpath = os.path.join(ROOT, 'profiles', 'traffic_latest.json')
    if os.path.exists(tpath) and world == 1:
        traffic = json.load(open(tpath)).get('hbm_bytes_per_launch')
    roofline = dict(bound='hbm', achieved=round(achieved, 1), peak=HBM_PEAK_GBPS, unit='GB/s',
                    frac=round(achieved / HBM_PEAK_GBPS, 4), traffic=traffic,
                    kernel='traversal launch of one frame: k_trace_packet4, k_trace_packet or k_trace, then k_trace_long (+ k_resolve_split) '
                           'where work was handed over.  Template tags in profiles: <1, 2> = the isolated launches this figure is '
                           'measured on, <1, 0> = the overlapping launches of the timed region, <1, 1> = first-frame calibration',
                    avg_launch_ms=round(trace_ms_avg, 4), isolated_launches=ISOLATED_FRAMES * len(SCENES),
                    concurrent_streams=len(SCENES),
                    timed_region_launch_span_ms=round(trace_ms_concurrent, 4),
                    achieved_all_streams=round(achieved_device, 1),
                    measured_hbm_gbps=(round(traffic / (trace_ms_avg * 1e-3) / 1e9, 1) if traffic else None),
                    measured_hbm_frac=(round(traffic / (trace_ms_avg * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if traffic else None),
                    binds='instruction issue (VALU / SALU) on cache-resident records, not HBM: frac > 1 and measured_hbm_frac say so',
                    algorithmic_bytes_per_launch=int(walked_per_launch),
                    reference_algorithm_bytes_per_launch=int(alg_per_launch),
                    achieved_reference_algorithm=round(achieved_ref, 1),
                    note='achieved = bytes of one traversal launch / its HIP-event duration, averaged over the five scenes, measured '
                         'live in this run on launches that have the device to themselves (avg_launch_ms; rocprofv3 --stats shows the '
                         'same launches as <1, 2>).  In the timed region the five frames of a step overlap on five streams: a launch\'s '
                         'event span there (timed_region_launch_span_ms) includes the time it shared the device, and '
                         'achieved_all_streams = bytes of all traversal launches of the timed region / its duration is the rate of the '
                         'device as a whole (the walked records are L2-resident for the small scenes, hence above the HBM peak).  '
                         'Bytes: 56 B per inner record + 40 B per triangle + 32 B contact of the hierarchy actually '
                         'walked (search tree over the reference leaves); *_reference_algorithm is the same time against the bytes '
                         'the reference traversal order would touch (SURVEY.md 8(d)); traffic = measured HBM bytes per launch (rocprofv3 FETCH_SIZE / '
                         'WRITE_SIZE passes, profiles/traffic_latest.json), measured_hbm_* = that traffic over the same launch duration')

    cpu_baseline = None
    if world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, 'oracle'))
        import oracle as orc                     # cpu_baseline leg only: the oracle is the thing timed here
        cores = orc.ncpu()
        t_cpu, rays_cpu = 0.0, 0
        c_buf = np.zeros(npx, dtype=orc.CONTACT)              # output buffers live outside the timed region, pages touched
        rgba_buf = np.zeros((npx, 4), dtype=np.float32)
        for name in SCENES:
            osc = orc.Scene(load_scene_bytes(name))
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
            osc = orc.Scene(load_scene_bytes(name))
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

    out = dict(metric='Mrays/s (primary rays, BVH trace + shade, bundled TestModels, 1920x1080)', value=round(value, 1),
               unit='Mrays/s', n_gpus=world, steps=args.steps, warmup=args.warmup,
               ms_per_step=round(elapsed / args.steps * 1e3, 4), higher_is_better=True, scaling='strong',
               vs_baseline=None, dtype='f32', data='bundled TestModels OBJ scenes (shipped with the reference), demo default camera',
               config=dict(workload='5 bundled TestModels scenes x 1920x1080 primary rays, BVH traversal + Blinn-Phong shading',
                           rays_per_step=rays_per_step, resolution=list(RES), scenes=SCENES,
                           parallelism=('single GPU, the five frames of a step on five HIP streams' if world == 1 else
                                        ('64x64 frame tiles dealt over %d ranks (rank 0 owns %d extra per period of %d tiles), one gather of RGBA '
                                         '(12 B per pixel on the wire) to rank 0 per step, the five frames of a step on five HIP streams per rank'
                                         % (world, root_extra, world + root_extra)))),
               ms_per_frame=round(elapsed / args.steps / len(SCENES) * 1e3, 4),
               host_issue_ms_per_step=round(issued / args.steps * 1e3, 4),
               roofline=roofline, cpu_baseline=cpu_baseline, per_scene=per_scene)
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
