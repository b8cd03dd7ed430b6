"""Wave timeline of k_trace_packet4 (librtmi_p4v3.so, make p4stamps): how many waves are resident over the launch, and where the
slowest waves sit.  RTMI_LIBRARY=librtmi_p4v3.so RTMI_PACKET=4 RTMI_DEFER_BUDGET=1048576 python tools/p4_timeline.py man"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
path = os.environ.setdefault('RTMI_P4_WAVES', '/tmp/p4_waves.bin')
import bench
from openclraytracer_amd import capi
out = {}
for n in sys.argv[1:] or ['man']:
    buf = bench.load_atrium_bytes() if n == 'atrium' else bench.load_scene_bytes(n)
    res = (3840, 2160) if n == 'atrium' else (1920, 1080)
    ctx = capi.Context(0)
    sc = capi.DeviceScene(ctx, buf); bvh = capi.DeviceBVH(ctx, sc).build()
    cam = bench.demo_camera(buf, *res)
    dc = ctx.alloc(res[0] * res[1] * 32)
    for _ in range(4):
        bvh.generate_contacts_primary_d(cam, dc)
    ctx.synchronize()
    raw = np.fromfile(path, dtype=np.uint64)
    if raw[0] == 0xFFFFFFFF:
        # the sharing kernel: P4_DBG_JOBS x (start, end, steps << 1 | is_task) per wave, block waves first, then the helpers
        n_waves, per = int(raw[1]), int(raw[2])
        j = raw[3:].reshape(n_waves, per, 3)
        used = j[:, :, 1] != 0
        t0 = j[:, :, 0][used].astype(np.int64); t1 = j[:, :, 1][used].astype(np.int64)
        steps = (j[:, :, 2][used] >> np.uint64(1)).astype(np.int64); is_task = (j[:, :, 2][used] & np.uint64(1)).astype(bool)
        base = t0.min(); t0 -= base; t1 -= base
        total = int(t1.max())
        edges = np.linspace(0, total, 21)
        dur = (t1 - t0)
        def busy_in(e0, e1, sel):
            a = np.clip(t1[sel], e0, e1) - np.clip(t0[sel], e0, e1)
            return float(a.sum()) / max(e1 - e0, 1)
        out[n] = dict(launch_us=total / 100.0, jobs=int(used.sum()), tasks=int(is_task.sum()),
                      block_jobs_busy_waves=[round(busy_in(a, b, ~is_task)) for a, b in zip(edges[:-1], edges[1:])],
                      task_jobs_busy_waves=[round(busy_in(a, b, is_task)) for a, b in zip(edges[:-1], edges[1:])],
                      task_us_mean=float(dur[is_task].mean()) / 100.0 if is_task.any() else 0.0, task_steps_mean=float(steps[is_task].mean()) if is_task.any() else 0.0,
                      task_us_per_step=float(dur[is_task].sum()) / 100.0 / max(int(steps[is_task].sum()), 1),
                      block_us_per_step=float(dur[~is_task].sum()) / 100.0 / max(int(steps[~is_task].sum()), 1),
                      steps_blocks=int(steps[~is_task].sum()), steps_tasks=int(steps[is_task].sum()),
                      first_task_start_us=float(t0[is_task].min()) / 100.0 if is_task.any() else None,
                      last_block_end_us=float(t1[~is_task].max()) / 100.0,
                      waves_overflowing_record=int((used.sum(axis=1) == per).sum()))
        for name, sel in (('slowest_block_jobs', ~is_task), ('slowest_tasks', is_task), ('last_ending_jobs', np.ones_like(is_task))):
            idx = np.nonzero(sel)[0]
            key = (t1 - t0) if name != 'last_ending_jobs' else t1
            top = idx[np.argsort(-key[idx])[:6]]
            out[n][name] = [dict(start_us=float(t0[i]) / 100.0, us=float(t1[i] - t0[i]) / 100.0, steps=int(steps[i]), task=bool(is_task[i])) for i in top]
        hist = np.histogram(steps[~is_task], bins=[0, 16, 32, 64, 128, 192, 256, 384, 512, 100000])[0]
        out[n]['block_job_steps_hist_0_16_32_64_128_192_256_384_512'] = [int(x) for x in hist]
        ctx.free(dc); bvh.close(); sc.close()
        continue
    w = raw.reshape(-1, 3)
    nw = (res[0] // 64 + (1 if res[0] % 64 else 0)) * (res[1] // 64 + (1 if res[1] % 64 else 0)) * 64
    w = w[:nw]
    t0 = w[:, 0].astype(np.int64); t1 = w[:, 1].astype(np.int64); steps = w[:, 2].astype(np.int64)
    base = t0.min(); t0 -= base; t1 -= base
    total = int(t1.max())
    edges = np.linspace(0, total, 21)
    busy = float((t1 - t0).sum())
    resident = [round(float((np.clip(t1, a, b) - np.clip(t0, a, b)).sum()) / max(b - a, 1)) for a, b in zip(edges[:-1], edges[1:])]
    order = np.argsort(-(t1 - t0))[:5]
    out[n] = dict(waves=int(nw), launch_us=total / 100.0, mean_resident=busy / total, resident_waves_per_5pct=resident,
                  last_start_us=float(t0.max()) / 100.0, mean_wave_us=float((t1 - t0).mean()) / 100.0,
                  slowest=[dict(wave=int(i), start_us=float(t0[i]) / 100.0, us=float(t1[i] - t0[i]) / 100.0, steps=int(steps[i])) for i in order],
                  us_per_step_slowest=float((t1[order[0]] - t0[order[0]])) / 100.0 / max(int(steps[order[0]]), 1),
                  steps_total=int(steps.sum()), us_per_step_mean=busy / 100.0 / max(int(steps.sum()), 1))
    ctx.free(dc); bvh.close(); sc.close()
print(json.dumps(out, indent=1))
