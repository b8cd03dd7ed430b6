"""Exploration: what the reference's own OpenCL kernels (oracle/_ref/cl_*, see oracle/ref_cl.py) produce on the GPU
box next to the oracle.  Prints, asserts nothing; tests/test_gpu_reference_cl.py holds what this established."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'oracle')]
import oracle as orc, scene_pack, ref_cl

def rows(a): return a.view(np.uint8).reshape(len(a), -1)
def ndiff(a, b): return int((rows(a) != rows(b)).any(axis=1).sum())

def cmp_contacts(tag, got, want):
    gm, wm = got['normal_t'][:, 3] > 0, want['normal_t'][:, 3] > 0
    both = gm & wm
    t_g, t_w = got['normal_t'][both, 3], want['normal_t'][both, 3]
    rel = np.abs(t_g - t_w) / np.maximum(np.abs(t_w), 1e-30)
    nrm = np.abs(got['normal_t'][both, :3] - want['normal_t'][both, :3]).max() if both.any() else 0.0
    print('   %-22s rays %d hits ref %d oracle %d mask-diff %d material-diff %d  t-bit-exact %d/%d  t max rel %.3g  normal max abs %.3g  pixelIndex ok %s  all-bytes-diff %d'
          % (tag, len(got), gm.sum(), wm.sum(), (gm != wm).sum(), (got['materialIndex'][both] != want['materialIndex'][both]).sum(),
             (t_g.view(np.uint32) == t_w.view(np.uint32)).sum(), both.sum(), rel.max() if both.any() else 0.0, nrm,
             bool((got['pixelIndex'] == np.arange(len(got))).all()), ndiff(got, want)), flush=True)

def main():
  models = sys.argv[1:] or ['cube', 'gardenlight', 'tenryuu', 'man', 'clumsydragon']
  os.chdir(ROOT)
  for variant in ('strict', 'default'):
      ref = ref_cl.RefCL(variant)
      for name in models:
          buf = scene_pack.load_scene(os.path.join('assets', 'scenes', name + '.txt'))
          sc = orc.Scene(buf)
          F, Lz = orc.build_stages(sc, 0), orc.build_stages(sc, 1)
          n, N = F['n'], F['N']
          print('== %s / %s: n=%d N=%d' % (variant, name, n, N), flush=True)
          nodes, pairs = ref.calculate_morton_codes(buf, n)
          print('   morton: nodes differing %d (data %d lo %d hi %d), pairs differing %d' % (
              ndiff(nodes, F['leaves']), (nodes['data'] != F['leaves']['data']).any(axis=1).sum(),
              (nodes['lo'].view(np.uint32) != F['leaves']['lo'].view(np.uint32)).any(axis=1).sum(),
              (nodes['hi'].view(np.uint32) != F['leaves']['hi'].view(np.uint32)).any(axis=1).sum(), ndiff(pairs, F['pairs'])), flush=True)
          t0 = time.time()
          srt = ref.bitonic_sort(F['pairs'])
          print('   sort: differing %d of %d (%d launches, %.2fs)' % (ndiff(srt, F['sorted']), N, ref.sort_launches, time.time() - t0), flush=True)
          rad = ref.build_radix_tree(F['leaves'], F['sorted'], n)
          dl = np.nonzero((rows(rad) != rows(Lz['radix'])).any(axis=1))[0]
          df = np.nonzero((rows(rad) != rows(F['radix'])).any(axis=1))[0]
          print('   radix: vs oracle LITERAL differing %d %s ; vs oracle FIXED differing %d %s (lost node %d)' % (len(dl), dl[:6], len(df), df[:6], 2 * n - 2), flush=True)
          box = ref.compute_bounding_boxes(F['radix'], n)
          db = np.nonzero((rows(box) != rows(F['nodes'])).any(axis=1))[0]
          print('   refit (on the FIXED tree): differing %d %s' % (len(db), db[:6]), flush=True)
          if len(db):
              k = db[0]; print('      ', box[k], F['nodes'][k])
          for res in ((64, 48), (320, 240)):
              cam = sc.demo_camera(*res)
              ob = orc.BVH(sc, 0)
              want, st = ob.trace_primary(cam)
              if st.max_stack >= 30:
                  print('   skip %s: max stack %d' % (res, st.max_stack)); continue
              got = ref.generate_contacts(cam, F['nodes'], n, buf)
              cmp_contacts('primary %dx%d' % res, got, want)
              rays = orc.generate_rays(cam)
              want2, _ = ob.trace_rays(rays)
              got2 = ref.generate_contacts2(rays, F['nodes'], n, buf)
              cmp_contacts('rays %dx%d' % res, got2, want2)
              obl = orc.BVH(sc, 1)
              wantl, stl = obl.trace_primary(cam)
              if stl.max_stack < 30:
                  gotl = ref.generate_contacts2(rays, Lz['nodes'], n, buf)
                  wl2, _ = obl.trace_rays(rays)
                  cmp_contacts('LITERAL rays %dx%d' % res, gotl, wl2)
      ref.close()


if __name__ == '__main__':
    main()
