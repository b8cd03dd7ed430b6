import sys, time, os
sys.path.insert(0,'.'); sys.path.insert(0,'oracle')
import numpy as np
from openclraytracer_amd import capi
import scene_pack as sp, oracle as orc
ctx=capi.Context(0)
print(ctx.device_info())
for name in ['cube','gardenlight','tenryuu','man','clumsydragon']:
    buf=sp.load_scene('assets/scenes/%s.txt'%name)
    osc=orc.Scene(buf)
    sc=capi.DeviceScene(ctx,buf)
    bvh=capi.DeviceBVH(ctx,sc)
    t=time.time(); bvh.build(); tb=time.time()-t
    t=time.time(); bvh.build(); tb2=time.time()-t
    cam=osc.demo_camera(1920,1080).view(capi.CAMERA)
    npx=1920*1080
    dc=ctx.alloc(npx*32); dr=ctx.alloc(npx*16)
    tm=capi.Timer(ctx)
    for i in range(3): bvh.generate_contacts_primary_d(cam,dc)
    ctx.synchronize()
    K=20
    tm.start()
    for i in range(K): bvh.generate_contacts_primary_d(cam,dc)
    tm.stop(); ms=tm.elapsed_ms()/K
    tm.start()
    for i in range(K): capi.shade_blinn_phong_d(ctx,sc,cam,dc,npx,dr)
    tm.stop(); ms2=tm.elapsed_ms()/K
    print('%-13s tris %6d depth %3d build %.1f/%.1f ms  trace %.3f ms = %.0f Mrays/s   shade %.3f ms'%(name,sc.n_tris,bvh.info()['max_depth'],tb*1e3,tb2*1e3,ms,npx/ms/1e3,ms2),flush=True)
