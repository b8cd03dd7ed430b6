import sys, os
sys.path.insert(0,'.'); sys.path.insert(0,'oracle')
import numpy as np
from openclraytracer_amd import capi
import scene_pack as sp, oracle as orc
names=sys.argv[1:] or ['tenryuu']
ctx=capi.Context(0)
for name in names:
    buf=sp.load_scene('assets/scenes/%s.txt'%name)
    osc=orc.Scene(buf)
    sc=capi.DeviceScene(ctx,buf); bvh=capi.DeviceBVH(ctx,sc).build()
    cam=osc.demo_camera(1920,1080).view(capi.CAMERA)
    dc=ctx.alloc(1920*1080*32)
    for i in range(5): bvh.generate_contacts_primary_d(cam,dc)
    ctx.synchronize()
