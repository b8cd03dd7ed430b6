// CLData/Shading.h -- the host shading function of the reference's frame loop
// (OpenCLRayTracer/Include/CLData/Shading.h:74-166; called per contact by RTFrame, src/Demos/OpenCLRayTracerDemo/OpenCLRayTracerDemo.cpp:298-316).
// Same name, same signature, global namespace as in the reference: an unmodified RTFrame (getPrimaryContacts()->copyToHost, then the host
// loop) links against librtmi_host.so.  The arithmetic is ONE statement for host and device: csrc/shade_math.h (shade_one) is compiled
// here by g++ -ffp-contract=off and by hipcc for k_shade / the fused frames; powf comes from the respective math library, hence the
// 1e-5 relative tolerance between this function and the device-shaded frame (tests/test_gpu_more.py::test_cpp_reference_frame_loop).
#ifndef RTMI_HOST_SHADING_H
#define RTMI_HOST_SHADING_H
#include <CLData/CLStructs.h>

void cpuShadeBlinnPhong(Camera* cam, Contact* contact, char* sceneBuffer, CL_FLOAT4& resultingColor);

#endif
