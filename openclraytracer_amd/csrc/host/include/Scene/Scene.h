// Scene/Scene.h -- scene text file + Wavefront OBJ/MTL -> packed scene buffer -> device
// (OpenCLRayTracer/Include/Scene/Scene.h:66-106, src/CLRayTracerEngine/Scene.cpp:231-456).
// File grammar as in the reference (Scene.h:29-49): one object per line, "LIGHT x y z e", "SPHERE x y z r",
// "MESH path.obj"; mesh paths are taken relative to the working directory, then to the scene file.
#ifndef RTMI_HOST_SCENE_H
#define RTMI_HOST_SCENE_H
#include <OpenCLUtils/CLBuffer.h>
#include <OpenCLUtils/CLExecutionContext.h>

struct rt_scene;

namespace CLRayTracer {

class Scene {
public:
    Scene();
    virtual ~Scene();
    Common::Result load(const char* filename, Common::Errata& err);
    Common::Result loadToGPU(const OpenCLUtils::CLExecutionContext& context, Common::Errata& err);
    const cl_mem& getDeviceSceneData() const { return _deviceSceneData; }
    const char* getHostSceneData() const { return _hostSceneData; }
    cl_ulong getSceneDataSize() const { return _sceneDataSize; }
    rt_scene* deviceHandle() const { return _deviceScene; }      // used by the managers of this library
    /** extension: 32 = pack and upload 32-bit vertex indices (the reference's ushort indices, SceneBufferParser.h:160, wrap for a
     *  mesh with more than 65 535 unique vertices); 16 = the reference format (default).  Call before load().           */
    void setIndexWidth(int bits) { _indexBytes = bits == 32 ? 4 : 2; }
    int getIndexWidth() const { return _indexBytes * 8; }
private:
    char* _hostSceneData;
    cl_ulong _sceneDataSize;
    cl_mem _deviceSceneData;
    rt_scene* _deviceScene;
    int _indexBytes = 2;
};

}  // namespace
#endif
