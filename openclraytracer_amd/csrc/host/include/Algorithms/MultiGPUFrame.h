// Algorithms/MultiGPUFrame.h -- the frame tiled over the GPUs of one node (new: the reference drives one device,
// OpenCLRayTracer/src/OpenCLUtils/CLExecutionContext.cpp:193-196).  One process, one execution context +
// one replica of the scene and of the built BVHManager per GPU; every GPU traces and shades its 64x64 tiles, one gather
// (RCCL point-to-point into GPU 0) assembles the frame.  Implemented on rt_comm_* / rt_multi_frame_* of librtmi.so.
#ifndef RTMI_HOST_MULTIGPUFRAME_H
#define RTMI_HOST_MULTIGPUFRAME_H
#include <vector>
#include <Algorithms/BVHManager.h>

struct rt_comm;
struct rt_multi_frame;

namespace CLRayTracer { namespace AccelerationStructures {

class MultiGPUFrame {
public:
    /** rank r = contexts[r] with its scene replica scenes[r] and its built manager managers[r]; all three outlive this object */
    MultiGPUFrame(const std::vector<const OpenCLUtils::CLExecutionContext*>& contexts, const std::vector<const Scene*>& scenes,
                  const std::vector<BVHManager*>& managers);
    ~MultiGPUFrame();
    MultiGPUFrame(const MultiGPUFrame&) = delete;
    MultiGPUFrame& operator=(const MultiGPUFrame&) = delete;
    /** forms the communicator group (RCCL when the contexts' devices differ) and prepares the frame for `cam`;
     *  rootExtra = extra tiles per period that GPU 0, which gathers the frame, traces itself */
    Common::Result initialize(Camera& cam, unsigned int rootExtra, Common::Errata& err);
    /** the camera of the following frames (the Demo updates its camera before every frame, OpenCLRayTracerDemo.cpp:275-276);
     *  same resolution as at initialize */
    Common::Result setCamera(Camera& cam, Common::Errata& err);
    /** one frame: asynchronous (trace + shade on every GPU, gather, untile on GPU 0) */
    Common::Result render(Common::Errata& err);
    /** waits for the frames issued so far */
    Common::Result finish(Common::Errata& err);
    /** the assembled frame, resX*resY float4, to host memory (waits) */
    Common::Result copyFrameToHost(void* rgba, Common::Errata& err);
    bool usesRCCL() const;
private:
    std::vector<const OpenCLUtils::CLExecutionContext*> _contexts;
    std::vector<const Scene*> _scenes;
    std::vector<BVHManager*> _managers;
    std::vector<rt_comm*> _comms;
    std::vector<rt_multi_frame*> _frames;
    Camera _camera;
};

}}  // namespace
#endif
