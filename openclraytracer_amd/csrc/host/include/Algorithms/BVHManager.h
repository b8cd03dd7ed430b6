// Algorithms/BVHManager.h -- host interface of the BVH (OpenCLRayTracer/Include/Algorithms/BVHManager.h:59-126,
// src/CLRayTracerEngine/BVHManager.cpp), implemented on rt_bvh_* of librtmi.so.
#ifndef RTMI_HOST_BVHMANAGER_H
#define RTMI_HOST_BVHMANAGER_H
#include <Algorithms/AccelerationStructureManager.h>
#include <CLData/CLStructs.h>

struct rt_bvh;

namespace CLRayTracer { namespace AccelerationStructures {

class BVHManager : public AccelerationStructureManager {
public:
    /** rt_bvh_create's mode word, value for value (include/rtmi.h: RT_MODE_*) */
    enum BuildMode { BuildFixed = 0, BuildLiteralZero = 1, BuildFast = 2, BuildDeviceArithmetic = 0x100 };
    BVHManager(const OpenCLUtils::CLExecutionContext& context, const Scene& scene);
    ~BVHManager() override;
    Common::Result initialize(Common::Errata& err) override;
    Common::Result initializeFrame(Common::Errata& err) override;
    Common::Result construct(Common::Errata& err) override;
    Common::Result generateContacts(Camera& cam, Common::Errata& err) override;
    Common::Result generateContacts(OpenCLUtils::CLBuffer& rays, OpenCLUtils::CLBuffer& contacts,
                                    const unsigned int rayCount, Common::Errata& err) override;
    const SharedPtr<OpenCLUtils::CLBuffer> getPrimaryContacts() const override { return _primaryContactsArray; }
    /** parity read-back of the hierarchy in the reference's 48-byte BVHNode layout (2n-1 nodes) */
    Common::Result readNodes(void* hostNodes, Common::Errata& err) const;
    /** rt_bvh_create's mode word (include/rtmi.h: RT_MODE_FIXED, the default / RT_MODE_LITERAL_ZERO / RT_MODE_FAST, optionally
     *  | RT_MODE_DEVICE_ARITH: dot() / cross() as the reference's DEVICE branch computes them on this GPU); before initialize() */
    void setBuildMode(int mode) { _buildMode = mode; }
    int buildMode() const { return _buildMode; }
    unsigned int leavesCount() const { return _bvhLeavesCount; }
    rt_bvh* deviceHandle() const { return _bvh; }                 // used by MultiGPUFrame of this library
protected:
    unsigned int _bvhLeavesCount;
    rt_bvh* _bvh;
    int _buildMode = 0;
    SharedPtr<OpenCLUtils::CLBuffer> _primaryContactsArray;
};

/** the host shading loop of RTFrame (OpenCLRayTracerDemo.cpp:298-316) on the device: colors = float4 per pixel */
Common::Result shadeBlinnPhong(const OpenCLUtils::CLExecutionContext& context, const Scene& scene, Camera& cam,
                               const OpenCLUtils::CLBuffer& contacts, unsigned int contactCount,
                               OpenCLUtils::CLBuffer& colors, Common::Errata& err);

/** rays of the next bounce from the contacts of this one (extension; the reference plans it, KNOWN_ISSUES_AND_TODO.txt 2.2):
 *  shadow rays towards `target` (kind 0) or reflections (kind 1), for every contact with a hit, in contact order.  Incoming
 *  rays: the camera's (cam != NULL) or rays[k] for contact k.  outRays must hold contactCount rays; outCount = rays written. */
Common::Result generateSecondaryRays(const OpenCLUtils::CLExecutionContext& context, Camera* cam, const OpenCLUtils::CLBuffer* rays,
                                     const OpenCLUtils::CLBuffer& contacts, unsigned int contactCount, int kind,
                                     const CL_FLOAT3& target, float eps, OpenCLUtils::CLBuffer& outRays, unsigned int& outCount,
                                     Common::Errata& err);

}}  // namespace
#endif
