// Algorithms/AccelerationStructureManager.h -- the drop-in boundary: the six virtuals of the reference
// (OpenCLRayTracer/Include/Algorithms/AccelerationStructureManager.h:52-78), unchanged in name and meaning.
#ifndef RTMI_HOST_ASM_H
#define RTMI_HOST_ASM_H
#include <Common/Errata.h>
#include <OpenCLUtils/CLBuffer.h>

struct Camera;

namespace CLRayTracer {
class Scene;
namespace AccelerationStructures {

class AccelerationStructureManager {
public:
    AccelerationStructureManager(const OpenCLUtils::CLExecutionContext& context, const Scene& scene)
        : _context(context), _scene(scene) {}
    virtual Common::Result initialize(Common::Errata& err) = 0;
    virtual Common::Result initializeFrame(Common::Errata& err) = 0;
    virtual Common::Result construct(Common::Errata& err) = 0;
    virtual Common::Result generateContacts(Camera& cam, Common::Errata& err) = 0;
    virtual Common::Result generateContacts(OpenCLUtils::CLBuffer& rays, OpenCLUtils::CLBuffer& contacts,
                                            const unsigned int rayCount, Common::Errata& err) = 0;
    virtual const SharedPtr<OpenCLUtils::CLBuffer> getPrimaryContacts() const = 0;
    virtual ~AccelerationStructureManager() {}
protected:
    const OpenCLUtils::CLExecutionContext& _context;
    const Scene& _scene;
};

}}  // namespace
#endif
