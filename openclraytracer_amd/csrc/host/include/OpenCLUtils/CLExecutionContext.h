// OpenCLUtils/CLExecutionContext.h -- stands where the reference has its OpenCL execution context
// (OpenCLRayTracer/Include/OpenCLUtils/CLExecutionContext.h:297-456).  The managers only hold a const
// reference to it; here it owns an rt_ctx (HIP device + stream) of librtmi.so.  `cl_mem` becomes the opaque
// device pointer so that Scene::getDeviceSceneData() / CLBuffer::getCLMem() keep their signatures.
#ifndef RTMI_HOST_CLEXECUTIONCONTEXT_H
#define RTMI_HOST_CLEXECUTIONCONTEXT_H
#include <cstddef>
#include <cstdint>
#include <memory>
#include <Common/Errata.h>

struct rt_ctx;
typedef void* cl_mem;
typedef uint64_t cl_ulong;

namespace CLRayTracer {
template <class T> using SharedPtr = std::shared_ptr<T>;     // the reference uses boost::shared_ptr

namespace OpenCLUtils {

namespace CLBufferFlags { enum CLBufferAccess { ReadOnly, WriteOnly, ReadWrite }; }

class CLExecutionContext {
public:
    /** opens HIP device `deviceOrdinal`; throws CLInterfaceException when no gfx950 device is there */
    explicit CLExecutionContext(int deviceOrdinal = 0);
    ~CLExecutionContext();
    CLExecutionContext(const CLExecutionContext&) = delete;
    CLExecutionContext& operator=(const CLExecutionContext&) = delete;
    Common::Result flushQueue(Common::Errata& err) const;      // CLExecutionContext.h:436
    Common::Result finishQueue(Common::Errata& err) const;     // CLExecutionContext.h:445
    /** beyond the reference's interface: how many frames the application keeps in flight on this device (rt_ctx_set_frames_in_flight,
     *  include/rtmi.h): 1 = the Demo's RTFrame loop (latency policy, the default), > 1 = throughput policy.  Same contacts either way. */
    Common::Result setFramesInFlight(unsigned n, Common::Errata& err) const;
    rt_ctx* handle() const { return _ctx; }
    /** turns a failed C-ABI call into the reference's (Result, Errata) pair */
    Common::Result check(int rc, Common::Errata& err, const char* file, const char* func, int line) const;
private:
    rt_ctx* _ctx;
};
#define RTMI_CHECK(ctx, call, err) \
    if ((ctx).check((call), (err), __FILE__, __func__, __LINE__) != CLRayTracer::Common::Success) return CLRayTracer::Common::Error

}}  // namespace
#endif
