// OpenCLUtils/CLBuffer.h -- device buffer with grow-only resize (OpenCLRayTracer/Include/OpenCLUtils/CLBuffer.h:39-112,
// src/OpenCLUtils/CLBuffer.cpp:40-121), backed by rt_buf_* of librtmi.so.
#ifndef RTMI_HOST_CLBUFFER_H
#define RTMI_HOST_CLBUFFER_H
#include <OpenCLUtils/CLExecutionContext.h>

namespace CLRayTracer { namespace OpenCLUtils {

class CLBuffer {
public:
    CLBuffer(const CLExecutionContext& context, size_t size, CLBufferFlags::CLBufferAccess access);
    CLBuffer(const CLExecutionContext& context, size_t size, void* source, CLBufferFlags::CLBufferAccess access);
    virtual ~CLBuffer();
    void resize(size_t newSize);                                 // expands, never shrinks; throws on failure
    size_t getSize() const { return _size; }
    size_t getActualSize() const { return _actualSize; }
    CLBufferFlags::CLBufferAccess getAccess() const { return _access; }
    const cl_mem& getCLMem() const { return _actualBuffer; }
    Common::Result copyToHost(void* target, Common::Errata& err) const;
    Common::Result copyToHost(void* target, size_t offset, Common::Errata& err) const;
private:
    const CLExecutionContext& _context;
    size_t _size, _actualSize;
    CLBufferFlags::CLBufferAccess _access;
    cl_mem _actualBuffer;
};

}}  // namespace
#endif
