// HostRuntime.cpp -- CLExecutionContext, CLBuffer, Scene, BVHManager: the reference's host classes
// (same names, argument meaning and error behaviour) implemented on the C ABI of librtmi.so (include/rtmi.h).
#include <cstring>
#include <new>

#include <Algorithms/BVHManager.h>
#include <Algorithms/TwoLevelGridManager.h>
#include <CLData/CLStructs.h>
#include <OpenCLUtils/CLBuffer.h>
#include <OpenCLUtils/CLExecutionContext.h>
#include <Scene/Scene.h>

#include "../../../../include/rtmi.h"
#include "SceneLoader.h"

using namespace CLRayTracer;
using namespace CLRayTracer::Common;
using namespace CLRayTracer::OpenCLUtils;
using namespace CLRayTracer::AccelerationStructures;

static_assert((int)BVHManager::BuildFixed == (int)RT_MODE_FIXED && (int)BVHManager::BuildLiteralZero == (int)RT_MODE_LITERAL_ZERO && (int)BVHManager::BuildFast == (int)RT_MODE_FAST &&
              (int)BVHManager::BuildDeviceArithmetic == (int)RT_MODE_DEVICE_ARITH, "BVHManager::BuildMode mirrors include/rtmi.h");
static_assert(sizeof(Camera) == sizeof(rt_camera) && sizeof(Contact) == sizeof(rt_contact) && sizeof(Ray) == sizeof(rt_ray),
              "the reference PODs and the C-ABI PODs are the same bytes");

// ------------------------------------------------------------------ CLExecutionContext
CLExecutionContext::CLExecutionContext(int deviceOrdinal) : _ctx(nullptr)
{
    if (rt_ctx_create(deviceOrdinal, &_ctx) != RT_SUCCESS)
        throw CLInterfaceException(std::string(rt_last_error(nullptr)), __FILE__, __func__, __LINE__);   // constructors throw, CLBuffer.cpp:45-46
}
CLExecutionContext::~CLExecutionContext() { rt_ctx_destroy(_ctx); }

Result CLExecutionContext::check(int rc, Errata& err, const char* file, const char* func, int line) const
{
    if (rc == RT_SUCCESS) return Success;
    err = Errata(std::string(rt_last_error(_ctx)), file, func, line);
    return Error;
}
Result CLExecutionContext::flushQueue(Errata&) const { return Success; }     // HIP submits eagerly
Result CLExecutionContext::finishQueue(Errata& err) const { return check(rt_ctx_synchronize(_ctx), err, __FILE__, __func__, __LINE__); }
Result CLExecutionContext::setFramesInFlight(unsigned n, Errata& err) const { return check(rt_ctx_set_frames_in_flight(_ctx, n), err, __FILE__, __func__, __LINE__); }

// ------------------------------------------------------------------ CLBuffer (CLBuffer.cpp:40-121)
CLBuffer::CLBuffer(const CLExecutionContext& context, size_t size, CLBufferFlags::CLBufferAccess access)
    : _context(context), _size(size), _actualSize(size), _access(access), _actualBuffer(nullptr)
{
    if (rt_buf_alloc(context.handle(), size, &_actualBuffer) != RT_SUCCESS)
        throw CLInterfaceException(std::string(rt_last_error(context.handle())), __FILE__, __func__, __LINE__);
}
CLBuffer::CLBuffer(const CLExecutionContext& context, size_t size, void* source, CLBufferFlags::CLBufferAccess access)
    : CLBuffer(context, size, access)
{
    if (rt_buf_upload(context.handle(), _actualBuffer, source, size) != RT_SUCCESS)
        throw CLInterfaceException(std::string(rt_last_error(context.handle())), __FILE__, __func__, __LINE__);
}
CLBuffer::~CLBuffer() { rt_buf_free(_context.handle(), _actualBuffer); }

void CLBuffer::resize(size_t newSize)
{
    // expands when the buffer is smaller than requested, leaves the memory intact when shrinking (CLBuffer.cpp:70-94)
    if (newSize <= _actualSize) { _size = newSize; return; }
    cl_mem fresh = nullptr;
    if (rt_buf_alloc(_context.handle(), newSize, &fresh) != RT_SUCCESS)
        throw CLInterfaceException(std::string(rt_last_error(_context.handle())), __FILE__, __func__, __LINE__);
    rt_buf_free(_context.handle(), _actualBuffer);
    _actualBuffer = fresh;
    _size = _actualSize = newSize;
}
Result CLBuffer::copyToHost(void* target, Errata& err) const { return copyToHost(target, 0, err); }
Result CLBuffer::copyToHost(void* target, size_t offset, Errata& err) const
{
    if (offset > _actualSize) { FILL_ERRATA(err, "offset beyond the buffer"); return Error; }
    RTMI_CHECK(_context, rt_buf_download(_context.handle(), target, (const char*)_actualBuffer + offset, _actualSize - offset), err);
    return Success;
}

// ------------------------------------------------------------------ Scene (Scene.cpp:206-456)
Scene::Scene() : _hostSceneData(nullptr), _sceneDataSize(0), _deviceSceneData(nullptr), _deviceScene(nullptr) {}
Scene::~Scene()
{
    if (_deviceScene) rt_scene_destroy(_deviceScene);
    delete[] _hostSceneData;
}
Result Scene::load(const char* filename, Errata& err)
{
    std::vector<char> buf;
    std::string msg;
    if (!rtmi_host::packSceneFile(filename, buf, msg, _indexBytes)) { FILL_ERRATA(err, msg); return Error; }
    delete[] _hostSceneData;
    _hostSceneData = new char[buf.size()];
    std::memcpy(_hostSceneData, buf.data(), buf.size());
    _sceneDataSize = buf.size();
    return Success;
}
Result Scene::loadToGPU(const CLExecutionContext& context, Errata& err)
{
    if (!_hostSceneData) { FILL_ERRATA(err, "Scene::load must precede loadToGPU"); return Error; }
    if (_deviceScene) { rt_scene_destroy(_deviceScene); _deviceScene = nullptr; }
    RTMI_CHECK(context, rt_scene_upload_indexed(context.handle(), _hostSceneData, _sceneDataSize, _indexBytes, &_deviceScene), err);
    _deviceSceneData = const_cast<void*>(rt_scene_device_ptr(_deviceScene));
    return Success;
}

// ------------------------------------------------------------------ BVHManager (BVHManager.cpp:57-329)
BVHManager::BVHManager(const CLExecutionContext& context, const Scene& scene)
    : AccelerationStructureManager(context, scene), _bvhLeavesCount(0), _bvh(nullptr)
{
    // default memory allocation: single ray per pixel at 512x512 (BVHManager.cpp:69)
    _primaryContactsArray.reset(new CLBuffer(context, 512 * 512 * sizeof(Contact), CLBufferFlags::ReadWrite));
}
BVHManager::~BVHManager() { if (_bvh) rt_bvh_destroy(_bvh); }

Result BVHManager::initialize(Errata& err)
{
    // the reference JIT-compiles its OpenCL kernels here; the HIP kernels are compiled offline
    if (_bvh) return Success;
    if (!_scene.deviceHandle()) { FILL_ERRATA(err, "Scene::loadToGPU must precede initialize"); return Error; }
    RTMI_CHECK(_context, rt_bvh_create(_context.handle(), _scene.deviceHandle(), _buildMode, &_bvh), err);
    return Success;
}
Result BVHManager::initializeFrame(Errata& err)
{
    if (!_bvh) { FILL_ERRATA(err, "initialize must precede initializeFrame"); return Error; }
    RTMI_CHECK(_context, rt_bvh_initialize_frame(_bvh), err);
    rt_bvh_info(_bvh, &_bvhLeavesCount, nullptr, nullptr);
    return Success;
}
Result BVHManager::construct(Errata& err)
{
    if (!_bvh) { FILL_ERRATA(err, "initialize must precede construct"); return Error; }
    RTMI_CHECK(_context, rt_bvh_construct(_bvh), err);
    return Success;
}
Result BVHManager::generateContacts(Camera& cam, Errata& err)
{
    if (!_bvh) { FILL_ERRATA(err, "initialize must precede generateContacts"); return Error; }
    size_t contactBufSize = (size_t)cam.resX * cam.resY * sizeof(Contact);
    try { _primaryContactsArray->resize(contactBufSize); }                    // BVHManager.cpp:240-244
    catch (const CLInterfaceException& e) { err = e; return Error; }
    RTMI_CHECK(_context, rt_bvh_generate_contacts_primary(_bvh, reinterpret_cast<const rt_camera*>(&cam),
                                                           _primaryContactsArray->getCLMem()), err);
    // flush + wait, as after every enqueue of the reference (BVHManager.cpp:270-278)
    RTMI_CHECK(_context, rt_ctx_synchronize(_context.handle()), err);
    return Success;
}
Result BVHManager::generateContacts(CLBuffer& rays, CLBuffer& contacts, const unsigned int rayCount, Errata& err)
{
    if (!_bvh) { FILL_ERRATA(err, "initialize must precede generateContacts"); return Error; }
    if ((size_t)rayCount * sizeof(Ray) > rays.getActualSize() || (size_t)rayCount * sizeof(Contact) > contacts.getActualSize()) {
        FILL_ERRATA(err, "rayCount exceeds the ray or contact buffer");
        return Error;
    }
    RTMI_CHECK(_context, rt_bvh_generate_contacts_rays(_bvh, rays.getCLMem(), rayCount, contacts.getCLMem()), err);
    RTMI_CHECK(_context, rt_ctx_synchronize(_context.handle()), err);
    return Success;
}
Result BVHManager::readNodes(void* hostNodes, Errata& err) const
{
    if (!_bvh) { FILL_ERRATA(err, "BVH not built"); return Error; }
    RTMI_CHECK(_context, rt_bvh_read_nodes(_bvh, hostNodes), err);
    return Success;
}

// ------------------------------------------------------------------ TwoLevelGridManager (TwoLevelGridManager.cpp:60-600)
TwoLevelGridManager::TwoLevelGridManager(const CLExecutionContext& context, const Scene& scene)
    : AccelerationStructureManager(context, scene), _topLevelDensity(2.0f), _leafDensity(2.0f),   // :66-67
      _pairsCount(0), _leafPairsCount(0), _leafCellsCount(0), _tlg(nullptr)
{
}
TwoLevelGridManager::~TwoLevelGridManager() { if (_tlg) rt_tlg_destroy(_tlg); }

Result TwoLevelGridManager::initialize(Errata& err)
{
    if (_tlg) return Success;
    if (!_scene.deviceHandle()) { FILL_ERRATA(err, "Scene::loadToGPU must precede initialize"); return Error; }
    RTMI_CHECK(_context, rt_tlg_create(_context.handle(), _scene.deviceHandle(), &_tlg), err);
    return Success;
}
Result TwoLevelGridManager::initializeFrame(Errata& err)
{
    if (!_tlg) { FILL_ERRATA(err, "initialize must precede initializeFrame"); return Error; }
    RTMI_CHECK(_context, rt_tlg_set_densities(_tlg, _topLevelDensity, _leafDensity), err);
    RTMI_CHECK(_context, rt_tlg_initialize_frame(_tlg), err);          // calculateGridData + buffers, :160-201
    return Success;
}
Result TwoLevelGridManager::construct(Errata& err)
{
    if (!_tlg) { FILL_ERRATA(err, "initialize must precede construct"); return Error; }
    RTMI_CHECK(_context, rt_tlg_construct(_tlg), err);
    rt_tlg_info info;
    RTMI_CHECK(_context, rt_tlg_info_get(_tlg, &info), err);
    _pairsCount = info.pairs_count; _leafPairsCount = info.leaf_pairs_count; _leafCellsCount = info.leaf_cells_count;
    return Success;
}
Result TwoLevelGridManager::generateContacts(Camera& cam, Errata& err)
{
    if (!_tlg) { FILL_ERRATA(err, "initialize must precede generateContacts"); return Error; }
    size_t contactBufSize = (size_t)cam.resX * cam.resY * sizeof(Contact);
    try {                                                                       // TwoLevelGridManager.cpp:478-482
        if (_primaryContactsArray) _primaryContactsArray->resize(contactBufSize);
        else _primaryContactsArray.reset(new CLBuffer(_context, contactBufSize, CLBufferFlags::ReadWrite));
    } catch (const CLInterfaceException& e) { err = e; return Error; }
    RTMI_CHECK(_context, rt_tlg_generate_contacts_primary(_tlg, reinterpret_cast<const rt_camera*>(&cam),
                                                           _primaryContactsArray->getCLMem()), err);
    RTMI_CHECK(_context, rt_ctx_synchronize(_context.handle()), err);           // flush + wait, :504-510
    return Success;
}
Result TwoLevelGridManager::generateContacts(CLBuffer& rays, CLBuffer& contacts, const unsigned int rayCount, Errata& err)
{
    if (!_tlg) { FILL_ERRATA(err, "initialize must precede generateContacts"); return Error; }
    if ((size_t)rayCount * sizeof(Ray) > rays.getActualSize() || (size_t)rayCount * sizeof(Contact) > contacts.getActualSize()) {
        FILL_ERRATA(err, "rayCount exceeds the ray or contact buffer");
        return Error;
    }
    RTMI_CHECK(_context, rt_tlg_generate_contacts_rays(_tlg, rays.getCLMem(), rayCount, contacts.getCLMem()), err);
    RTMI_CHECK(_context, rt_ctx_synchronize(_context.handle()), err);
    return Success;
}
struct AABB TwoLevelGridManager::getBounds() const
{
    struct AABB b;
    std::memcpy(&b, _scene.getHostSceneData() + 64, sizeof b);          // SceneHeader.modelsBoundingBox
    return b;
}
CL_UINT3 TwoLevelGridManager::getResolution() const
{
    CL_UINT3 r = { 0, 0, 0, 0 };
    rt_grid_data g;
    if (_tlg && rt_tlg_read(_tlg, RT_TLG_GRID_DATA, &g, sizeof g) == RT_SUCCESS) { r.x = g.resX; r.y = g.resY; r.z = g.resZ; }
    return r;
}

Result CLRayTracer::AccelerationStructures::shadeBlinnPhong(const CLExecutionContext& context, const Scene& scene, Camera& cam,
                                                            const CLBuffer& contacts, unsigned int contactCount, CLBuffer& colors,
                                                            Errata& err)
{
    if ((size_t)cam.resX * cam.resY * 16 > colors.getActualSize() || (size_t)contactCount * sizeof(Contact) > contacts.getActualSize()) {
        FILL_ERRATA(err, "colour or contact buffer too small");
        return Error;
    }
    RTMI_CHECK(context, rt_shade_blinn_phong(context.handle(), scene.deviceHandle(), reinterpret_cast<const rt_camera*>(&cam),
                                             contacts.getCLMem(), contactCount, colors.getCLMem()), err);
    return Success;
}

Result CLRayTracer::AccelerationStructures::generateSecondaryRays(const CLExecutionContext& context, Camera* cam, const CLBuffer* rays,
                                                                  const CLBuffer& contacts, unsigned int contactCount, int kind,
                                                                  const CL_FLOAT3& target, float eps, CLBuffer& outRays,
                                                                  unsigned int& outCount, Errata& err)
{
    outCount = 0;
    if ((size_t)contactCount * sizeof(Contact) > contacts.getActualSize() || (size_t)contactCount * sizeof(Ray) > outRays.getActualSize() ||
        (rays && (size_t)contactCount * sizeof(Ray) > rays->getActualSize())) {
        FILL_ERRATA(err, "contactCount exceeds a buffer");
        return Error;
    }
    const float tgt[3] = { target.x, target.y, target.z };
    uint32_t n = 0;
    RTMI_CHECK(context, rt_generate_secondary_rays(context.handle(), reinterpret_cast<const rt_camera*>(cam), rays ? rays->getCLMem() : nullptr,
                                                   contacts.getCLMem(), contactCount, kind, tgt, eps, outRays.getCLMem(), &n), err);
    outCount = n;
    return Success;
}


// ------------------------------------------------------------------ MultiGPUFrame (new)
#include <Algorithms/MultiGPUFrame.h>
namespace CLRayTracer { namespace AccelerationStructures {

MultiGPUFrame::MultiGPUFrame(const std::vector<const CLExecutionContext*>& contexts, const std::vector<const Scene*>& scenes,
                             const std::vector<BVHManager*>& managers)
    : _contexts(contexts), _scenes(scenes), _managers(managers), _camera() {}

MultiGPUFrame::~MultiGPUFrame()
{
    for (rt_multi_frame* f : _frames) rt_multi_frame_destroy(f);
    for (rt_comm* c : _comms) rt_comm_destroy(c);
}

Result MultiGPUFrame::initialize(Camera& cam, unsigned int rootExtra, Errata& err)
{
    const size_t n = _contexts.size();
    if (n == 0 || _scenes.size() != n || _managers.size() != n) { FILL_ERRATA(err, "one context, scene and manager per GPU"); return Error; }
    for (size_t r = 0; r < n; r++)
        if (!_contexts[r] || !_scenes[r] || !_managers[r] || !_scenes[r]->deviceHandle() || !_managers[r]->deviceHandle()) {
            FILL_ERRATA(err, "every rank needs its scene on the GPU and its BVH built"); return Error; }
    for (rt_multi_frame* f : _frames) rt_multi_frame_destroy(f);
    _frames.clear();
    if (_comms.empty()) {
        std::vector<rt_ctx*> ctxs;
        for (const CLExecutionContext* c : _contexts) ctxs.push_back(c->handle());
        _comms.assign(n, nullptr);
        RTMI_CHECK(*_contexts[0], rt_comm_create_group(ctxs.data(), (int)n, _comms.data()), err);
    }
    _camera = cam;
    _frames.assign(n, nullptr);
    for (size_t r = 0; r < n; r++)
        RTMI_CHECK(*_contexts[r], rt_multi_frame_create(_comms[r], _managers[r]->deviceHandle(), _scenes[r]->deviceHandle(),
                                                        reinterpret_cast<const rt_camera*>(&_camera), rootExtra, &_frames[r]), err);
    return Success;
}

Result MultiGPUFrame::setCamera(Camera& cam, Errata& err)
{
    if (_frames.empty()) { FILL_ERRATA(err, "initialize must precede setCamera"); return Error; }
    for (size_t r = 0; r < _frames.size(); r++)
        RTMI_CHECK(*_contexts[r], rt_multi_frame_set_camera(_frames[r], reinterpret_cast<const rt_camera*>(&cam)), err);
    _camera = cam;
    return Success;
}

Result MultiGPUFrame::render(Errata& err)
{
    if (_frames.empty()) { FILL_ERRATA(err, "initialize must precede render"); return Error; }
    if (rt_multi_frame_render(_frames.data(), (int)_frames.size()) != RT_SUCCESS) {
        for (const CLExecutionContext* c : _contexts) {
            const char* m = rt_last_error(c->handle());
            if (m && *m) { FILL_ERRATA(err, m); return Error; }
        }
        FILL_ERRATA(err, "rt_multi_frame_render failed");
        return Error;
    }
    return Success;
}

Result MultiGPUFrame::finish(Errata& err)
{
    for (size_t r = 0; r < _frames.size(); r++) RTMI_CHECK(*_contexts[r], rt_multi_frame_wait(_frames[r]), err);
    return Success;
}

Result MultiGPUFrame::copyFrameToHost(void* rgba, Errata& err)
{
    if (_frames.empty()) { FILL_ERRATA(err, "initialize must precede copyFrameToHost"); return Error; }
    if (finish(err) != Success) return Error;
    RTMI_CHECK(*_contexts[0], rt_buf_download(_contexts[0]->handle(), rgba, rt_multi_frame_pixels(_frames[0]),
                                              (uint64_t)_camera.resX * _camera.resY * 16u), err);
    return Success;
}

bool MultiGPUFrame::usesRCCL() const { return !_comms.empty() && rt_comm_uses_rccl(_comms[0]) != 0; }

}}  // namespace
