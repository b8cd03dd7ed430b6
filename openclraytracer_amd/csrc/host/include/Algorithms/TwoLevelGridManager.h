// Algorithms/TwoLevelGridManager.h -- host interface of the Two-Level Grid
// (OpenCLRayTracer/Include/Algorithms/TwoLevelGridManager.h:61-178, src/CLRayTracerEngine/TwoLevelGridManager.cpp),
// implemented on rt_tlg_* of librtmi.so.
#ifndef RTMI_HOST_TWOLEVELGRIDMANAGER_H
#define RTMI_HOST_TWOLEVELGRIDMANAGER_H
#include <Algorithms/AccelerationStructureManager.h>
#include <CLData/CLStructs.h>

struct rt_tlg;

namespace CLRayTracer { namespace AccelerationStructures {

class TwoLevelGridManager : public AccelerationStructureManager {
public:
    TwoLevelGridManager(const OpenCLUtils::CLExecutionContext& context, const Scene& scene);
    ~TwoLevelGridManager() override;
    Common::Result initialize(Common::Errata& err) override;
    Common::Result initializeFrame(Common::Errata& err) override;
    Common::Result construct(Common::Errata& err) override;
    Common::Result generateContacts(Camera& cam, Common::Errata& err) override;
    Common::Result generateContacts(OpenCLUtils::CLBuffer& rays, OpenCLUtils::CLBuffer& contacts,
                                    const unsigned int rayCount, Common::Errata& err) override;
    const SharedPtr<OpenCLUtils::CLBuffer> getPrimaryContacts() const override { return _primaryContactsArray; }

    /** properties, TwoLevelGridManager.h:112-118: take effect at the next initializeFrame */
    inline void setTopLevelDensity(CL_FLOAT value) { _topLevelDensity = value; }
    inline void setLeafDensity(CL_FLOAT value) { _leafDensity = value; }
    inline CL_FLOAT getTopLevelDensity() const { return _topLevelDensity; }
    inline CL_FLOAT getLeafDensity() const { return _leafDensity; }
    /** grid resolution of the last initializeFrame (TwoLevelGridManager.h:127) */
    CL_UINT3 getResolution() const;
    /** bounding box of the associated scene = box of the grid (TwoLevelGridManager.h:129-134, .cpp:605-608) */
    struct AABB getBounds() const;
    /** sizes of the last build: (cell, triangle) pairs and (leaf cell, triangle) pairs incl. holes, leaf cells */
    unsigned int pairsCount() const { return _pairsCount; }
    unsigned int leafPairsCount() const { return _leafPairsCount; }
    unsigned int leafCellsCount() const { return _leafCellsCount; }
protected:
    CL_FLOAT _topLevelDensity, _leafDensity;
    unsigned int _pairsCount, _leafPairsCount, _leafCellsCount;
    rt_tlg* _tlg;
    SharedPtr<OpenCLUtils::CLBuffer> _primaryContactsArray;
};

}}  // namespace
#endif
