// Shading.cpp -- cpuShadeBlinnPhong of the host mirror (include/CLData/Shading.h; reference: Include/CLData/Shading.h:74-166).
// shade_one (csrc/shade_math.h) is the library's one statement of that function: the device kernels compile it with hipcc, this
// file with g++ -ffp-contract=off (same operations in the same order; sqrtf and the divides are IEEE on both sides, powf is each
// side's library function).
#include <cstring>

#include <CLData/Shading.h>

#include "../../shade_math.h"

static_assert(sizeof(Camera) == sizeof(rt_camera) && sizeof(Contact) == sizeof(rt_contact), "the reference PODs and the C-ABI PODs are the same bytes");

void cpuShadeBlinnPhong(Camera* cam, Contact* contact, char* sceneBuffer, CL_FLOAT4& resultingColor)
{
    // SceneHeader, SceneBufferParser.h:55-66: the counts the light / material walkers need (LIGHTS_PTR / MATERIALS_PTR, :69-74)
    uint64_t nLights, nSpheres, nMaterials;
    std::memcpy(&nLights, sceneBuffer + 16, 8);
    std::memcpy(&nSpheres, sceneBuffer + 24, 8);
    std::memcpy(&nMaterials, sceneBuffer + 32, 8);
    ShadeScene sc;
    sc.scene = reinterpret_cast<const unsigned char*>(sceneBuffer);
    sc.n_lights = (uint32_t)nLights; sc.n_spheres = (uint32_t)nSpheres; sc.n_materials = (uint32_t)nMaterials;
    rt_camera c; rt_contact k;
    std::memcpy(&c, cam, sizeof c);
    std::memcpy(&k, contact, sizeof k);
    const float4 col = shade_one(sc, c, k);
    resultingColor.x = col.x; resultingColor.y = col.y; resultingColor.z = col.z; resultingColor.w = col.w;
}
