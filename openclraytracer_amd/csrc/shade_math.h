// shade_math.h -- cpuShadeBlinnPhong per contact (Include/CLData/Shading.h:74-166, lightEnergyPercentage Primitives/Light.h:55-58),
// shared by k_shade and by the traversal kernels that shade a contact as they store it (trace_kernels.hip, fused frames).
// Dual target like device_math.h: the host mirror's cpuShadeBlinnPhong (csrc/host/include/CLData/Shading.h) compiles shade_one with g++.
#pragma once
#if defined(__HIP__) || defined(__HIPCC__)
#include "rtmi_internal.h"
#else
#include "../../include/rtmi.h"
#endif
#include "device_math.h"

struct ShadeScene {
    const unsigned char* __restrict__ scene;   // raw packed scene buffer: lights + materials
    uint32_t n_lights, n_spheres, n_materials;
};

RT_FN float light_energy_percentage(float distance, float lightEnergy)
{
    return RMAX((1.0f - distance / lightEnergy), 0.0f);
}

RT_FN float4 shade_one(const ShadeScene& a, const rt_camera& cam, const rt_contact& c)
{
    // default material, Shading.h:76-96
    f3 ambient = { 0.2f, 0.2f, 0.2f }, diffuse = { 1.0f, 1.0f, 1.0f }, specular = { 1.0f, 0.0f, 0.0f };
    float shininess = 0.4f;
    const float4* lights = reinterpret_cast<const float4*>(a.scene + 96);                       // LIGHTS_PTR
    const float4* materials = reinterpret_cast<const float4*>(a.scene + 96 + 16ull * a.n_lights + 16ull * a.n_spheres);
    f3 o, d;
    generate_ray(cam, c.pixelIndex, o, d);
    float t = c.normalAndDistance[3];
    f3 P = f3_add(o, f3_scale(d, t));
    if (c.materialIndex < a.n_materials) {
        const float4* m = materials + 6ull * c.materialIndex;     // Material = 96 B
        ambient = f3_from4(m[0]); diffuse = f3_from4(m[1]); specular = f3_from4(m[2]);
        shininess = m[5].x;
    }
    f3 camPos = { cam.viewTransform[3], cam.viewTransform[7], cam.viewTransform[11] };
    f3 N = { c.normalAndDistance[0], c.normalAndDistance[1], c.normalAndDistance[2] };
    f3 V = f3_normalize(f3_sub(camPos, P));
    float cr = ambient.x, cg = ambient.y, cb = ambient.z;
    float totalEnergy = 0.0f;
    uint32_t nl = a.n_lights + 1;          // scene lights, then the camera light (energy 100000)
    for (uint32_t i = 0; i < nl; i++) {
        float4 l = (i < a.n_lights) ? lights[i] : make_float4(camPos.x, camPos.y, camPos.z, 100000.0f);
        float dist = f3_length(f3_sub(f3_from4(l), P));
        totalEnergy += light_energy_percentage(dist, l.w) * l.w;
    }
    for (uint32_t i = 0; i < nl; i++) {
        float4 l = (i < a.n_lights) ? lights[i] : make_float4(camPos.x, camPos.y, camPos.z, 100000.0f);
        f3 L = f3_sub(f3_from4(l), P);
        float dist = f3_length(L);
        float lightIntensity = light_energy_percentage(dist, l.w);
        float percentageOfTotal = (lightIntensity * l.w) / totalEnergy;
        float lsEnergy = lightIntensity * percentageOfTotal;
        L = f3_normalize(L);
        float LDotN = f3_dot(L, N);
        LDotN = LDotN > 0.0f ? LDotN : 0.0f;
        cr += diffuse.x * LDotN * lsEnergy;
        cg += diffuse.y * LDotN * lsEnergy;
        cb += diffuse.z * LDotN * lsEnergy;
        f3 R = f3_normalize(f3_sub(f3_scale(N, 2.0f * LDotN), L));
        float nSpecular = powf(f3_dot(R, V), shininess);
        float px = specular.x * nSpecular * lsEnergy;
        float py = specular.y * nSpecular * lsEnergy;
        float pz = specular.z * nSpecular * lsEnergy;
        cr += RMAX(px, 0.0f);
        cg += RMAX(py, 0.0f);
        cb += RMAX(pz, 0.0f);
    }
    return make_float4(RMIN(cr, 1.0f), RMIN(cg, 1.0f), RMIN(cb, 1.0f), 1.0f);
}

