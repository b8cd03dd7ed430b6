// ref_tinyobj_dump.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Driver around the REFERENCE's own vendored OBJ parser
// (OpenCLRayTracer/Include/3rdParty/tiny_obj_loader.h, v0.9.17), compiled from
// where it lies under /root/reference (never copied) into oracle/_ref/ by
// oracle/Makefile.  It calls tinyobj::LoadObj exactly as the reference's
// loadMesh does (src/CLRayTracerEngine/Scene.cpp:55-75: mtl base path = the
// directory prefix of the OBJ path, trailing separator kept) and dumps
// the resulting shapes/materials to a flat binary file that
// oracle/scene_pack.py reads.  This is the one piece of the path that the
// real reference code pins: it validates the repo's own OBJ/MTL parsers.
//
// Dump format (little endian):
//   "TOBJ1\n" | u32 nshapes | u32 nmaterials
//   per material: 19 x f32  (ambient[3] diffuse[3] specular[3] transmittance[3]
//                            emission[3] shininess ior dissolve (float)illum)
//   per shape   : u32 npositions(floats) u32 nindices u32 nmaterial_ids
//                 f32 positions[] | u32 indices[] | i32 material_ids[]
#define TINYOBJLOADER_IMPLEMENTATION
#include <3rdParty/tiny_obj_loader.h>

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

static void w32(FILE* f, uint32_t v) { fwrite(&v, 4, 1, f); }

int main(int argc, char** argv)
{
    if (argc != 3) { fprintf(stderr, "usage: %s model.obj out.bin\n", argv[0]); return 2; }
    std::string fileName = argv[1];
    // Scene.cpp:58-68
    size_t slashIdx = fileName.find_last_of('\\');
    size_t rSlashIdx = fileName.find_last_of('/');
    if (!(slashIdx == std::string::npos || rSlashIdx == std::string::npos))
        slashIdx = slashIdx > rSlashIdx ? slashIdx : rSlashIdx;
    else
        slashIdx = slashIdx < rSlashIdx ? slashIdx : rSlashIdx;
    std::string mtlBase = fileName;
    slashIdx++;
    mtlBase.erase(mtlBase.begin() + slashIdx, mtlBase.end());

    std::vector<tinyobj::shape_t> shapes;
    std::vector<tinyobj::material_t> materials;
    std::string err;
    if (!tinyobj::LoadObj(shapes, materials, err, fileName.c_str(), mtlBase.c_str())) {
        fprintf(stderr, "LoadObj failed: %s\n", err.c_str());
        return 1;
    }
    FILE* f = fopen(argv[2], "wb");
    if (!f) { perror("fopen"); return 1; }
    fwrite("TOBJ1\n", 1, 6, f);
    w32(f, (uint32_t)shapes.size());
    w32(f, (uint32_t)materials.size());
    for (size_t i = 0; i < materials.size(); i++) {
        const tinyobj::material_t& m = materials[i];
        float v[19];
        for (int k = 0; k < 3; k++) {
            v[k] = m.ambient[k]; v[3 + k] = m.diffuse[k]; v[6 + k] = m.specular[k];
            v[9 + k] = m.transmittance[k]; v[12 + k] = m.emission[k];
        }
        v[15] = m.shininess; v[16] = m.ior; v[17] = m.dissolve; v[18] = (float)m.illum;
        fwrite(v, 4, 19, f);
    }
    for (size_t i = 0; i < shapes.size(); i++) {
        const tinyobj::mesh_t& me = shapes[i].mesh;
        w32(f, (uint32_t)me.positions.size());
        w32(f, (uint32_t)me.indices.size());
        w32(f, (uint32_t)me.material_ids.size());
        if (!me.positions.empty()) fwrite(&me.positions[0], 4, me.positions.size(), f);
        if (!me.indices.empty()) fwrite(&me.indices[0], 4, me.indices.size(), f);
        if (!me.material_ids.empty()) fwrite(&me.material_ids[0], 4, me.material_ids.size(), f);
    }
    fclose(f);
    return 0;
}
