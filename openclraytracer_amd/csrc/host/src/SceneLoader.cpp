// SceneLoader.cpp -- scene text file + Wavefront OBJ/MTL -> the reference's packed scene buffer.  Pure C++, no GPU.
//
// Restates Scene::load (OpenCLRayTracer/src/CLRayTracerEngine/Scene.cpp:231-444: key map, size computation,
// processMaterials :129-167, mesh packing :380-441) and the behaviour of the OBJ/MTL parser the reference vendors
// (Include/3rdParty/tiny_obj_loader.h v0.9.17: float grammar :266-383, index triples :421-452, face groups flushed
// on usemtl / g / o with a per-flush vertex cache :516-585, fan triangulation, LoadMtl :587-813) with own code.
// Third-party notice: fix_index and parse_triple below follow tinyobj's fixIndex / parseTriple (tiny_obj_loader.h v0.9.17 :408-449,
// Copyright (c) 2012-2015 Syoyo Fujita, MIT licence) statement for statement -- twenty lines whose behaviour (atoi, which separators end a
// field, what "1//3" and "-1" mean) IS the specification of the index grammar; everything else in this file is own code.
// The byte layout is SceneBufferParser.h:55-160.  Bytes the reference leaves uninitialised (.w of CL_FLOAT3
// temporaries) are 0.  tests/ compare the output with the oracle's packer and with the committed fixtures.
#include "SceneLoader.h"

#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <tuple>
#include <vector>

namespace rtmi_host {
namespace {

enum { SCENE_HEADER = 96, LIGHT = 16, SPHERE = 16, MATERIAL = 96, MODEL_HEADER = 64, MESH_HEADER = 48, VERTEX = 16 };

struct ObjMaterial { float v[19]; std::string name; };    // ambient diffuse specular transmittance emission | Ns Ni d illum
struct ObjShape { std::vector<float> positions; std::vector<uint32_t> indices; std::vector<int> material_ids; };

// A number of the reference's OBJ parser (tiny_obj_loader.h v0.9.17 :266-383), scanned in three parts: integer digits
// (at least one is required, so ".5" is not a number), an optional fraction, an optional exponent (which needs at least one
// digit, else the whole token is rejected).  Anything else ends the number silently.  It is NOT a correctly rounded strtod,
// and byte-identical packing needs its roundings: integer digits accumulate as value * 10 + d, fraction digit k adds
// d * 10^-k, and the decimal exponent E is applied as value * 5^E * 2^E.
struct NumberCursor {
    const char* at;
    const char* end;
    bool more() const { return at != end; }
    bool on_digit() const { return more() && *at >= '0' && *at <= '9'; }
    bool on_sign() const { return more() && (*at == '+' || *at == '-'); }
    int take_digit() { return *at++ - '0'; }
};

bool scan_obj_number(const char* begin, const char* end, double* value_out)
{
    NumberCursor c{ begin, end };
    if (!c.more()) return false;
    bool negative = false;
    if (c.on_sign()) negative = *c.at++ == '-';
    else if (!c.on_digit()) return false;
    if (!c.on_digit()) return false;                       // a sign alone, or a sign followed by '.'
    double value = 0.0;
    while (c.on_digit()) { value *= 10; value += c.take_digit(); }
    if (c.more() && *c.at == '.') {
        c.at++;
        for (int place = 1; c.on_digit(); place++) value += c.take_digit() * std::pow(10.0, -place);
    }
    int exp10 = 0;
    if (c.more() && (*c.at == 'e' || *c.at == 'E')) {
        c.at++;
        bool exp_negative = false;
        if (c.on_sign()) exp_negative = *c.at++ == '-';
        else if (!c.on_digit()) return false;
        if (!c.on_digit()) return false;                   // "1e", "1e+": rejected as a whole
        while (c.on_digit()) { exp10 *= 10; exp10 += c.take_digit(); }
        if (exp_negative) exp10 = -exp10;
    }
    *value_out = (negative ? -1 : 1) * std::ldexp(value * std::pow(5.0, exp10), exp10);
    return true;
}

float parse_float(const char*& token)
{
    token += std::strspn(token, " \t");
    const char* end = token + std::strcspn(token, " \t\r");
    double val = 0.0;
    scan_obj_number(token, end, &val);
    token = end;
    return (float)val;
}

int fix_index(int idx, int n) { return idx > 0 ? idx - 1 : (idx == 0 ? 0 : n + idx); }

struct VertexIndex { int v, vt, vn; bool operator<(const VertexIndex& o) const { return std::tie(v, vn, vt) < std::tie(o.v, o.vn, o.vt); } };

VertexIndex parse_triple(const char*& token, int vsize, int vnsize, int vtsize)
{
    VertexIndex vi{ -1, -1, -1 };
    vi.v = fix_index(std::atoi(token), vsize);
    token += std::strcspn(token, "/ \t\r");
    if (token[0] != '/') return vi;
    token++;
    if (token[0] == '/') { token++; vi.vn = fix_index(std::atoi(token), vnsize); token += std::strcspn(token, "/ \t\r"); return vi; }
    vi.vt = fix_index(std::atoi(token), vtsize);
    token += std::strcspn(token, "/ \t\r");
    if (token[0] != '/') return vi;
    token++;
    vi.vn = fix_index(std::atoi(token), vnsize);
    token += std::strcspn(token, "/ \t\r");
    return vi;
}

std::string first_word(const char* s)   // sscanf(s, "%s", buf)
{
    s += std::strspn(s, " \t\n\v\f\r");
    size_t n = std::strcspn(s, " \t\n\v\f\r");
    return std::string(s, n);
}

bool is_space(char c) { return c == ' ' || c == '\t'; }

void trim_line(std::string& line)
{
    if (!line.empty() && line.back() == '\n') line.pop_back();
    if (!line.empty() && line.back() == '\r') line.pop_back();
}

void init_material(ObjMaterial& m)
{
    for (float& f : m.v) f = 0.f;
    m.v[15] = 1.f; m.v[16] = 1.f; m.v[17] = 1.f; m.v[18] = 0.f;     // shininess ior dissolve illum
    m.name.clear();
}

void load_mtl(const std::string& path, std::map<std::string, int>& material_map, std::vector<ObjMaterial>& materials)
{
    ObjMaterial material; init_material(material);
    std::ifstream in(path.c_str());
    std::string line;
    while (in && std::getline(in, line)) {
        trim_line(line);
        if (line.empty()) continue;
        const char* token = line.c_str();
        token += std::strspn(token, " \t");
        if (token[0] == '\0' || token[0] == '#') continue;
        if (0 == std::strncmp(token, "newmtl", 6) && is_space(token[6])) {
            if (!material.name.empty()) { material_map.insert(std::make_pair(material.name, (int)materials.size())); materials.push_back(material); }
            init_material(material);
            material.name = first_word(token + 7);
            continue;
        }
        int slot = -1;
        if (token[0] == 'K' && is_space(token[2])) {
            if (token[1] == 'a') slot = 0; else if (token[1] == 'd') slot = 3; else if (token[1] == 's') slot = 6;
            else if (token[1] == 't') slot = 9; else if (token[1] == 'e') slot = 12;
        }
        if (slot >= 0) { token += 2; for (int k = 0; k < 3; k++) material.v[slot + k] = parse_float(token); continue; }
        if (token[0] == 'N' && token[1] == 'i' && is_space(token[2])) { token += 2; material.v[16] = parse_float(token); continue; }
        if (token[0] == 'N' && token[1] == 's' && is_space(token[2])) { token += 2; material.v[15] = parse_float(token); continue; }
        if (0 == std::strncmp(token, "illum", 5) && is_space(token[5])) { token += 6; token += std::strspn(token, " \t"); material.v[18] = (float)std::atoi(token); continue; }
        if (token[0] == 'd' && is_space(token[1])) { token += 1; material.v[17] = parse_float(token); continue; }
        if (token[0] == 'T' && token[1] == 'r' && is_space(token[2])) { token += 2; material.v[17] = 1.0f - parse_float(token); continue; }
    }
    material_map.insert(std::make_pair(material.name, (int)materials.size()));
    materials.push_back(material);
}

bool load_obj(const std::string& file, const std::string& mtl_base, std::vector<ObjShape>& shapes, std::vector<ObjMaterial>& materials,
              std::string& err)
{
    std::ifstream in(file.c_str());
    if (!in) { err = "Cannot open file [" + file + "]"; return false; }
    std::vector<float> v;
    int vn_count = 0, vt_count = 0;
    std::vector<std::vector<VertexIndex>> face_group;
    std::map<std::string, int> material_map;
    int material = -1;

    auto flush = [&]() {
        if (face_group.empty()) return;
        ObjShape shape;
        std::map<VertexIndex, uint32_t> cache;       // per flush: the reference passes the cache by value
        auto upd = [&](const VertexIndex& i) -> uint32_t {
            auto it = cache.find(i);
            if (it != cache.end()) return it->second;
            shape.positions.push_back(v[3 * (size_t)i.v + 0]);
            shape.positions.push_back(v[3 * (size_t)i.v + 1]);
            shape.positions.push_back(v[3 * (size_t)i.v + 2]);
            uint32_t idx = (uint32_t)(shape.positions.size() / 3 - 1);
            cache[i] = idx;
            return idx;
        };
        for (const auto& face : face_group) {
            VertexIndex i0 = face[0], i1{ -1, -1, -1 }, i2 = face[1];
            for (size_t k = 2; k < face.size(); k++) {
                i1 = i2; i2 = face[k];
                uint32_t a = upd(i0), b = upd(i1), c = upd(i2);
                shape.indices.push_back(a); shape.indices.push_back(b); shape.indices.push_back(c);
                shape.material_ids.push_back(material);
            }
        }
        shapes.push_back(shape);
    };

    std::string line;
    while (std::getline(in, line)) {
        trim_line(line);
        if (line.empty()) continue;
        const char* token = line.c_str();
        token += std::strspn(token, " \t");
        if (token[0] == '\0' || token[0] == '#') continue;
        if (token[0] == 'v' && is_space(token[1])) { token += 2; for (int k = 0; k < 3; k++) v.push_back(parse_float(token)); continue; }
        if (token[0] == 'v' && token[1] == 'n' && is_space(token[2])) { vn_count++; continue; }
        if (token[0] == 'v' && token[1] == 't' && is_space(token[2])) { vt_count++; continue; }
        if (token[0] == 'f' && is_space(token[1])) {
            token += 2; token += std::strspn(token, " \t");
            std::vector<VertexIndex> face;
            while (!(token[0] == '\r' || token[0] == '\n' || token[0] == '\0')) {
                VertexIndex vi = parse_triple(token, (int)(v.size() / 3), vn_count, vt_count);
                if (vi.v < 0 || (size_t)vi.v * 3 + 2 >= v.size()) { err = "face references a missing vertex in " + file; return false; }
                face.push_back(vi);
                token += std::strspn(token, " \t\r");
            }
            if (face.size() >= 2) face_group.push_back(face);
            continue;
        }
        if (0 == std::strncmp(token, "usemtl", 6) && is_space(token[6])) {
            std::string name = first_word(token + 7);
            flush(); face_group.clear();
            auto it = material_map.find(name);
            material = it != material_map.end() ? it->second : -1;
            continue;
        }
        if (0 == std::strncmp(token, "mtllib", 6) && is_space(token[6])) {
            std::string name = first_word(token + 7);
            load_mtl(mtl_base.empty() ? name : mtl_base + name, material_map, materials);
            continue;
        }
        if ((token[0] == 'g' || token[0] == 'o') && is_space(token[1])) { flush(); face_group.clear(); continue; }
    }
    flush();
    return true;
}

bool float_equals(float a, float b) { float diff = a - b; return diff <= FLT_EPSILON && diff >= -FLT_EPSILON; }   // RTKernelUtils.h:125-129
bool material_equals(const ObjMaterial& a, const ObjMaterial& b) { for (int k = 0; k < 19; k++) if (!float_equals(a.v[k], b.v[k])) return false; return true; }

// parseNumber (Scene.cpp:169-182) + stringToFloats (:184-200)
bool parse_number(const std::string& val, float& out)
{
    float f = (float)std::atof(val.c_str());
    if (f == 0.0) for (char c : val) if (c != '0' && c != '.') return false;
    out = f;
    return true;
}
int string_to_floats(const std::string& str, float* result, int size)
{
    std::vector<std::string> toks;
    std::string cur;
    bool prev_sep = false;
    for (char c : str) {
        if (c == ' ') { if (!prev_sep) { toks.push_back(cur); cur.clear(); } prev_sep = true; }
        else { cur += c; prev_sep = false; }
    }
    toks.push_back(cur);
    int count = (int)toks.size() < size ? (int)toks.size() : size, index = 0;
    for (int i = 0; i < count; i++) { float val = 0.f; if (parse_number(toks[i], val)) result[index++] = val; }
    return index;
}

template <class T> void put(std::vector<char>& buf, size_t off, const T& v) { std::memcpy(&buf[off], &v, sizeof(T)); }

std::string mtl_base_path(const std::string& fileName)   // loadMesh, Scene.cpp:58-68
{
    size_t s = fileName.find_last_of('\\'), r = fileName.find_last_of('/');
    size_t k;
    if (!(s == std::string::npos || r == std::string::npos)) k = s > r ? s : r; else k = s < r ? s : r;
    if (k == std::string::npos) return std::string();
    return fileName.substr(0, k + 1);
}

bool file_exists(const std::string& p) { std::ifstream f(p.c_str()); return (bool)f; }

}  // namespace

bool packSceneFile(const char* filename, std::vector<char>& out, std::string& err, int indexBytes)
{
    // INDEX_TYPE: the reference packs ushort indices (SceneBufferParser.h:160, Scene.cpp:430), which wrap for a mesh with
    // more than 65 535 unique vertices; 4 = the 32-bit extension (rt_scene_upload_indexed / Scene::setIndexWidth)
    if (indexBytes != 2 && indexBytes != 4) { err = "index width must be 2 or 4 bytes"; return false; }
    const uint64_t INDEX = (uint64_t)indexBytes;
    std::map<std::string, std::vector<std::string>> raw;     // std::map => keys in sorted order, as in the reference
    {
        std::ifstream f(filename);
        if (!f.is_open()) { err = std::string("File: ") + filename + " couldn't be opened!"; return false; }
        std::string line;
        while (std::getline(f, line)) {
            if (!line.empty() && line.back() == '\r') line.pop_back();     // text-mode CRLF folding
            size_t sp = line.find_first_of(' ');
            if (sp == std::string::npos) continue;
            raw[line.substr(0, sp)].push_back(line.substr(sp + 1));
        }
    }
    std::string scene_dir = mtl_base_path(filename);
    struct Model { std::vector<ObjShape> shapes; std::vector<ObjMaterial> materials; uint64_t size; std::vector<uint32_t> mesh_material; };
    std::vector<Model> models;
    std::vector<ObjMaterial> unique;
    uint64_t size = SCENE_HEADER, total_prims = 0, n_lights = 0, n_spheres = 0, total_model_size = 0;
    for (auto& kv : raw) {
        if (kv.first == "LIGHT") { n_lights = kv.second.size(); size += n_lights * LIGHT; }
        else if (kv.first == "SPHERE") { n_spheres = kv.second.size(); size += n_spheres * SPHERE; }
        else if (kv.first == "MESH") {
            for (std::string mesh : kv.second) {
                for (char& c : mesh) if (c == '\\') c = '/';
                std::string path = mesh;
                if (!file_exists(path) && file_exists(scene_dir + mesh)) path = scene_dir + mesh;
                Model m;
                if (!load_obj(path, mtl_base_path(path), m.shapes, m.materials, err)) return false;
                m.size = MODEL_HEADER;
                for (const ObjShape& s : m.shapes) m.size += s.indices.size() * INDEX + (s.positions.size() / 3) * VERTEX + MESH_HEADER;
                total_model_size += m.size;
                // processMaterials: unique materials + index remap; ids without an entry (e.g. -1) map to 0
                std::map<uint32_t, uint32_t> index_transform;
                for (uint32_t i = 0; i < m.materials.size(); i++) {
                    uint32_t found = (uint32_t)unique.size();
                    for (uint32_t j = 0; j < unique.size(); j++) if (material_equals(m.materials[i], unique[j])) { found = j; break; }
                    if (found == unique.size()) unique.push_back(m.materials[i]);
                    index_transform[i] = found;
                }
                for (const ObjShape& s : m.shapes) {
                    uint32_t id = s.material_ids.empty() ? 0u : (uint32_t)s.material_ids[0];
                    m.mesh_material.push_back(index_transform[id]);
                }
                models.push_back(std::move(m));
            }
            size += total_model_size;
        } else continue;
        total_prims += kv.second.size();
        size += unique.size() * MATERIAL;     // inside the key loop, as written (Scene.cpp:305)
    }

    out.assign(size, 0);
    uint64_t hdr[8] = { size, total_prims, n_lights, n_spheres, unique.size(), total_model_size, models.size(), 0 };
    std::memcpy(&out[0], hdr, sizeof hdr);
    size_t lights_off = SCENE_HEADER, spheres_off = lights_off + n_lights * LIGHT, mats_off = spheres_off + n_spheres * SPHERE;
    size_t models_off = mats_off + unique.size() * MATERIAL;
    const char* keys[2] = { "LIGHT", "SPHERE" };
    size_t offs[2] = { lights_off, spheres_off };
    for (int t = 0; t < 2; t++) {
        auto it = raw.find(keys[t]);
        if (it == raw.end()) continue;
        for (size_t i = 0; i < it->second.size(); i++) {
            float num[4];
            if (string_to_floats(it->second[i], num, 4) != 4) continue;      // invalid line: entry stays zero
            std::memcpy(&out[offs[t] + i * 16], num, 16);
        }
    }
    for (size_t i = 0; i < unique.size(); i++) {
        size_t o = mats_off + i * MATERIAL;
        for (int k = 0; k < 5; k++) std::memcpy(&out[o + 16 * k], &unique[i].v[3 * k], 12);
        std::memcpy(&out[o + 80], &unique[i].v[15], 16);
    }
    float scene_lo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, scene_hi[3] = { FLT_MIN, FLT_MIN, FLT_MIN };
    uint64_t scene_tris = 0;
    size_t off = models_off;
    for (const Model& m : models) {
        size_t mbase = off;
        float lo[3] = { FLT_MAX, FLT_MAX, FLT_MAX }, hi[3] = { FLT_MIN, FLT_MIN, FLT_MIN };
        uint64_t model_tris = 0;
        size_t moff = mbase + MODEL_HEADER;
        for (size_t si = 0; si < m.shapes.size(); si++) {
            const ObjShape& s = m.shapes[si];
            uint64_t nv = s.positions.size() / 3, ni = s.indices.size();
            uint64_t msize = ni * INDEX + nv * VERTEX + MESH_HEADER;
            uint64_t mh[6] = { msize, ni / 3, nv, ni, m.mesh_material[si], 0 };
            std::memcpy(&out[moff], mh, sizeof mh);
            model_tris += ni / 3;
            for (uint64_t vi = 0; vi < nv; vi++) {
                const float* p = &s.positions[3 * vi];
                for (int k = 0; k < 3; k++) { lo[k] = lo[k] < p[k] ? lo[k] : p[k]; hi[k] = hi[k] > p[k] ? hi[k] : p[k]; }
                std::memcpy(&out[moff + MESH_HEADER + vi * VERTEX], p, 12);
            }
            size_t ioff = moff + MESH_HEADER + nv * VERTEX;
            for (uint64_t ii = 0; ii < ni; ii++) {
                if (INDEX == 2) { uint16_t v16 = (uint16_t)s.indices[ii]; std::memcpy(&out[ioff + ii * 2], &v16, 2); }   // ushort truncation, as the reference
                else { uint32_t v32 = (uint32_t)s.indices[ii]; std::memcpy(&out[ioff + ii * 4], &v32, 4); }
            }
            for (int k = 0; k < 3; k++) { scene_lo[k] = lo[k] < scene_lo[k] ? lo[k] : scene_lo[k]; scene_hi[k] = hi[k] > scene_hi[k] ? hi[k] : scene_hi[k]; }
            moff += msize;
        }
        uint64_t mh[4] = { m.size, m.shapes.size(), model_tris, 0 };
        std::memcpy(&out[mbase], mh, sizeof mh);
        std::memcpy(&out[mbase + 32], lo, 12);
        std::memcpy(&out[mbase + 48], hi, 12);
        scene_tris += model_tris;
        off = mbase + m.size;
    }
    std::memcpy(&out[56], &scene_tris, 8);
    std::memcpy(&out[64], scene_lo, 12);
    std::memcpy(&out[80], scene_hi, 12);
    return true;
}

}  // namespace rtmi_host
