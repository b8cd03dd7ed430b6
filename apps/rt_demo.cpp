// rt_demo -- headless counterpart of the reference's GLUT demo (src/Demos/OpenCLRayTracerDemo/OpenCLRayTracerDemo.cpp):
// the same five switches (-winH -winW -accStruct -headersPath -scene, :66-73,513-552), the same main() sequence
// (:130-266: load scene, upload, default camera, manager initialize / initializeFrame / construct) and the same
// RTFrame() sequence (:271-326) with the host shading loop replaced by the device shading call, then a PPM / PFM (float) / raw
// dump instead of the GL point-sprite display.  Written against the reference's class names only.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include <Algorithms/BVHManager.h>
#include <Algorithms/MultiGPUFrame.h>
#include <Algorithms/TwoLevelGridManager.h>
#include <CLData/CLStructs.h>
#include <CLData/Shading.h>
#include <Scene/Scene.h>

using namespace CLRayTracer;
using namespace CLRayTracer::Common;
using namespace CLRayTracer::OpenCLUtils;
using namespace CLRayTracer::AccelerationStructures;

#define CHECKED_CALL(result) if ((result) != Success) { std::cout << err; return 1; }

int main(int argc, char* argv[])
{
    int window_width = 0, window_height = 0, frames = 1;
    std::string scenePath, accStruct, headersPath, outPath, rawContacts, devices, arith = "host", build = "fixed";
    int gpus = 1;
    bool referenceFrame = false;
    for (int i = 1; i + 1 < argc; i += 2) {
        std::string k = argv[i], v = argv[i + 1];
        if (k == "-winH") window_height = std::atoi(v.c_str());
        else if (k == "-winW") window_width = std::atoi(v.c_str());
        else if (k == "-accStruct") accStruct = v;
        else if (k == "-headersPath") headersPath = v;        // accepted and ignored: kernels are compiled offline
        else if (k == "-scene") scenePath = v;
        else if (k == "--frames") frames = std::atoi(v.c_str());
        else if (k == "--out") outPath = v;
        else if (k == "--contacts") rawContacts = v;
        else if (k == "--gpus") gpus = std::atoi(v.c_str());          // frame tiled over N GPUs, one gather (BVH only)
        else if (k == "--reference-frame") referenceFrame = std::atoi(v.c_str()) != 0;   // RTFrame as the reference wrote it: contacts to the host, host shading loop
        else if (k == "--arith") arith = v;                          // host (default) | device: dot() / cross() as the reference's device branch on this GPU (RT_MODE_DEVICE_ARITH)
        else if (k == "--build") build = v;                          // fixed (default) | literal | fast: rt_bvh_create's parity mode
        else if (k == "--devices") devices = v;                        // e.g. 0,0,0: which HIP device each rank uses (default rank r -> device r)
    }
    if (window_width <= 0 || window_height <= 0 || scenePath.empty() || accStruct.empty()) {
        std::cout << "usage: rt_demo -winH h -winW w -accStruct BVH|GRID -headersPath dir -scene file.txt [--frames n] [--out img.ppm|img.pfm] [--contacts raw.bin] [--reference-frame 1] [--arith host|device] [--build fixed|literal|fast] [--gpus n [--devices d0,d1,..]]" << std::endl;
        return 0;
    }
    if (accStruct != "BVH" && accStruct != "GRID") {              // OpenCLRayTracerDemo.cpp:221-224
        std::cout << "Unsupported acceleration structure: " << accStruct << " (BVH or GRID)" << std::endl; return 1; }
    Errata err;
    if (gpus > 1 || !devices.empty()) {
        // ---- the frame tiled over several GPUs (new): one context, scene replica and BVH per rank ----
        if (accStruct != "BVH") { std::cout << "--gpus supports -accStruct BVH" << std::endl; return 1; }
        std::vector<int> dev;
        for (size_t p = 0; p < devices.size();) { size_t q = devices.find(',', p); if (q == std::string::npos) q = devices.size(); dev.push_back(std::atoi(devices.substr(p, q - p).c_str())); p = q + 1; }
        if (dev.empty()) for (int r = 0; r < gpus; r++) dev.push_back(r);
        const size_t n = dev.size();
        std::vector<std::unique_ptr<CLExecutionContext>> ctxs(n);
        std::vector<std::unique_ptr<Scene>> scenes(n);
        std::vector<std::unique_ptr<BVHManager>> managers(n);
        std::vector<const CLExecutionContext*> pc; std::vector<const Scene*> ps; std::vector<BVHManager*> pm;
        for (size_t r = 0; r < n; r++) {
            try { ctxs[r].reset(new CLExecutionContext(dev[r])); } catch (const CLInterfaceException& e) { std::cout << e; return 1; }
            scenes[r].reset(new Scene());
            CHECKED_CALL(scenes[r]->load(scenePath.c_str(), err));
            CHECKED_CALL(scenes[r]->loadToGPU(*ctxs[r], err));
            managers[r].reset(new BVHManager(*ctxs[r], *scenes[r]));
            CHECKED_CALL(managers[r]->initialize(err));
            CHECKED_CALL(managers[r]->initializeFrame(err));
            CHECKED_CALL(managers[r]->construct(err));
            pc.push_back(ctxs[r].get()); ps.push_back(scenes[r].get()); pm.push_back(managers[r].get());
        }
        const float* bb = reinterpret_cast<const float*>(scenes[0]->getHostSceneData() + 64);
        Camera camera;
        camera.resX = window_width; camera.resY = window_height;
        camera.FOVDistance = FOVDistFromAngle(90, (float)window_width, (float)window_height);
        camera.supersampilngFactor = 1;
        CL_FLOAT3 half, cameraPosition;
        half.x = (bb[4] - bb[0]) * 0.5f; half.y = (bb[5] - bb[1]) * 0.5f; half.z = (bb[6] - bb[2]) * 0.5f;
        cameraPosition.x = bb[0] + half.x; cameraPosition.y = bb[1] + half.y; cameraPosition.z = bb[2] - half.z; cameraPosition.w = 0;
        camera.viewTransform = identityTransform();
        Quaternion cameraOrientation = zeroRotation();
        normalizeQuaternion(&cameraOrientation);
        setOrientationAndPos(&camera.viewTransform, cameraOrientation, cameraPosition);
        MultiGPUFrame frame(pc, ps, pm);
        CHECKED_CALL(frame.initialize(camera, n > 1 ? 1u : 0u, err));
        CHECKED_CALL(frame.render(err));                               // warm-up
        CHECKED_CALL(frame.finish(err));
        auto t0 = std::chrono::steady_clock::now();
        for (int f = 0; f < frames; f++) CHECKED_CALL(frame.render(err));
        CHECKED_CALL(frame.finish(err));
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / frames;
        std::vector<CL_FLOAT4> colors((size_t)window_width * window_height);
        CHECKED_CALL(frame.copyFrameToHost(colors.data(), err));
        size_t hits = 0;
        for (const CL_FLOAT4& c : colors) hits += c.w > 0;
        std::cout << "ranks " << n << (frame.usesRCCL() ? " (RCCL gather)" : " (device-copy gather: ranks share a device)") << "  frames " << frames << "  "
                  << ms << " ms/frame  lit pixels " << hits << std::endl;
        if (!outPath.empty()) {
            FILE* fo = std::fopen(outPath.c_str(), "wb");
            if (!fo) { std::perror("fopen"); return 1; }
            std::fwrite(colors.data(), sizeof(CL_FLOAT4), colors.size(), fo);      // raw float4 frame
            std::fclose(fo);
        }
        return 0;
    }
    std::unique_ptr<CLExecutionContext> context;
    try { context.reset(new CLExecutionContext(0)); }
    catch (const CLInterfaceException& e) { std::cout << e; return 1; }

    Scene scene;
    CHECKED_CALL(scene.load(scenePath.c_str(), err));
    const float* bb = reinterpret_cast<const float*>(scene.getHostSceneData() + 64);     // SceneHeader.modelsBoundingBox
    std::cout << "Scene loaded: " << scene.getSceneDataSize() << " bytes" << std::endl;
    CHECKED_CALL(scene.loadToGPU(*context, err));

    // camera: OpenCLRayTracerDemo.cpp:202-217
    Camera camera;
    camera.resX = window_width; camera.resY = window_height;
    camera.FOVDistance = FOVDistFromAngle(90, (float)window_width, (float)window_height);
    camera.supersampilngFactor = 1;
    CL_FLOAT3 half, cameraPosition;
    half.x = (bb[4] - bb[0]) * 0.5f; half.y = (bb[5] - bb[1]) * 0.5f; half.z = (bb[6] - bb[2]) * 0.5f;
    cameraPosition.x = bb[0] + half.x; cameraPosition.y = bb[1] + half.y; cameraPosition.z = bb[2] - half.z; cameraPosition.w = 0;
    camera.viewTransform = identityTransform();
    Quaternion cameraOrientation = zeroRotation();

    std::unique_ptr<AccelerationStructureManager> accelerationStruct;
    if (accStruct == "BVH") {
        BVHManager* m = new BVHManager(*context, scene);
        if ((arith != "host" && arith != "device") || (build != "fixed" && build != "literal" && build != "fast")) { std::cout << "unknown --arith / --build value" << std::endl; return 1; }
        m->setBuildMode((build == "literal" ? BVHManager::BuildLiteralZero : build == "fast" ? BVHManager::BuildFast : BVHManager::BuildFixed) | (arith == "device" ? BVHManager::BuildDeviceArithmetic : 0));
        accelerationStruct.reset(m);
    }
    else accelerationStruct.reset(new TwoLevelGridManager(*context, scene));
    CHECKED_CALL(accelerationStruct->initialize(err));
    CHECKED_CALL(accelerationStruct->initializeFrame(err));
    CHECKED_CALL(accelerationStruct->construct(err));

    const size_t pixelCount = (size_t)window_width * window_height;
    CLBuffer colorsBuffer(*context, pixelCount * sizeof(CL_FLOAT4), CLBufferFlags::ReadWrite);
    std::vector<CL_FLOAT4> colors(pixelCount);
    auto t0 = std::chrono::steady_clock::now();
    for (int f = 0; f < frames; f++) {
        // RTFrame, OpenCLRayTracerDemo.cpp:271-326
        normalizeQuaternion(&cameraOrientation);
        setOrientationAndPos(&camera.viewTransform, cameraOrientation, cameraPosition);
        CHECKED_CALL(accelerationStruct->generateContacts(camera, err));
        if (referenceFrame) {
            // the rest of RTFrame as the reference has it (OpenCLRayTracerDemo.cpp:287-316), statement by statement: every contact the
            // buffer holds comes to the host (its ALLOCATED size: the buffer only grows, :287), the colours are cleared, and
            // cpuShadeBlinnPhong runs per contact with a hit
            int contactsCount = (int)(accelerationStruct->getPrimaryContacts()->getActualSize() / sizeof(struct Contact));
            std::vector<Contact> contacts((size_t)contactsCount);
            CHECKED_CALL(accelerationStruct->getPrimaryContacts()->copyToHost(contacts.data(), err));
            std::memset(colors.data(), 0, pixelCount * sizeof(CL_FLOAT4));
            CL_FLOAT4 color;
            for (int i = 0; i < contactsCount; i++) {
                color.x = 0.0f; color.y = 0.0f; color.z = 0.0f; color.w = 1.0f;
                struct Contact contact = contacts[i];
                if (contact.contactDist > 0 && contact.pixelIndex < pixelCount) {
                    cpuShadeBlinnPhong(&camera, &contact, (char*)scene.getHostSceneData(), color);
                    colors[contact.pixelIndex] = color;
                }
            }
            continue;
        }
        unsigned contactsCount = (unsigned)pixelCount;
        CHECKED_CALL(shadeBlinnPhong(*context, scene, camera, *accelerationStruct->getPrimaryContacts(), contactsCount, colorsBuffer, err));
        CHECKED_CALL(colorsBuffer.copyToHost(colors.data(), err));
    }
    double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / frames;
    size_t hits = 0;
    for (const CL_FLOAT4& c : colors) hits += c.w > 0;
    std::cout << "frames " << frames << "  " << ms << (referenceFrame ? " ms/frame (contacts to the host + host shading loop)" : " ms/frame (incl. read-back)") << "  lit pixels " << hits << std::endl;
    if (!rawContacts.empty()) {
        std::vector<Contact> contacts(pixelCount);
        std::vector<char> all(accelerationStruct->getPrimaryContacts()->getActualSize());
        CHECKED_CALL(accelerationStruct->getPrimaryContacts()->copyToHost(all.data(), err));
        FILE* fc = std::fopen(rawContacts.c_str(), "wb");
        if (fc) { std::fwrite(all.data(), sizeof(Contact), pixelCount, fc); std::fclose(fc); }
    }
    const bool pfm = outPath.size() > 4 && outPath.compare(outPath.size() - 4, 4, ".pfm") == 0;
    if (!outPath.empty() && pfm) {
        // Portable Float Map: the shaded float RGB as it is (the reference hands the float4 colours to a GL VBO,
        // OpenCLRayTracerDemo.cpp:319); "PF", width height, a negative scale = little endian, rows bottom to top
        FILE* fo = std::fopen(outPath.c_str(), "wb");
        if (!fo) { std::perror("fopen"); return 1; }
        std::fprintf(fo, "PF\n%d %d\n-1.0\n", window_width, window_height);
        std::vector<float> row((size_t)window_width * 3);
        for (int y = window_height - 1; y >= 0; y--) {
            for (int x = 0; x < window_width; x++) {
                const CL_FLOAT4& c = colors[(size_t)y * window_width + x];
                row[3 * x] = c.x; row[3 * x + 1] = c.y; row[3 * x + 2] = c.z;
            }
            std::fwrite(row.data(), sizeof(float), row.size(), fo);
        }
        std::fclose(fo);
    } else if (!outPath.empty()) {
        FILE* fo = std::fopen(outPath.c_str(), "wb");
        if (!fo) { std::perror("fopen"); return 1; }
        std::fprintf(fo, "P6\n%d %d\n255\n", window_width, window_height);
        for (const CL_FLOAT4& c : colors) {
            unsigned char px[3] = { (unsigned char)(c.x * 255.0f + 0.5f), (unsigned char)(c.y * 255.0f + 0.5f), (unsigned char)(c.z * 255.0f + 0.5f) };
            std::fwrite(px, 1, 3, fo);
        }
        std::fclose(fo);
    }
    return 0;
}
