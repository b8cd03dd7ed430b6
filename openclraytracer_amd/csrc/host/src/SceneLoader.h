// SceneLoader.h -- pure C++ half of Scene::load (no GPU needed): scene file -> packed scene buffer.
#pragma once
#include <string>
#include <vector>
namespace rtmi_host {
// indexBytes = 2: the reference's ushort vertex indices (wrap beyond 65 535 unique vertices per mesh); 4: 32-bit indices (extension)
bool packSceneFile(const char* filename, std::vector<char>& out, std::string& err, int indexBytes = 2);
}
