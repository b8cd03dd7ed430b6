// SceneLoader.h -- pure C++ half of Scene::load (no GPU needed): scene file -> packed scene buffer.
#pragma once
#include <string>
#include <vector>
namespace rtmi_host {
bool packSceneFile(const char* filename, std::vector<char>& out, std::string& err);
}
