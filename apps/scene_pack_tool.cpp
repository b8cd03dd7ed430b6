// scene_pack_tool -- CPU-only: runs the product's Scene::load packing and writes the packed scene buffer.
// Used by tests to compare the C++ loader with the oracle's packer and the committed fixtures.
#include <cstdio>
#include <string>
#include <vector>
#include "../openclraytracer_amd/csrc/host/src/SceneLoader.h"

int main(int argc, char** argv)
{
    const bool index32 = argc == 4 && std::string(argv[3]) == "--index32";
    if (argc != 3 && !index32) { std::fprintf(stderr, "usage: %s scene.txt out.bin [--index32]\n", argv[0]); return 2; }
    std::vector<char> buf; std::string err;
    if (!rtmi_host::packSceneFile(argv[1], buf, err, index32 ? 4 : 2)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
    FILE* f = std::fopen(argv[2], "wb");
    if (!f) { std::perror("fopen"); return 1; }
    std::fwrite(buf.data(), 1, buf.size(), f);
    std::fclose(f);
    return 0;
}
