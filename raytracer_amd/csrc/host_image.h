// Image file I/O kept on the host: PPM (P3/P6) read, P6 write, PFM read, JPG/PNG texture conversion.
// Formats and error behaviour follow Code/utilities/Image.cpp:84-188, Code/environment/HDRImage.cpp:22-109
// and Code/utilities/scene.cpp:31-91.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace rt {

struct Image8 {
    int width = 0, height = 0;
    std::vector<uint8_t> rgb;  // row 0 = top, 3 bytes per pixel
};
struct ImageF {
    int width = 0, height = 0;
    std::vector<float> rgb;  // rows in file order
};

// Throws std::runtime_error with the reference's messages on malformed input.
Image8 read_ppm(const std::string& path);
void write_ppm(const std::string& path, int width, int height, const uint8_t* rgb);
// Returns false (and leaves `out` empty) on any failure, like HDRImage::load.
bool read_pfm(const std::string& path, ImageF& out, std::string* why = nullptr);
// scene.cpp:31-91: .jpg/.jpeg/.png are converted to a temporary PPM by shelling out to python3/python + PIL;
// returns false when nothing could be loaded (the reference then renders the shape untextured).
bool load_texture(const std::string& path, Image8& out, std::string* why = nullptr);

}  // namespace rt
