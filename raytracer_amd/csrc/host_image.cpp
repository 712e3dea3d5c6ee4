#include "host_image.h"

#include <spawn.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cerrno>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <stdexcept>

extern char** environ;

namespace rt {

static void skip_blanks(std::istream& f) {
    int c = f.peek();
    while (f && (c == '\n' || c == '\r' || c == ' ' || c == '\t')) {
        f.get();
        c = f.peek();
    }
}

// Image.cpp:84-167.  One optional '#' comment line after the magic number; maxval must be 255; exactly one
// separator byte before the payload.
Image8 read_ppm(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("Cannot open file: " + path);
    std::string magic;
    f >> magic;
    if (magic != "P6" && magic != "P3") throw std::runtime_error("Invalid PPM file format. Expected P6 (Binary) or P3 (ASCII).");
    skip_blanks(f);
    if (f && f.peek() == '#') {
        std::string comment;
        std::getline(f, comment);
        skip_blanks(f);
    }
    Image8 img;
    int maxval = 0;
    f >> img.width >> img.height >> maxval;
    if (maxval != 255) throw std::runtime_error("Unsupported max color value. Only 255 is supported.");
    if (img.width <= 0 || img.height <= 0) throw std::runtime_error("Invalid PPM dimensions: " + path);
    f.get();
    size_t n = static_cast<size_t>(img.width) * img.height * 3;
    img.rgb.resize(n);
    if (magic == "P6") {
        f.read(reinterpret_cast<char*>(img.rgb.data()), static_cast<std::streamsize>(n));
        if (!f) throw std::runtime_error("Error reading pixel data from file: " + path + (f.eof() ? ". File ended unexpectedly." : "."));
    } else {
        for (size_t i = 0; i < n; ++i) {
            int v;
            if (!(f >> v)) throw std::runtime_error("Error reading pixel data from P3 file: " + path + ". Expected more color values.");
            img.rgb[i] = static_cast<unsigned char>(v);  // Image.cpp:159: no clamping, plain narrowing
        }
    }
    return img;
}

// Image.cpp:170-188
void write_ppm(const std::string& path, int width, int height, const uint8_t* rgb) {
    std::ofstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("Cannot open file for writing: " + path);
    f << "P6\n" << width << " " << height << "\n" << 255 << "\n";
    f.write(reinterpret_cast<const char*>(rgb), static_cast<std::streamsize>(static_cast<size_t>(width) * height * 3));
    if (!f) throw std::runtime_error("Error writing pixel data to file: " + path);
}

// HDRImage.cpp:22-109.  "PF", comments, W H, scale (sign = endianness, magnitude = multiplier), one newline, floats.
bool read_pfm(const std::string& path, ImageF& out, std::string* why) {
    auto fail = [&](const std::string& m) {
        if (why) *why = m;
        out = ImageF();
        return false;
    };
    std::ifstream f(path, std::ios::binary);
    if (!f.is_open()) return fail("Could not open PFM file " + path);
    std::string header;
    f >> header;
    if (header != "PF") return fail("Unsupported PFM format (must be 'PF'): " + path);
    while (f >> std::ws) {
        if (f.peek() == '#') {
            std::string comment;
            std::getline(f, comment);
        } else {
            break;
        }
    }
    int w = 0, h = 0;
    float scale = 0.f;
    f >> w >> h;
    f >> scale;
    char ch;
    while (f.get(ch)) {
        if (ch == '\n') break;
        if (!std::isspace(static_cast<unsigned char>(ch))) return fail("Unexpected garbage in PFM header.");
    }
    if (w <= 0 || h <= 0) return fail("Invalid dimensions.");
    out.width = w;
    out.height = h;
    out.rgb.resize(static_cast<size_t>(w) * h * 3);
    f.read(reinterpret_cast<char*>(out.rgb.data()), static_cast<std::streamsize>(out.rgb.size() * sizeof(float)));
    if (!f) return fail("File ended prematurely.");
    if (scale > 0.0f) {  // big-endian payload
        for (float& v : out.rgb) {
            char* b = reinterpret_cast<char*>(&v);
            std::swap(b[0], b[3]);
            std::swap(b[1], b[2]);
        }
    }
    float a = scale < 0 ? -scale : scale;
    if (a != 1.0f && a != 0.0f)
        for (float& v : out.rgb) v *= a;
    return true;
}

// Runs `interpreter -c <fixed script> <src> <dst>` without a shell (the reference splices the texture path into a
// `python3 -c "..."` command line for std::system, scene.cpp:46-66: a file name with a quote, a backtick or $(...) in a scene
// file would run as shell or Python code).  The paths travel as sys.argv, never as program text.
static bool convert_with_pil(const char* interpreter, const std::string& src, const std::string& dst) {
    static const char* kScript =
        "import sys\n"
        "try:\n"
        "    from PIL import Image\n"
        "    Image.open(sys.argv[1]).convert('RGB').save(sys.argv[2], format='PPM')\n"
        "except Exception as e:\n"
        "    print(e)\n"
        "    sys.exit(1)\n";
    char* const argv[] = {const_cast<char*>(interpreter), const_cast<char*>("-c"), const_cast<char*>(kScript),
                          const_cast<char*>(src.c_str()), const_cast<char*>(dst.c_str()), nullptr};
    // what the child prints (PIL's error text for a missing or broken file) goes to stderr: stdout belongs to the caller
    posix_spawn_file_actions_t fa;
    if (posix_spawn_file_actions_init(&fa) != 0) return false;
    posix_spawn_file_actions_adddup2(&fa, 2, 1);
    pid_t pid = 0;
    const int rc = posix_spawnp(&pid, interpreter, &fa, nullptr, argv, environ);
    posix_spawn_file_actions_destroy(&fa);
    if (rc != 0) return false;
    int status = 0;
    while (waitpid(pid, &status, 0) < 0)
        if (errno != EINTR) return false;
    return WIFEXITED(status) && WEXITSTATUS(status) == 0;
}

bool load_texture(const std::string& path, Image8& out, std::string* why) {
    std::string ext;
    size_t dot = path.find_last_of('.');
    if (dot != std::string::npos) {
        ext = path.substr(dot);
        for (char& c : ext) c = static_cast<char>(std::tolower(static_cast<unsigned char>(c)));
    }
    std::string final_path = path, temp;
    if (ext == ".jpg" || ext == ".jpeg" || ext == ".png") {
        // Same conversion the reference performs (PIL: open, convert('RGB'), save as PPM; python3, else python), into a
        // temporary file of its own (mkstemp: unpredictable name, no collision between threads or processes).
        char name[] = "/tmp/rt_texture_XXXXXX";
        const int fd = mkstemp(name);
        if (fd >= 0) {
            close(fd);
            temp = name;
            if (convert_with_pil("python3", path, temp) || convert_with_pil("python", path, temp)) final_path = temp;
        }
    }
    bool ok = true;
    try {
        out = read_ppm(final_path);
    } catch (const std::exception& e) {
        if (why) *why = e.what();
        out = Image8();
        ok = false;
    }
    if (!temp.empty()) std::remove(temp.c_str());
    return ok;
}

}  // namespace rt
