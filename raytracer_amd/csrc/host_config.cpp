#include "host_config.h"

#include <fstream>
#include <sstream>

namespace rt {

static std::string trimmed(const std::string& s, const char* ws) {
    size_t b = s.find_first_not_of(ws);
    if (b == std::string::npos) return std::string();
    size_t e = s.find_last_not_of(ws);
    return s.substr(b, e - b + 1);
}

void ConfigData::parse_text(const std::string& text) {
    std::istringstream in(text);
    std::string line, section = "global";  // config.cpp:15
    while (std::getline(in, line)) {
        size_t c = line.find("//");  // config.cpp:18-21 (also cuts "//" inside values, as the reference does)
        if (c != std::string::npos) line.erase(c);
        line = trimmed(line, " \t\r\n");
        if (line.empty() || line == "{" || line == "}") continue;
        parse_line(line, section);
    }
}

bool ConfigData::load_file(const std::string& path) {
    std::ifstream f(path);
    if (!f.is_open()) return false;
    std::stringstream ss;
    ss << f.rdbuf();
    parse_text(ss.str());
    return true;
}

// config.cpp:35-77
void ConfigData::parse_line(std::string line, std::string& section) {
    if (line.back() == '{') {  // section header: "name": {
        size_t q0 = line.find('"');
        size_t q1 = (q0 == std::string::npos) ? std::string::npos : line.find('"', q0 + 1);
        if (q0 != std::string::npos && q1 != std::string::npos) section = line.substr(q0 + 1, q1 - q0 - 1);
        return;
    }
    size_t colon = line.find(':');
    if (colon == std::string::npos) return;
    std::string key_part = line.substr(0, colon), val = line.substr(colon + 1);
    size_t k0 = key_part.find('"'), k1 = key_part.rfind('"');
    if (k0 == std::string::npos || k1 == std::string::npos) return;
    std::string key = key_part.substr(k0 + 1, k1 - k0 - 1);
    if (!val.empty() && val.back() == ',') val.pop_back();
    val = trimmed(val, " \t");
    std::string full = section + "." + key;
    if (val == "true") {
        data_[full] = true;
    } else if (val == "false") {
        data_[full] = false;
    } else if (val.find('.') != std::string::npos) {
        try { data_[full] = std::stod(val); } catch (...) {}
    } else {
        try { data_[full] = std::stoi(val); } catch (...) {}
    }
}

int ConfigData::get_int(const std::string& key, int def) const {
    auto it = data_.find(key);
    if (it != data_.end()) {
        if (auto p = std::get_if<int>(&it->second)) return *p;
        if (auto p = std::get_if<double>(&it->second)) return static_cast<int>(*p);
    }
    return def;
}
double ConfigData::get_double(const std::string& key, double def) const {
    auto it = data_.find(key);
    if (it != data_.end()) {
        if (auto p = std::get_if<double>(&it->second)) return *p;
        if (auto p = std::get_if<int>(&it->second)) return static_cast<double>(*p);
    }
    return def;
}
bool ConfigData::get_bool(const std::string& key, bool def) const {
    auto it = data_.find(key);
    if (it != data_.end())
        if (auto p = std::get_if<bool>(&it->second)) return *p;
    return def;
}

}  // namespace rt
