// config.json reader with the reference's grammar and typing rules (Code/config.cpp:7-103).
#pragma once
#include <map>
#include <string>
#include <variant>

namespace rt {

class ConfigData {
public:
    // Line-oriented "JSON": `//` comments, one section level ("name": {), `"key": value[,]`.
    // Values: true/false -> bool, contains '.' -> double, otherwise int (config.cpp:64-76).
    void parse_text(const std::string& text);
    bool load_file(const std::string& path);  // false if the file cannot be opened (defaults then apply)

    int get_int(const std::string& key, int def) const;
    double get_double(const std::string& key, double def) const;
    bool get_bool(const std::string& key, bool def) const;

private:
    std::map<std::string, std::variant<int, double, bool>> data_;
};

}  // namespace rt
