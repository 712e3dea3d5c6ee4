#include "host_config.h"

#include <fstream>
#include <optional>
#include <sstream>
#include <string_view>

// config.json as the reference reads it (Code/config.cpp:7-103).  The format is "JSON by convention": the reference never
// parses JSON, it looks at one LINE at a time, and what it accepts is decided by a handful of character searches.  Here the
// same language is written as three small pieces -- a line splitter, a statement recogniser and a value typer -- whose
// corner cases (pinned by tests/test_abi_host.py) are the reference's:
//   * everything from the first "//" of a line on is a comment, inside values too;
//   * a line whose last character is '{' names a section (the text between its first two quotes); one level, no nesting;
//   * any other line with a ':' is `"key": value` -- the key runs from the FIRST to the LAST quote left of the first ':'
//     (a single quote yields everything behind it), ONE trailing ',' is dropped from the value;
//   * the value's text decides its type: the words true / false, a '.' anywhere -> the leading floating-point number, else
//     the leading integer; text that starts with neither, or an integer that does not fit, is ignored.

namespace rt {

namespace {

constexpr std::string_view kLineBlanks = " \t\r\n", kValueBlanks = " \t";

std::string_view strip(std::string_view s, std::string_view blanks) {
    const size_t first = s.find_first_not_of(blanks);
    if (first == std::string_view::npos) return {};
    return s.substr(first, s.find_last_not_of(blanks) - first + 1);
}

// The text between the first quote of `s` and a closing one: the next quote (sections) or the last quote of `s` (keys).
// No quote at all: nothing.  A key with ONE quote closes on itself and yields the rest of the text (length -1 wraps to
// "everything", config.cpp:52-55).
std::optional<std::string_view> quoted(std::string_view s, bool close_on_last) {
    const size_t open = s.find('"');
    if (open == std::string_view::npos) return std::nullopt;
    const size_t close = close_on_last ? s.rfind('"') : s.find('"', open + 1);
    if (close == std::string_view::npos) return std::nullopt;
    return s.substr(open + 1, close - open - 1);
}

struct Statement {
    enum Kind { kNothing, kSection, kEntry } kind = kNothing;
    std::string_view name, value;
};

// One stripped, non-empty line -> what it says.
Statement recognise(std::string_view line) {
    Statement st;
    if (line.back() == '{') {
        if (const auto name = quoted(line, false)) {
            st.kind = Statement::kSection;
            st.name = *name;
        }
        return st;
    }
    const size_t colon = line.find(':');
    if (colon == std::string_view::npos) return st;
    const auto key = quoted(line.substr(0, colon), true);
    if (!key) return st;
    std::string_view value = line.substr(colon + 1);
    if (!value.empty() && value.back() == ',') value.remove_suffix(1);
    st.kind = Statement::kEntry;
    st.name = *key;
    st.value = strip(value, kValueBlanks);
    return st;
}

using Value = std::variant<int, double, bool>;

std::optional<Value> typed(std::string_view text) {
    if (text == "true") return Value(true);
    if (text == "false") return Value(false);
    const std::string s(text);
    try {
        if (text.find('.') != std::string_view::npos) return Value(std::stod(s));   // leading number, as std::stod reads it
        return Value(std::stoi(s));
    } catch (...) {   // not a number, or out of range: the reference swallows the exception and keeps no entry
        return std::nullopt;
    }
}

}  // namespace

void ConfigData::parse_text(const std::string& text) {
    std::string section = "global";   // config.cpp:15
    std::string_view rest(text);
    while (!rest.empty()) {
        const size_t nl = rest.find('\n');
        std::string_view line = rest.substr(0, nl);
        rest = nl == std::string_view::npos ? std::string_view() : rest.substr(nl + 1);
        line = strip(line.substr(0, line.find("//")), kLineBlanks);
        if (line.empty() || line == "{" || line == "}") continue;
        const Statement st = recognise(line);
        if (st.kind == Statement::kSection) {
            section = std::string(st.name);
        } else if (st.kind == Statement::kEntry) {
            if (const auto v = typed(st.value)) data_[section + "." + std::string(st.name)] = *v;
        }
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

// The getters convert between the two number types and nothing else (config.cpp:79-103): an int entry read as double and a
// double entry read as int (truncated) work, a bool is only ever a bool, and any other mismatch yields the default.
namespace {
template <class Want>
std::optional<Want> as_number(const std::variant<int, double, bool>& v) {
    if (const int* i = std::get_if<int>(&v)) return static_cast<Want>(*i);
    if (const double* d = std::get_if<double>(&v)) return static_cast<Want>(*d);
    return std::nullopt;
}
}  // namespace

int ConfigData::get_int(const std::string& key, int def) const {
    const auto it = data_.find(key);
    return it == data_.end() ? def : as_number<int>(it->second).value_or(def);
}
double ConfigData::get_double(const std::string& key, double def) const {
    const auto it = data_.find(key);
    return it == data_.end() ? def : as_number<double>(it->second).value_or(def);
}
bool ConfigData::get_bool(const std::string& key, bool def) const {
    const auto it = data_.find(key);
    if (it == data_.end()) return def;
    const bool* b = std::get_if<bool>(&it->second);
    return b ? *b : def;
}

}  // namespace rt
