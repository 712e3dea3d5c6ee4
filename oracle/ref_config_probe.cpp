// TEST INFRASTRUCTURE ONLY (oracle).  Harness around the REFERENCE's own Config class (Code/config.{h,cpp}, compiled where it
// lies by oracle/Makefile): loads the config file named by argv[1] and prints, for every key named on stdin as
// `<type> <key>` (type i / d / b), the value the reference's getter returns for two different defaults -- which also tells
// whether the key exists with that type.  Used by oracle/gen_golden_r3.py to write tests/golden/config_cases.json.
#include <cstdio>
#include <iostream>
#include <string>

#include "config.h"

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    Config::Instance().load(argv[1]);
    std::string type, key;
    while (std::cin >> type >> key) {
        if (type == "i") std::printf("i %s %d %d\n", key.c_str(), Config::Instance().getInt(key, -7), Config::Instance().getInt(key, 13));
        else if (type == "d") std::printf("d %s %.17g %.17g\n", key.c_str(), Config::Instance().getDouble(key, -7.5), Config::Instance().getDouble(key, 13.25));
        else std::printf("b %s %d %d\n", key.c_str(), (int)Config::Instance().getBool(key, false), (int)Config::Instance().getBool(key, true));
    }
    return 0;
}
