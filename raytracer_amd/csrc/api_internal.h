// What the translation units behind the C-ABI share with api.cpp: the error string and the device of rt_init.
#pragma once
#include <string>

namespace rt {

int api_fail(int code, const std::string& msg);   // sets rt_last_error() for this thread, returns code
bool api_ready();                                 // rt_init has succeeded
int api_device();                                 // ... on this HIP device

}  // namespace rt
