// p3d_error.hpp — thread-local "last error" shared by the host and device halves of libp3d.
#pragma once
#include <string>

namespace p3d {
// Records msg as the calling thread's last error and returns code (a negative p3d_status).
int fail(int code, const std::string& msg);
const char* last_error_cstr();
}  // namespace p3d
