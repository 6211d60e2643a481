// Test programs only: append the kernel symbols this process launched through the library (toyni_launched_kernels, an in-memory
// list) to $TOYNI_LAUNCH_LOG, where tests/test_zz_kernel_coverage.py counts them.  The library itself writes no file.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "toyni_hip.h"

inline void toyni_test_dump_launched_kernels() {
    const char* path = std::getenv("TOYNI_LAUNCH_LOG");
    if (!path) return;
    const size_t need = toyni_launched_kernels(nullptr, 0);
    std::vector<char> buf(need + 1, 0);
    toyni_launched_kernels(buf.data(), buf.size());
    if (FILE* f = std::fopen(path, "a")) { std::fputs(buf.data(), f); std::fclose(f); }
}
