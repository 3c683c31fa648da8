// internal.h -- helpers shared by the translation units behind include/pqv.h (not part of the boundary).
#pragma once
#include <hip/hip_runtime.h>

#include <string>

namespace pqv_internal {

// records the calling thread's pqv_last_error() message and returns `code`
int fail(int code, const std::string &msg);
// hipSetDevice with the ABI's error convention (PQV_ERR_NO_DEVICE without a usable device)
int use_device(int device);

// No C++ exception may cross the C ABI.
template <class F>
int guard(F &&body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return fail(-4 /* PQV_ERR_OOM */, "host allocation failed");
    } catch (const std::exception &e) {
        return fail(-1 /* PQV_ERR_INVALID */, std::string("internal error: ") + e.what());
    } catch (...) {
        return fail(-1, "internal error");
    }
}

}  // namespace pqv_internal
