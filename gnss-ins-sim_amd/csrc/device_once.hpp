// "Once per DEVICE" for launch-side function attributes.  hipFuncSetAttribute acts on the current device's copy of a kernel,
// so a process that drives several GPUs (ginsim.multi: one context and one host thread per device) has to repeat it on every
// device it launches on -- a function-local `static bool once` would configure the first device only and the 100 KB dynamic-LDS
// launches of the wave-specialised kernels would be refused on the others.  Thread safe: the check and the attribute call sit
// under one lock, so a second thread on the same device cannot launch before the attribute is in place.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>

namespace ginsim {

class PerDeviceOnce {
public:
    template <class F>
    void run(F&& configure) {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 256) { configure(); return; }
        std::lock_guard<std::mutex> g(m_);
        const uint64_t bit = 1ull << (d & 63);
        if (done_[d >> 6] & bit) return;
        configure();
        done_[d >> 6] |= bit;
    }
private:
    std::mutex m_;
    uint64_t done_[4] = {0, 0, 0, 0};
};

}  // namespace ginsim
