// apt_host.h -- host-side state shared by the translation units of librender_mi355x.so
// (render_kernels.hip, host_helpers.cpp): the thread-local error record behind apt_last_error() /
// apt_last_status(), and the context object behind apt_context_* (include/render_mi355x.h).
//
// Thread-safety contract of the library (stated in the public header as well):
//   * apt_last_error() / apt_last_status() are per thread: every C-ABI entry point clears the record on entry
//     and sets it on failure, so after a call they describe THAT call, on THAT thread.
//   * An apt_context carries the settings the reference keeps as compile-time constants (the parameters
//     render_do() uses) plus the diagnostics knobs.  Its setters and the snapshot a render call takes of it are
//     serialised by a mutex inside the context: concurrent calls on one context are safe, each call sees a
//     consistent set of values.  Different contexts share nothing.
//   * The context-free entry points (render_do, apt_set_default_params, apt_set_trace_counter,
//     apt_set_refill_lanes) act on one process-wide default context with the same guarantees.
#pragma once
#include <stdint.h>

#include <mutex>

#include "../../include/render_mi355x.h"

namespace apt {

// ---- thread-local error record (defined in host_helpers.cpp) --------------------------------------
void clear_error();                                              // every C-ABI entry point calls this first
int set_error(int code, const char *fmt, const char *detail = ""); // returns `code`

constexpr uint32_t kDefaultRefillLanes = 32; // lanes with an empty ray slot that trigger a wave-wide ray-generate

} // namespace apt

// The opaque type of the public header.  Values are read through snapshot() only.
struct apt_context {
    struct Values {
        apt_render_params params;          // what render_do() renders with (reference defaults until set)
        unsigned long long *trace_counter; // optional device statistics block, or null
        uint32_t refill_lanes;             // APT_FLAG_RETIRE refill threshold of render_frame
    };
    apt_context();
    Values snapshot();                     // consistent copy under the lock
    void set_params(const apt_render_params &p);
    void set_trace_counter(unsigned long long *c);
    void set_refill_lanes(uint32_t lanes);

  private:
    std::mutex m_;
    Values v_;
};

namespace apt {
apt_context &default_context(); // process-wide, constructed on first use (thread-safe static)
}
