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
//     apt_set_refill_lanes, apt_set_debug) act on one process-wide default context with the same guarantees.
//   * Nothing in a launch path reads the process environment: the APT_* measurement knobs are read ONCE, when a
//     context is created, into its Debug values; after that they change only through apt_context_set_debug().
#pragma once
#include <stdint.h>

#include <mutex>

#include "../../include/render_mi355x.h"

// everything the library exports is named in apt_exports.map; APT_API marks the C++-linkage render_do forwarder
#define APT_API __attribute__((visibility("default")))

namespace apt {

// ---- thread-local error record (defined in host_helpers.cpp) --------------------------------------
void clear_error();                                              // every C-ABI entry point calls this first
int set_error(int code, const char *fmt, const char *detail = ""); // returns `code`

constexpr uint32_t kDefaultRefillLanes = 32; // lanes with an empty ray slot that trigger a wave-wide ray-generate
constexpr int kMaxStatusDevices = 16;        // devices a context keeps a status word for

// Measurement knobs (apt_context_set_debug; 0 = the library's own choice).  Speed only, except grid_walk, which picks
// between two bit-identical traversals.
struct Debug {
    uint32_t queue_ppw = 0;            // pixels per wave of the sample-queue kernels (1..4096)
    uint32_t queue_nbuf = 0;           // colour buffers of the sample-queue kernels (2..16)
    uint32_t queue_lds_pad = 0;        // extra dynamic LDS per wave (lowers the occupancy)
    uint32_t grid_walk = 0;            // 1 = frames of a scene behind a grid take render_frame_kernel's nested item walk only
    double grid_spheres_per_cell = 0;  // cell size of both grid builders (sphere centres per cell)
};

} // namespace apt

// The opaque type of the public header.  Values are read through snapshot() only.
struct apt_context {
    struct Values {
        apt_render_params params;          // what render_do() renders with (reference defaults until set)
        unsigned long long *trace_counter; // optional device statistics block, or null
        uint32_t refill_lanes;             // APT_FLAG_RETIRE refill threshold of render_frame
        apt::Debug debug;
    };
    apt_context();
    Values snapshot();                     // consistent copy under the lock
    void set_params(const apt_render_params &p);
    void set_trace_counter(unsigned long long *c);
    void set_refill_lanes(uint32_t lanes);
    int set_debug(const char *key, double value); // APT_OK / APT_ERR_ARG (error record set)
    int get_debug(const char *key, double *value); // the knob's current value (what set_debug last stored, or the environment's initial value)

    // The device status word of this context on device `dev` (render_kernels.hip: kernels OR failure bits into it, apt_context_check()
    // reads and clears it).  Device memory the context owns; `lookup` only returns what exists, `adopt` stores a freshly made word and
    // returns the one that stands (the loser of a race is handed back through *spare for the caller to free), `release` hands every
    // word to the caller (apt_context_destroy frees them; the process-wide default context keeps its words until the process ends).
    uint32_t *status_lookup(int dev);
    uint32_t *status_adopt(int dev, uint32_t *fresh, uint32_t **spare);
    void status_release(uint32_t *out[apt::kMaxStatusDevices]);

  private:
    std::mutex m_;
    Values v_;
    uint32_t *status_[apt::kMaxStatusDevices] = {};
};

namespace apt {
apt_context &default_context(); // process-wide, constructed on first use (thread-safe static)
}
