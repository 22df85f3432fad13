// context_threads.cpp -- ThreadSanitizer driver for apt_context (csrc/apt_host.h, host_helpers.cpp): the two-thread context test of
// tests/test_gpu_boundary.py against a STUB launch.  Two threads share one context and hammer its setters while two others take
// snapshots the way a render call does ("launch" = snapshot + a consistency check of what it got); two more use private contexts and
// the per-thread error record.  Built with -fsanitize=thread by tests/test_host_sanitizers.py; no GPU, no HIP.
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <thread>
#include <vector>

#include "../../ascendpathtracing_amd/csrc/apt_host.h"

static std::atomic<long> g_bad{0};

// what do_render_paths / do_render_frame do with a context before they launch
static void stub_launch(apt_context &ctx) {
    const apt_context::Values v = ctx.snapshot();
    // writers only ever store parameter sets with width == height == samples * 16 (one set_params call each): a torn read shows
    if (v.params.width != v.params.height || v.params.width != v.params.samples * 16u) ++g_bad;
    if (v.refill_lanes < 1 || v.refill_lanes > 64 || v.debug.queue_ppw > 64) ++g_bad;
    (void)ctx.status_lookup(0);
}

int main() {
    apt_context shared;
    uint32_t words[4] = {0, 0, 0, 0};           // stand-ins for device status words
    std::vector<std::thread> ts;
    for (int t = 0; t < 2; ++t)
        ts.emplace_back([&, t] {
            for (uint32_t i = 1; i <= 20000; ++i) {
                apt_render_params p;
                apt_default_params(&p);
                p.samples = 1 + (i + (uint32_t)t) % 7; p.width = p.height = p.samples * 16u;
                shared.set_params(p);
                const uint32_t lanes = 1 + i % 64;
                shared.set_refill_lanes(lanes);
                if (shared.set_debug("queue_ppw", lanes) != APT_OK) ++g_bad;
                if (shared.set_debug("bogus", 1) != APT_ERR_ARG || apt_last_status() != APT_ERR_ARG) ++g_bad;   // per-thread record
                apt::clear_error();
                uint32_t *spare = nullptr;
                uint32_t *w = shared.status_adopt(0, &words[t], &spare);
                if (w != &words[0] && w != &words[1]) ++g_bad;
            }
        });
    for (int t = 0; t < 2; ++t)
        ts.emplace_back([&] { for (int i = 0; i < 40000; ++i) stub_launch(shared); });
    for (int t = 0; t < 2; ++t)
        ts.emplace_back([&, t] {                 // private contexts share nothing; the default context is made exactly once
            apt_context mine;
            for (uint32_t i = 0; i < 20000; ++i) {
                apt_render_params p;
                apt_default_params(&p);
                p.samples = 2 + (uint32_t)t; p.width = p.height = p.samples * 16u;
                mine.set_params(p);
                stub_launch(mine);
                if (mine.snapshot().params.samples != 2u + (uint32_t)t) ++g_bad;
                (void)apt::default_context().snapshot();
                if (apt::set_error(APT_ERR_IO, "thread %s", t ? "b" : "a") != APT_ERR_IO || strcmp(apt_last_error(), t ? "thread b" : "thread a") != 0) ++g_bad;
            }
        });
    for (auto &t : ts) t.join();
    uint32_t *out[apt::kMaxStatusDevices];
    shared.status_release(out);
    if (out[0] != &words[0] && out[0] != &words[1]) ++g_bad;
    if (g_bad.load()) { fprintf(stderr, "FAILED: %ld inconsistent observations\n", g_bad.load()); return 1; }
    printf("ok\n");
    return 0;
}
