// Tuning / A-B knobs of the library (gl_set_option keys, include/gligen_hip.h) in ONE table.
//
// gl_set_option writes the PROCESS defaults.  A handle (gl_engine, gl_vae) may carry overrides of individual keys
// (gl_set_handle_option / gl_vae_set_option): while one of its entry points runs on a thread, that thread's lookups see the
// defaults with the handle's overrides applied, so two hosts in one process can tune independently.  Op-level calls
// (gl_gemm, gl_conv3x3, ...) made outside a handle see the defaults.
#pragma once
#include <cstdint>

constexpr int GL_OPT_MAX = 56;
// internal slots (not settable keys): the conv variant of key 5
constexpr int GL_OPT_SPLITK_TILES_CONV = 40;

struct gl_opts {
    int v[GL_OPT_MAX];
};

struct gl_opt_overrides {
    uint64_t mask = 0;            // bit k set: key k is overridden on this handle
    int v[GL_OPT_MAX] = {};
    int epoch = 0;                // bumped by every change: the handle's captured graphs are stale
};

extern gl_opts g_gl_opts;                              // process defaults (misc.hip)
extern int g_gl_option_epoch;                          // bumped by every gl_set_option call
extern thread_local const gl_opts* tl_gl_opts;         // effective table of the handle running on this thread (nullptr: defaults)

static inline int gl_opt(int key) {
    const gl_opts* o = tl_gl_opts;
    return (o ? o : &g_gl_opts)->v[key];
}

// normalises (key, value) exactly like the process-level setter and stores it into `t`; false for an unknown key
bool gl_opts_store(gl_opts& t, int key, int value);

// RAII: a handle's entry point installs [defaults + its overrides] for the duration of the call
struct gl_opts_scope {
    gl_opts eff;
    const gl_opts* prev;
    explicit gl_opts_scope(const gl_opt_overrides& o) : prev(tl_gl_opts) {
        eff = g_gl_opts;
        for (int k = 0; k < GL_OPT_MAX; ++k)
            if ((o.mask >> k) & 1u) gl_opts_store(eff, k, o.v[k]);
        tl_gl_opts = &eff;
    }
    ~gl_opts_scope() { tl_gl_opts = prev; }
    gl_opts_scope(const gl_opts_scope&) = delete;
    gl_opts_scope& operator=(const gl_opts_scope&) = delete;
};
