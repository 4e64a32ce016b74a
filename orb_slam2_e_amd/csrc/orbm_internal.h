// orbm_internal.h -- pieces shared by the matcher translation units (orbm_match.hip,
// orbm_search.hip): the Frame grid constants, the keypoint / query records of the
// windowed searches and small host / device helpers.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <vector>

#include "../../include/orbslam_hip.h"
#include "common.h"

namespace orbm_detail {

constexpr int MT = 256; // threads per block of the small helper kernels
constexpr int FRAME_GRID_ROWS = 48, FRAME_GRID_COLS = 64; // include/Frame.h:37-38

// order = grid cell << 16 | keypoint index: ascending order = the order in which
// Frame::GetFeaturesInArea (src/Frame.cc:342-395) lists its result (column-major cells,
// insertion order inside a cell); 0xffffffff = not in the grid / skipped.
struct WinKp { float x, y, uright; int octave; unsigned order; };
struct WinQuery { float u, v, r, xr; int min_level, max_level; };
// Sorted keypoint record of the windowed searches: position sp in the array = rank in GetFeaturesInArea order.
struct SeqKp { float x, y, uright; int octave; };
struct GridParams { float min_x, min_y, inv_w, inv_h; };
static_assert(sizeof(WinQuery) == sizeof(orbm_window_query), "query layout");

__device__ __forceinline__ int popc256(const uint4 &a0, const uint4 &a1, const uint4 &b0, const uint4 &b1)
{
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
           __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

// Frame::AssignFeaturesToGrid / PosInGrid (src/Frame.cc:245-260,397-407), on the host: n float ops.
inline void build_winkp(const orbx_keypoint *kps, int n, const uint8_t *skip, const float *uright, float min_x, float min_y,
                        float max_x, float max_y, std::vector<WinKp> &wk)
{
    const float invW = (float)FRAME_GRID_COLS / (max_x - min_x), invH = (float)FRAME_GRID_ROWS / (max_y - min_y);
    wk.resize(n ? n : 1);
    for (int j = 0; j < n; ++j) {
        const int px = (int)roundf((kps[j].x - min_x) * invW), py = (int)roundf((kps[j].y - min_y) * invH);
        const bool in = !(px < 0 || px >= FRAME_GRID_COLS || py < 0 || py >= FRAME_GRID_ROWS);
        wk[j].x = kps[j].x; wk[j].y = kps[j].y; wk[j].octave = kps[j].octave;
        wk[j].uright = uright ? uright[j] : -1.0f;
        wk[j].order = (in && !(skip && skip[j])) ? ((unsigned)(px * FRAME_GRID_ROWS + py) << 16) | (unsigned)j : 0xffffffffu;
    }
}

// Per-call scratch without per-call hipMalloc: a pool of workspaces (device arena, pinned
// host staging arena, a growable device buffer for candidate entries, a stream).  A call
// leases one, carves its arrays out of the arenas (staged inputs sit at the same offsets
// in both, so ONE copy uploads them) and releases the lease on return.  Concurrent
// callers (Tracking / LocalMapping / LoopClosing threads) get different workspaces.
// Workspaces live until process exit.
struct Workspace {
    char *dev = nullptr, *pin = nullptr;
    unsigned *ent = nullptr;
    size_t dev_cap = 0, pin_cap = 0, ent_cap = 0, used = 0;
    hipStream_t st = nullptr;
    unsigned gen = 0;    // call counter: flags a kernel raises are this number (a fresh arena is zeroed, 0 is never a generation)
    int reserve(size_t dev_bytes, size_t pin_bytes); // discards the contents
    int reserve_entries(size_t n);
    hipStream_t own_stream() { if (!st && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) st = nullptr; return st; }   // nullptr = failed (never hand out the legacy stream)
    size_t carve(size_t bytes) { const size_t o = used; used += (bytes + 255) & ~(size_t)255; return o; }
    template <typename T> T *d(size_t off) const { return reinterpret_cast<T *>(dev + off); }
    template <typename T> T *h(size_t off) const { return reinterpret_cast<T *>(pin + off); }
};
Workspace *workspace_acquire();
void workspace_release(Workspace *w);
struct WorkspaceLease {
    Workspace *w;
    WorkspaceLease() : w(workspace_acquire()) {}
    ~WorkspaceLease() { workspace_release(w); }
};

// One host-array call: declare inputs (staged in pinned memory, uploaded with ONE copy), device scratch and outputs
// (downloaded with ONE copy), in that order; everything runs on the leased workspace's stream.
struct StagedCall {
    WorkspaceLease lease;
    Workspace &w;
    struct In { size_t off; const void *src; size_t bytes; };
    std::vector<In> ins;
    size_t staged = 0, res_off = 0, res_bytes = 0;
    StagedCall() : w(*lease.w) { w.used = 0; }
    size_t in(const void *src, size_t bytes)
    {
        const size_t o = w.carve(bytes ? bytes : 1);
        if (src && bytes) ins.push_back({o, src, bytes});
        staged = w.used;
        return o;
    }
    void in_at(size_t off, const void *src, size_t bytes) { if (src && bytes) ins.push_back({off, src, bytes}); } // inside an in(nullptr, n) block
    size_t scratch(size_t bytes) { return w.carve(bytes ? bytes : 1); }
    size_t out(size_t bytes)
    {
        if (!res_bytes) res_off = w.used;
        const size_t o = w.carve(bytes ? bytes : 1);
        res_bytes = w.used - res_off;
        return o;
    }
    hipStream_t stream() const { return w.st; }
    template <typename T> T *d(size_t off) const { return reinterpret_cast<T *>(w.dev + off); }
    template <typename T> const T *r(size_t off) const { return reinterpret_cast<const T *>(w.pin + (off - res_off)); }
    int upload() // after the last in / scratch / out
    {
        if (w.reserve(w.used, staged > res_bytes ? staged : res_bytes)) return -1;
        for (const In &i : ins) memcpy(w.pin + i.off, i.src, i.bytes);
        return orbx::stage_in(w.dev, w.pin, staged, w.st) != hipSuccess ? -1 : 0;
    }
    int download()
    {
        if (orbx::stage_out(w.pin, w.dev + res_off, res_bytes, w.st) != hipSuccess) return -1;
        return hipStreamSynchronize(w.st) == hipSuccess ? 0 : -1;
    }
};

} // namespace orbm_detail
