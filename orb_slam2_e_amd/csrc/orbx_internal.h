// orbx_internal.h -- extractor state shared by the translation units that work
// on an orbx_extractor's device-resident results (orbx_extract.hip, orbx_stereo.hip).
#ifndef ORBX_INTERNAL_H
#define ORBX_INTERNAL_H

#include <stdint.h>

#include <vector>

#include "common.h"

namespace orbx_detail {

constexpr int EDGE = 19;  // EDGE_THRESHOLD, ORBextractor.cc:74
constexpr int PADX = 32;  // left pad (bytes) of every pyramid row
constexpr int MAXL = 16;
constexpr int MIN_BORDER = EDGE - 3; // minBorderX/Y, ORBextractor.cc:773
constexpr int OCT_T = 256;           // threads of k_octree (large batches)
constexpr int OCT_TW = 512;          // ... of the wide variant small batches run (orbx_extract_batch): same 2,048 register-resident keys, 4 per thread
constexpr int OCT_KPT = 8, OCT_KB = 4; // keys a k_octree thread keeps in registers (levels above OCT_KPT x OCT_T candidates: arrays); keys per batch of LDS reads
constexpr int OCT_MAXN = 2047;       // largest per-level feature quota supported (node state of k_octree in LDS)

struct LevelInfo {
    int w, h, stride, off; // inner size, row stride (bytes), offset of the padded block in a frame
    int W, H;              // octree box = FAST region (maxBorder - minBorder)
    int N;                 // mnFeaturesPerLevel[l]
    int ncells, cell_base; // cell slots [cell_base, cell_base + ncells) of a frame
    int key_base;          // first key slot of this level in a frame's key workspace
    int sel_base;          // first slot of this level in a frame's selected-keypoint array
    int nIni;
    float hX;
    float scale; // mvScaleFactor[l]
    int patch;   // scaledPatchSize = int(31 * scale)
    int xtab, ytab;
};

// What k_fast_cells reads of a cell: 36 bytes, every field dword-aligned within a 16-byte-aligned record, so that the
// wave fetches it with scalar loads (a 2-byte field at an odd dword offset turns the read into vector loads and puts a
// full memory latency in front of the tile loads).
struct FastCell {
    short cw, ch, dx, dy;                     // sub-image size; pt offset of the cell (ORBextractor.cc:828-833)
    short ngx, qstep, rstep, level;           // 4-pixel groups per zone row and the group walk's steps (below); pyramid level
    int cand_off;                             // first candidate slot of the cell in a frame
    int cap;
    // k_fast_cells walks the zone's groups 64 at a time: lane + 64 -> (gx + rstep, y + qstep) with one carry, and lane ->
    // (y, gx) = (lane * inv_ngx >> 16, lane - y ngx), exact for lane < 64 and ngx <= 16 -- divisions done on the host, once
    int inv_ngx;
    int img_off, stride;                      // byte offset of tile pixel (0, 0) = level pixel (x0 - 5, y0) in a frame's pyramid; row pitch
};
struct alignas(16) CellInfo : FastCell {
    short x0, y0;                             // sub-image origin in level coordinates
    int pad;
};

// k_pyr_chain: what one workgroup computes of every pyramid level.  Level l >= 1: the inner rectangle [nx0, nx0 + nw) x [ny0, ny0 + nh)
// it computes into LDS (what it owns of the padded level plus what its deeper levels read) and the dwords x rows of the padded level
// it stores; level 0: the rectangle it loads.  nx0 is a multiple of 4 (LDS dwords line up with the padded row's dwords).
struct ChainRect { short nx0, ny0, nw, nh, oxw0, oxw1, or0, or1; };
struct alignas(16) ChainTile { ChainRect r[MAXL]; };

// a blur tile strip with what k_blur needs of its level: 16 bytes, one scalar load (a tile record pointing into the level
// table was two dependent loads, the first a per-lane one, in front of the window loads)
struct alignas(16) BlurTile { short x0, y0, w, h; int off, stride; int boff, bcol; int pad[2]; };   // boff / bcol: the level in the blurred buffer's strip layout

} // namespace orbx_detail

struct orbx_extractor {
    orbx_params prm;
    int nlevels;
    float scale[orbx_detail::MAXL], inv_scale[orbx_detail::MAXL], sigma2[orbx_detail::MAXL], inv_sigma2[orbx_detail::MAXL];
    int nfeat[orbx_detail::MAXL];
    int taps[4];
    int resident_waves = 8192;              // CUs x 32 wave slots of the device the handle was created on
    int resize_nxi[orbx_detail::MAXL] = {};
    int resize_tailwin[orbx_detail::MAXL] = {}; // the tail tiles of k_pyr_resize may use 8-byte source windows at this level // interior workgroups per row group of k_pyr_resize (0: no fast path at this level)
    int kcap; // keypoints per frame the result arrays hold: nfeatures + 3*nlevels, more on wide frames with tiny quotas (orbx_reserve)
    int kcap_params = 0;
    size_t pin_result_off = 0;   // where the last orbx_extract's result block starts in h_pin
    // orbx_extract_pair: one frame's launch chain (image staging, the kernels, the result block) as a graph, captured on the second
    // call of a frame size and replayed from then on (dropped with the workspace, the pinned block or the input buffer)
    hipGraph_t g_graph = nullptr;
    hipGraphExec_t g_exec = nullptr;
    int g_w = 0, g_h = 0, g_stride = 0, g_seen_w = 0, g_seen_h = 0, g_seen_stride = 0;
    bool g_failed = false;

    // geometry of the reserved workspace
    int width = 0, height = 0, batch = 0;
    orbx_detail::LevelInfo lv[orbx_detail::MAXL];
    std::vector<orbx_detail::CellInfo> cells;
    std::vector<orbx_detail::BlurTile> tiles;
    size_t frame_bytes = 0, cands_per_frame = 0, keys_per_frame = 0;
    // the blurred pyramid (private to k_blur -> k_describe) in STRIPS: a level is lv.stride / 16 column strips of 16 px, a strip's rows
    // one after the other (16 bytes apart), bcol = rows x 16 bytes per strip with the row count rounded up to 8 -- a 128-byte line holds
    // 8 rows x 16 px, and the 37 x 37 window of a descriptor spans a dozen lines instead of 37-74 (k_describe is bound by the texture
    // addresser's line count)
    // the pyramid chain in one launch (k_pyr_chain): tiles per frame, the two LDS buffers' sizes; 0 tiles = the per-level launches
    std::vector<orbx_detail::ChainTile> chain;
    std::vector<uint4> chain_tabs;          // every tile's coefficient tables (rows, then columns, per level), 16-byte units
    std::vector<int2> chain_span;           // first unit, units of a tile
    orbx_detail::ChainTile *d_chain = nullptr;
    uint4 *d_chain_tabs = nullptr;
    int2 *d_chain_span = nullptr;
    int chain_tiles = 0, chain_ldsA = 0, chain_ldsB = 0, chain_ldsT = 0;
    uint32_t *mirror = nullptr; int mirror_desc = 0;   // set around the one-frame call's orbx_extract_batch: k_describe's second copy of the results (pinned host block)
    size_t blur_frame_bytes = 0;
    int boff[orbx_detail::MAXL] = {}, bcol[orbx_detail::MAXL] = {};
    int cells_per_frame = 0, sel_per_frame = 0, maxcells = 0, NC = 0;
    int TS = 0, tile_bytes = 0, SS = 0, sc_bytes = 0, fast_lds = 0, queue_bytes = 0, oct_lds = 0, oct_kcap = 0, oct_kshift = 11;

    hipStream_t stream = nullptr;
    hipStream_t last_stream = nullptr; // the stream the last batch was queued on (the handle's own or the caller's): downloads wait for it only
    hipStream_t st_stream = nullptr;   // ... and the last stereo match
    uint8_t *d_pyr = nullptr, *d_blur = nullptr, *d_in = nullptr;
    size_t in_bytes = 0;
    orbx_detail::LevelInfo *d_lv = nullptr;
    orbx_detail::CellInfo *d_cells = nullptr;
    orbx_detail::BlurTile *d_tiles = nullptr;
    uint4 *d_blur_frag = nullptr; // Toeplitz operand fragments of k_blur, 2 x 16 bytes per lane
    int2 *d_xt = nullptr;
    int4 *d_yt = nullptr;
    int *d_cell_count = nullptr, *d_level_count = nullptr, *d_level_ncand = nullptr, *d_counts = nullptr;
    uint32_t *d_cands = nullptr, *d_kpos = nullptr, *d_sel = nullptr;
    unsigned short *d_knode = nullptr;
    uint8_t *d_kq = nullptr, *d_desc = nullptr;
    orbx_keypoint *d_kps = nullptr;
    int last_batch = 0;
    // stereo (orbx_stereo.hip): results live in the LEFT handle
    uint8_t *h_pin = nullptr;                // pinned staging of the single-image call: image in, count + keypoints + descriptors out
    size_t pin_bytes = 0;
    unsigned *d_st_key = nullptr;
    void *d_st_rk = nullptr; // row bands of the right keypoints (orbx_stereo.hip)
    int *d_st_rowoff = nullptr, *d_st_items = nullptr; // vRowIndices as a CSR table per frame
    uint8_t *h_st_pin = nullptr; size_t st_pin_bytes = 0; // pinned block of orbx_stereo_download
    float *d_uright = nullptr, *d_depth = nullptr, *d_st_scale = nullptr;
    int *d_st_sad = nullptr, *d_st_nvalid = nullptr;
    int st_batch = 0;
    orbx::KernelProfiler prof;
    // ordering of chained *_dev calls that were not given one common stream (orbx_detail::order_after_producer)
    hipEvent_t order_ev = nullptr, reader_ev = nullptr;
    hipStream_t reader_stream = nullptr;
    bool reader_pending = false;
};

namespace orbx_detail {
// A consumer of a handle's device-resident results (orbm_match_batch_dev) running on stream `st`: if `ptr` lies in the
// result buffers of a live extractor whose last batch was queued on ANOTHER stream (e.g. both calls were handed the NULL
// stream: the extractor then uses its handle's stream), `st` is made to wait for that batch.  Returns the producer (or
// nullptr: unknown pointer or same stream -- nothing to order, nothing recorded); reader_done() afterwards makes the
// producer's next batch wait for the consumer in turn.  Same-stream chains pay one table lookup and no event.
orbx_extractor *order_after_producer(const void *ptr, hipStream_t st);
void reader_done(orbx_extractor *ex, hipStream_t st);
} // namespace orbx_detail

#endif // ORBX_INTERNAL_H
