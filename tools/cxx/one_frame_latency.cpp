// What the C++ drop-in pays for ONE frame: orbslam_hip::ORBextractor::operator() (include/orbslam_hip.hpp over the C-ABI) on a 640 x 480
// host image, 2000 features -- median / p90 / min of 500 calls.  Build and run on the GPU box (tools/cxx/one_frame_latency.sh).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "orbslam_hip.hpp"

using namespace orbslam_hip;

int main()
{
    const int W = 640, H = 480;
    std::vector<std::vector<uint8_t>> imgs(8, std::vector<uint8_t>((size_t)W * H));
    unsigned s = 12345;
    for (int k = 0; k < 8; ++k)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                s = s * 1664525u + 1013904223u;
                const int checker = ((((x + 3 * k) / 24) + ((y + 2 * k) / 24)) & 1) ? 200 : 60;
                imgs[k][(size_t)y * W + x] = (uint8_t)(checker + (int)((s >> 24) % 9) - 4);
            }
    ORBextractor ex(2000, 1.2f, 8, 20, 7);
    std::vector<KeyPoint> kps;
    std::vector<uint8_t> desc;
    for (int r = 0; r < 50; ++r) ex(ImageView{imgs[r % 8].data(), W, H, W}, ImageView{}, kps, desc);
    if (ex.status() != ORBX_OK) { printf("FAIL %d: %s\n", ex.status(), orbx_last_error()); return 1; }
    std::vector<double> t;
    for (int r = 0; r < 500; ++r) {
        const auto t0 = std::chrono::steady_clock::now();
        ex(ImageView{imgs[r % 8].data(), W, H, W}, ImageView{}, kps, desc);
        t.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    std::sort(t.begin(), t.end());
    printf("C++ ORBextractor::operator() 640x480, 2000 features: median %.4f ms  p90 %.4f  min %.4f  (%zu keypoints)\n", t[250], t[450], t[0], kps.size());
    // the same frames from PINNED memory (a capture buffer allocated with hipHostMalloc): read where they lie, no staging copy
    const size_t nk = kps.size();
    std::vector<uint8_t> d0 = desc;
    uint8_t *pin = nullptr;
    if (hipHostMalloc((void **)&pin, (size_t)8 * W * H, 0) != 0) { printf("hipHostMalloc failed\n"); return 1; }
    for (int k = 0; k < 8; ++k) memcpy(pin + (size_t)k * W * H, imgs[k].data(), (size_t)W * H);
    ex(ImageView{pin + (size_t)(499 % 8) * W * H, W, H, W}, ImageView{}, kps, desc);
    if (kps.size() != nk || desc != d0) { printf("FAIL: pinned source gives other results\n"); return 1; }
    t.clear();
    for (int r = 0; r < 500; ++r) {
        const auto t0 = std::chrono::steady_clock::now();
        ex(ImageView{pin + (size_t)(r % 8) * W * H, W, H, W}, ImageView{}, kps, desc);
        t.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
    }
    std::sort(t.begin(), t.end());
    printf("  ... image already in pinned memory:              median %.4f ms  p90 %.4f  min %.4f\n", t[250], t[450], t[0]);
    hipHostFree(pin);
    return 0;
}
