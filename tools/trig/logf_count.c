/* Every positive finite float x: (1) where does the C library's logf(x) differ from (float)log((double)x), and (2) does the product's
 * restatement orbx_logf_glibc_f32 (orb_slam2_e_amd/csrc/orbx_math.h, compiled by the device code too) equal the library's logf
 * everywhere?  And (3) for the level prediction that uses it -- MapPoint::PredictScale, MapPoint.cc:448-480:
 * ceil(log(ratio) / mfLogScaleFactor), float throughout -- at how many ratios in [1, 1.2^8] do the two logarithms give different levels
 * for the reference's scale factor 1.2 (mfLogScaleFactor = logf(1.2f))?
 * Run with GLIBC_TUNABLES=glibc.cpu.hwcaps=-FMA,-AVX2 as well: the library then uses its generic build; counts must not change.
 * build: gcc -O2 -fopenmp -ffp-contract=off tools/trig/logf_count.c -o /tmp/logf_count -lm        run: /tmp/logf_count */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../orb_slam2_e_amd/csrc/orbx_math.h"

static float as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static uint32_t as_u32(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

int main(void)
{
    long long differs = 0, restated_wrong = 0, level_differs = 0, in_range = 0;
    const float ls = logf(1.2f), top = 4.2998f;           /* 1.2^8 */
    uint32_t example = 0;
#pragma omp parallel for schedule(static) reduction(+ : differs, restated_wrong, level_differs, in_range)
    for (uint32_t u = 1; u < 0x7f800000u; ++u) {
        const float x = as_float(u);
        const float l1 = logf(x), l2 = (float)log((double)x), l3 = orbx_logf_glibc_f32(x);
        const int d = as_u32(l1) != as_u32(l2);
        differs += d;
        restated_wrong += as_u32(l3) != as_u32(l1);
        if (x >= 1.0f && x <= top) {
            in_range++;
            if (ceilf(l1 / ls) != ceilf(l2 / ls)) { level_differs++; example = u; }
        }
    }
    printf("{\"floats_tried\": %u, \"logf_differs_from_rounded_double_log\": %lld, \"restatement_vs_libc_logf\": %lld, "
           "\"ratios_in_1_to_1.2pow8\": %lld, \"predicted_level_differs_at_scale_1.2\": %lld, \"example_ratio\": \"%a\"}\n",
           0x7f800000u - 1u, differs, restated_wrong, in_range, level_differs, (double)as_float(example));
    return 0;
}
