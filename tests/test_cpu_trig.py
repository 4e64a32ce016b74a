"""The two evaluations of cos / sin that orb_slam2_e_amd/csrc/orbx_math.h offers the descriptor kernel, compiled for the host and
checked against the C library on a sample (tools/trig/trig_variant_count.c does every float in [0, 2 pi]; a few seconds on eight
cores -- run by hand, its figures are in DESIGN 4.2): variant 0 must equal cosf / sinf, variant 1 the rounded double functions."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "orbx_math.h"
static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
int main(void)
{
    long bad0 = 0, bad1 = 0, differ = 0, n = 0;
    uint32_t u, s = 12345u;
    /* every float in [1, 2) and [4, 2 pi], every 97th float below, and the edges of the small-argument branches */
    for (u = 0; u <= 0x40c90fdbu; u += (u >= 0x3f800000u && u < 0x40000000u) || u >= 0x40800000u ? 1 : 97) {
        float x, gs, gc, rs, rc;
        memcpy(&x, &u, 4);
        orbx_sincos_glibc_f32(x, &gs, &gc);
        orbx_sincos_f32(x, &rs, &rc);
        bad0 += bits(gs) != bits(sinf(x)) || bits(gc) != bits(cosf(x));
        bad1 += bits(rs) != bits((float)sin((double)x)) || bits(rc) != bits((float)cos((double)x));
        differ += bits(gs) != bits(rs) || bits(gc) != bits(rc);
        n++;
    }
    (void)s;
    printf("%ld %ld %ld %ld\n", n, bad0, bad1, differ);
    return 0;
}
"""


def test_both_trig_variants_against_the_c_library(tmp_path):
    c = tmp_path / "t.c"
    c.write_text(SRC)
    exe = str(tmp_path / "t")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-I", os.path.join(ROOT, "orb_slam2_e_amd", "csrc"), str(c), "-o", exe, "-lm"])
    n, bad0, bad1, differ = (int(v) for v in subprocess.check_output([exe], text=True).split())
    assert n > 2.0e7
    assert bad0 == 0, "orbx_sincos_glibc_f32 differs from the C library's cosf / sinf (glibc >= 2.28 expected)"
    assert bad1 == 0
    assert 0 < differ < n // 50          # the two variants do differ, by one ulp, at well under a percent or two of the arguments


SRC_LOG = r"""
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "orbx_math.h"
static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
int main(void)
{
    long bad = 0, differ = 0, n = 0;
    uint32_t u;
    /* every float in [0.5, 8) -- PredictScale's ratios live in [1, 1.2^8] --, every 251st float elsewhere, subnormals included */
    for (u = 1; u < 0x7f800000u; u += (u >= 0x3f000000u && u < 0x41000000u) ? 1 : 251) {
        float x;
        memcpy(&x, &u, 4);
        bad += bits(orbx_logf_glibc_f32(x)) != bits(logf(x));
        differ += bits(logf(x)) != bits((float)log((double)x));
        n++;
    }
    {
        const float z = 0.0f, inf = 1.0f / z;
        bad += orbx_logf_glibc_f32(1.0f) != 0.0f || orbx_logf_glibc_f32(inf) != inf || orbx_logf_glibc_f32(z) != -inf;
        bad += orbx_logf_glibc_f32(-1.0f) == orbx_logf_glibc_f32(-1.0f);     /* NaN */
    }
    printf("%ld %ld %ld\n", n, bad, differ);
    return 0;
}
"""


def test_restated_logf_against_the_c_library(tmp_path):
    """orbx_logf_glibc_f32 (MapPoint::PredictScale's log(ratio) on a float = logf; orbm_search.hip) against the C library on a sample;
    tools/trig/logf_count.c tries all 2,139,095,039 positive floats (0 mismatches on glibc 2.35, FMA and generic builds)."""
    c = tmp_path / "l.c"
    c.write_text(SRC_LOG)
    exe = str(tmp_path / "l")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-I", os.path.join(ROOT, "orb_slam2_e_amd", "csrc"), str(c), "-o", exe, "-lm"])
    n, bad, differ = (int(v) for v in subprocess.check_output([exe], text=True).split())
    assert n > 3.0e7
    assert bad == 0, "orbx_logf_glibc_f32 differs from the C library's logf (glibc >= 2.27 expected)"
    assert 0 < differ < n // 100         # logf is not correctly rounded everywhere (0.6 % of the floats in [0.5, 8)): the restatement is needed
