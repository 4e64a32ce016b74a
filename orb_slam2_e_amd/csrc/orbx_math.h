// orbx_math.h -- exact-arithmetic helpers shared by the HIP kernels.
//
// Every function here is a fixed sequence of IEEE-754 operations (no FMA
// contraction: the library is built with -ffp-contract=off), so the device
// result is a pure function of the inputs and can be checked on the host by
// compiling this header with g++ (tests/test_math_host.py does that).
#ifndef ORBX_MATH_H
#define ORBX_MATH_H

#if defined(__HIPCC__)
#define ORBX_HD __host__ __device__ __forceinline__
#else
#define ORBX_HD static inline
#endif

#include <math.h>

// cvRound(float): round-half-to-even.
ORBX_HD int orbx_cvround(float v) { return (int)rintf(v); }

// cv::fastAtan2(y, x) in degrees, OpenCV 3.4 scalar path;
// called by IC_Angle, src/ORBextractor.cc:103.
ORBX_HD float orbx_fast_atan2(float y, float x)
{
    const float s = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * s, p3 = -0.3258083974640975f * s;
    const float p5 = 0.1555786518463281f * s, p7 = -0.04432655554792128f * s;
    const float eps = 2.2204460492503131e-16f; // (float)DBL_EPSILON
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + eps);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + eps);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// sin/cos of a float angle in [0, 2*pi], evaluated in double with a fixed
// operation sequence (Cody-Waite reduction by pi/2 + degree-13/12 minimax
// kernels) and rounded to float.  Replaces `(float)cos(angle)`, `(float)sin(angle)`
// of computeOrbDescriptor (src/ORBextractor.cc:112-113).  The host test checks
// it against libm's correctly-rounded-in-practice (float)cos((double)x).
ORBX_HD void orbx_sincos_f32(float xf, float *sn, float *cs)
{
    const double x = (double)xf;
    const double fn = rint(x * 6.36619772367581382433e-01); // 2/pi
    const int n = (int)fn;
    const double r = (x - fn * 1.57079632673412561417e+00) - fn * 6.07710050650619224932e-11;
    const double z = r * r;
    const double ps = -1.66666666666666324348e-01 + z * (8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 +
                      z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10))));
    const double s = r + (r * z) * ps;
    const double pc = 4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                      z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
    const double c = 1.0 - (0.5 * z - (z * z) * pc);
    double so, co;
    switch (n & 3) {
    case 0: so = s; co = c; break;
    case 1: so = c; co = -s; break;
    case 2: so = -s; co = -c; break;
    default: so = -c; co = s; break;
    }
    *sn = (float)so;
    *cs = (float)co;
}

// cosf / sinf of a float angle in [0, 2*pi] as glibc >= 2.28 computes them (sysdeps/ieee754/flt-32/s_cosf.c, s_sinf.c: the ARM
// optimized-routines algorithm -- argument in double, quadrant by a scaled float-to-int conversion, degree-8 / degree-7 polynomials in
// double, one rounding to float).  This is what `cos(angle)` / `sin(angle)` of computeOrbDescriptor (src/ORBextractor.cc:112-113, float
// argument under `using namespace std`) call on a current Linux.  A fixed sequence of double multiplies and adds: tools/trig/
// trig_variant_count.c checks the same sequence bit for bit against the C library over every float in [0, 2 pi] (it also holds with
// every a + b c fused: the FMA build glibc selects on CPUs with FMA returns the same floats).
ORBX_HD float orbx_glibc_poly_f32(double x, double x2, int alt, int n)
{
    const double sg = alt ? -1.0 : 1.0;
    if ((n & 1) == 0) {
        const double x3 = x * x2;
        const double t1 = 0x1.1107605230bc4p-7 + x2 * -0x1.994eb3774cf24p-13;
        const double x7 = x3 * x2;
        const double s = x + x3 * -0x1.555545995a603p-3;
        return (float)(s + x7 * t1);
    }
    const double x4 = x2 * x2;
    const double t2 = sg * -0x1.6c087e89a359dp-10 + x2 * (sg * 0x1.99343027bf8c3p-16);
    const double t1 = sg * 0x1p0 + x2 * (sg * -0x1.ffffffd0c621cp-2);
    const double x6 = x4 * x2;
    const double c = t1 + x4 * (sg * 0x1.55553e1068f19p-5);
    return (float)(c + x6 * t2);
}
ORBX_HD void orbx_sincos_glibc_f32(float y, float *sn, float *cs)
{
    double x = (double)y;
    union { float f; unsigned u; } v;
    v.f = y;
    const unsigned top = (v.u >> 20) & 0x7ffu;
    if (top < 0x3f4u) {                                   // |y| < pi / 4 (abstop12(0x1.921FB6p-1f))
        const double x2 = x * x;
        if (top < 0x398u) { *sn = y; *cs = 1.0f; return; }   // |y| < 2^-12
        *sn = orbx_glibc_poly_f32(x, x2, 0, 0);
        *cs = orbx_glibc_poly_f32(x, x2, 0, 1);
        return;
    }
    const double r = x * 0x1.45F306DC9C883p+23;           // 2 / pi * 2^24
    const int n = ((int)r + 0x800000) >> 24;
    x = x - n * 0x1.921FB54442D18p0;
    const double s = ((n + 1) & 2) ? -1.0 : 1.0;          // sign[n & 3] = {1, -1, -1, 1}
    const int alt = (n & 2) != 0;
    *sn = orbx_glibc_poly_f32(x * s, x * x, alt, n);
    *cs = orbx_glibc_poly_f32(x * s, x * x, alt, n ^ 1);
}

// glibc >= 2.27's logf (sysdeps/ieee754/flt-32/e_logf.c, the table-driven double-precision evaluation: 16 intervals of the mantissa,
// z = x / 2^k, r = z invc - 1, log x = k ln 2 + logc + r + r^2 (A2 + A1 r + A0 r^2)), restated operation for operation.
// Why: MapPoint::PredictScale (MapPoint.cc:448-480) computes ceil(log(ratio) / mfLogScaleFactor) with a FLOAT ratio, which under the
// reference's headers is logf; logf is not correctly rounded everywhere, and where it differs from (float)log((double)x) the quotient
// can land on the other side of an integer.  tools/trig/logf_count.c tries every positive float against the C library.
ORBX_HD float orbx_logf_glibc_f32(float x)
{
    const double T[16][2] = {
        {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2}, {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},
        {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3}, {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},
        {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4}, {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5},
        {0x1p+0, 0x0p+0}, {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5}, {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
        {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3}, {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3}, {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},
        {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};
    const double A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2, Ln2 = 0x1.62e42fefa39efp-1;
    union { float f; unsigned u; } v;
    v.f = x;
    unsigned ix = v.u;
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {          // zero, subnormal, negative, inf, nan
        if (ix * 2u == 0u) return -1.0f / 0.0f;
        if (ix == 0x7f800000u) return x;
        if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return (x - x) / (x - x);   // nan (glibc: __math_invalidf)
        v.f = x * 0x1p23f;                                         // subnormal: normalise
        ix = v.u - (23u << 23);
    }
    const unsigned tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> (23 - 4)) % 16u);
    const int k = (int)tmp >> 23;
    v.u = ix - (tmp & (0x1ffu << 23));
    const double invc = T[i][0], logc = T[i][1];
    const double z = (double)v.f;
    const double r = z * invc - 1;
    const double y0 = logc + (double)k * Ln2;
    const double r2 = r * r;
    double y = A1 * r + A2;
    y = A0 * r2 + y;
    y = y * r2 + (y0 + r);
    return (float)y;
}

#endif // ORBX_MATH_H
