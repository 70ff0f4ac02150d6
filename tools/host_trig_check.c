// Evidence for the device's restatements of the host C library's float trigonometry
// (raytracer_amd/csrc/rtx_device.hpp: hostSinf, hostCosf, hostTanf, hostAcosf, hostAtan2f): the same
// algorithms written in C and compared with this machine's libm, bit for bit, over every float or
// every few floats of their domains.  The routines follow glibc 2.35: sinf / cosf as in
// sysdeps/ieee754/flt-32/s_sinf.c, s_cosf.c, sincosf.h, sincosf_data.c (Arm's optimized routines;
// evaluated with fused multiply-add, as the variant selected on CPUs that have it), atanf /
// atan2f / acosf / the tangent kernel as in s_atanf.c, e_atan2f.c, e_acosf.c, k_tanf.c (fdlibm in
// float) and tanf's argument reduction as in e_rem_pio2f.c (the sine's reduction in double).
// Build and run (about two minutes on one core):
//   gcc -O2 -ffp-contract=off -mfma -o /tmp/host_trig_check tools/host_trig_check.c -lm && /tmp/host_trig_check
// Prints the number of arguments tried and of mismatches per routine; all mismatch counts are 0
// against glibc 2.35 on x86-64.
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
static inline uint32_t fbits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float fromBits(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static float my_atanf(float x)
{
    static const float atanhi[] = { 4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f };
    static const float atanlo[] = { 5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f };
    static const float aT[] = { 3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f, 9.0908870101e-02f, -7.6918758452e-02f,
        6.6610731184e-02f, -5.8335702866e-02f, 4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f };
    const int32_t hx = (int32_t)fbits(x), ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x4c000000) {
        if (ix > 0x7f800000) return x + x;
        return hx > 0 ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3ee00000) {
        if (ix < 0x31000000) return x;
        id = -1;
    } else {
        x = fabsf(x);
        if (ix < 0x3f980000) {
            if (ix < 0x3f300000) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); }
            else { id = 1; x = (x - 1.0f) / (x + 1.0f); }
        } else {
            if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); }
            else { id = 3; x = -1.0f / x; }
        }
    }
    float z = x * x;
    const float w = z * z;
    const float s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    const float s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) return x - x * (s1 + s2);
    z = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return hx < 0 ? -z : z;
}

static float my_atan2f(float y, float x)
{
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const int32_t hx = (int32_t)fbits(x), ix = hx & 0x7fffffff, hy = (int32_t)fbits(y), iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return my_atanf(y);
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) { switch (m) { case 0: case 1: return y; case 2: return pi + tiny; default: return -pi - tiny; } }
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) { switch (m) { case 0: return pi_o_4 + tiny; case 1: return -pi_o_4 - tiny; case 2: return 3.0f * pi_o_4 + tiny; default: return -3.0f * pi_o_4 - tiny; } }
        switch (m) { case 0: return 0.0f; case 1: return -0.0f; case 2: return pi + tiny; default: return -pi - tiny; }
    }
    if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int32_t k = (iy - ix) >> 23;
    float z;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = my_atanf(fabsf(y / x));
    switch (m) {
    case 0: return z;
    case 1: return fromBits(fbits(z) ^ 0x80000000u);
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
    }
}

static float my_acosf(float x)
{
    const float pi = 3.1415925026e+00f, pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f;
    const float pS0 = 1.6666667163e-01f, pS1 = -3.2556581497e-01f, pS2 = 2.0121252537e-01f, pS3 = -4.0055535734e-02f, pS4 = 7.9153501429e-04f, pS5 = 3.4793309169e-05f;
    const float qS1 = -2.4033949375e+00f, qS2 = 2.0209457874e+00f, qS3 = -6.8828397989e-01f, qS4 = 7.7038154006e-02f;
    const int32_t hx = (int32_t)fbits(x), ix = hx & 0x7fffffff;
    if (ix == 0x3f800000) return hx > 0 ? 0.0f : pi + 2.0f * pio2_lo;
    if (ix > 0x3f800000) return (x - x) / (x - x);
    if (ix < 0x3f000000) {
        if (ix <= 0x23000000) return pio2_hi + pio2_lo;
        const float z = x * x;
        const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const float q = 1.0f + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const float r = p / q;
        return pio2_hi - (x - (pio2_lo - x * r));
    }
    if (hx < 0) {
        const float z = (1.0f + x) * 0.5f;
        const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const float q = 1.0f + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const float s = sqrtf(z);
        const float r = p / q;
        const float w = r * s - pio2_lo;
        return pi - 2.0f * (s + w);
    }
    const float z = (1.0f - x) * 0.5f;
    const float s = sqrtf(z);
    const float df = fromBits(fbits(s) & 0xfffff000u);
    const float c = (z - df * df) / (s + df);
    const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const float q = 1.0f + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const float r = p / q;
    const float w = r * s + c;
    return 2.0f * (df + w);
}

static float my_kernel_tanf(float x, float y, int iy)
{
    static const float T[] = { 3.3333334327e-01f, 1.3333334029e-01f, 5.3968254477e-02f, 2.1869488060e-02f, 8.8632395491e-03f, 3.5920790397e-03f, 1.4562094584e-03f,
        5.8804126456e-04f, 2.4646313977e-04f, 7.8179444245e-05f, 7.1407252108e-05f, -1.8558637748e-05f, 2.5907305826e-05f };
    const float pio4 = 7.8539812565e-01f, pio4lo = 3.7748947079e-08f;
    const int32_t hx = (int32_t)fbits(x), ix = hx & 0x7fffffff;
    float z, r, v, w, s;
    if (ix < 0x39000000) {
        if ((int)x == 0) {
            if ((ix | (iy + 1)) == 0) return 1.0f / fabsf(x);
            else if (iy == 1) return x;
            else return -1.0f / x;
        }
    }
    if (ix >= 0x3f2ca140) {
        if (hx < 0) { x = -x; y = -y; }
        z = pio4 - x;
        w = pio4lo - y;
        x = z + w;
        y = 0.0f;
        if (fabsf(x) < 0x1p-13f) return (1 - ((hx >> 30) & 2)) * iy * (1.0f - 2 * iy * x);
    }
    z = x * x;
    w = z * z;
    r = T[1] + w * (T[3] + w * (T[5] + w * (T[7] + w * (T[9] + w * T[11]))));
    v = z * (T[2] + w * (T[4] + w * (T[6] + w * (T[8] + w * (T[10] + w * T[12])))));
    s = z * x;
    r = y + z * (s * (r + v) + y);
    r += T[0] * s;
    w = x + r;
    if (ix >= 0x3f2ca140) {
        v = (float)iy;
        return (float)(1 - ((hx >> 30) & 2)) * (v - 2.0f * (x - (w * w / (w + v) - r)));
    }
    if (iy == 1) return w;
    z = fromBits(fbits(w) & 0xfffff000u);
    v = r - (z - x);
    const float a = -1.0f / w;
    const float t = fromBits(fbits(a) & 0xfffff000u);
    s = 1.0f + t * z;
    return t + a * (s + t * v);
}
// quarter turns in |x| < 120 and the remainder as head + tail (reduction in double); -1000 beyond
static int my_rem_pio2f(float x, float* y)
{
    if (((fbits(x) >> 20) & 0x7ff) >= ((fbits(120.0f) >> 20) & 0x7ff)) return -1000;
    double dx = x;
    const double r = dx * 0x1.45F306DC9C883p+23;
    const int n = ((int32_t)r + 0x800000) >> 24;
    dx = dx - n * 0x1.921FB54442D18p0;
    y[0] = (float)dx;
    y[1] = (float)(dx - y[0]);
    return n;
}
static float my_tanf(float x)
{
    const int32_t ix = (int32_t)fbits(x) & 0x7fffffff;
    if (ix <= 0x3f490fda) return my_kernel_tanf(x, 0.0f, 1);
    if (((fbits(x) >> 20) & 0x7ff) >= ((fbits(120.0f) >> 20) & 0x7ff)) return tanf(x);
    float y[2];
    const int n = my_rem_pio2f(x, y);
    return my_kernel_tanf(y[0], y[1], 1 - ((n & 1) << 1));
}

typedef struct { double sign[4]; double hpi_inv, hpi, c0, c1, c2, c3, c4, s1, s2, s3; } sincos_t;
static const sincos_t T[2] = {
  { { 1.0, -1.0, -1.0, 1.0 }, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, 0x1p0, -0x1.ffffffd0c621cp-2, 0x1.55553e1068f19p-5, -0x1.6c087e89a359dp-10, 0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13 },
  { { 1.0, -1.0, -1.0, 1.0 }, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, -0x1p0, 0x1.ffffffd0c621cp-2, -0x1.55553e1068f19p-5, 0x1.6c087e89a359dp-10, -0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13 } };
static inline uint32_t abstop12(float x) { return (fbits(x) >> 20) & 0x7ff; }
static inline float poly(double x, double x2, const sincos_t* p, int n) {
  if ((n & 1) == 0) { double x3 = x * x2; double s1 = fma(x2, p->s3, p->s2); double x7 = x3 * x2; double s = fma(x3, p->s1, x); return (float)fma(x7, s1, s); }
  double x4 = x2 * x2; double c2 = fma(x2, p->c4, p->c3); double c1 = fma(x2, p->c1, p->c0); double x6 = x4 * x2; double c = fma(x4, p->c2, c1); return (float)fma(x6, c2, c);
}
static inline double reduce_fast(double x, const sincos_t* p, int* np) { double r = x * p->hpi_inv; int n = ((int32_t)r + 0x800000) >> 24; *np = n; return fma(-(double)n, p->hpi, x); }
float my_sinf(float y) {
  double x = y; const sincos_t* p = &T[0]; int n;
  if (abstop12(y) < abstop12(0x1.921FB6p-1f)) { double s = x * x; if (abstop12(y) < abstop12(0x1p-12f)) return y; return poly(x, s, p, 0); }
  if (abstop12(y) < abstop12(120.0f)) { x = reduce_fast(x, p, &n); double s = p->sign[n & 3]; if (n & 2) p = &T[1]; return poly(x * s, x * x, p, n); }
  return sinf(y);
}
float my_cosf(float y) {
  double x = y; const sincos_t* p = &T[0]; int n;
  if (abstop12(y) < abstop12(0x1.921FB6p-1f)) { double x2 = x * x; if (abstop12(y) < abstop12(0x1p-12f)) return 1.0f; return poly(x, x2, p, 1); }
  if (abstop12(y) < abstop12(120.0f)) { x = reduce_fast(x, p, &n); double s = p->sign[n & 3]; if (n & 2) p = &T[1]; return poly(x * s, x * x, p, n ^ 1); }
  return cosf(y);
}

static unsigned long long tried, wrong;
static void check(const char* what, float x, float y, float want, float got)
{
    tried++;
    if (fbits(want) != fbits(got) && !(want != want && got != got)) {
        if (wrong < 5) printf("%s(%a, %a): libm %a, restated %a\n", what, x, y, want, got);
        wrong++;
    }
}
static void report(const char* what) { printf("%s: %llu arguments, %llu mismatches\n", what, tried, wrong); tried = wrong = 0; }
int main(void)
{
    for (uint32_t u = 0; u < 0x42f00000u; u += 3) // every third float below 120, both signs
        for (int sg = 0; sg < 2; sg++) { const float x = fromBits(u | (sg ? 0x80000000u : 0)); check("sinf", x, 0, sinf(x), my_sinf(x)); }
    report("sinf");
    for (uint32_t u = 0; u < 0x42f00000u; u += 3)
        for (int sg = 0; sg < 2; sg++) { const float x = fromBits(u | (sg ? 0x80000000u : 0)); check("cosf", x, 0, cosf(x), my_cosf(x)); }
    report("cosf");
    for (uint32_t u = 0; u <= 0x42efffffu; u++) // every float below 120
        for (int sg = 0; sg < 2; sg++) { const float x = fromBits(u | (sg ? 0x80000000u : 0)); check("tanf", x, 0, tanf(x), my_tanf(x)); }
    report("tanf");
    for (uint32_t u = 0; u <= 0x3f800000u; u += 2) // every second float of [-1, 1]
        for (int sg = 0; sg < 2; sg++) { const float x = fromBits(u | (sg ? 0x80000000u : 0)); check("acosf", x, 0, acosf(x), my_acosf(x)); }
    check("acosf", 1.0f, 0, acosf(1.0f), my_acosf(1.0f));
    check("acosf", -1.0f, 0, acosf(-1.0f), my_acosf(-1.0f));
    report("acosf");
    for (uint32_t u = 0; u < 0x7f800000u; u += 3)
        for (int sg = 0; sg < 2; sg++) { const float x = fromBits(u | (sg ? 0x80000000u : 0)); check("atanf", x, 0, atanf(x), my_atanf(x)); }
    report("atanf");
    uint64_t s = 88172645463325252ull; // xorshift: random bit patterns, and components of directions
    for (long i = 0; i < 400000000L; i++) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        float y = fromBits((uint32_t)s), x = fromBits((uint32_t)(s >> 32));
        if (i & 1) { y = (float)((int32_t)(uint32_t)s) * 0x1p-31f; x = (float)((int32_t)(uint32_t)(s >> 32)) * 0x1p-31f; }
        check("atan2f", x, y, atan2f(y, x), my_atan2f(y, x));
    }
    const float sp[] = { 0.0f, -0.0f, 1.0f, -1.0f, INFINITY, -INFINITY, 1e-40f, -1e-40f, 3e38f, -3e38f, 1e-30f };
    for (int i = 0; i < 11; i++) for (int j = 0; j < 11; j++) check("atan2f", sp[j], sp[i], atan2f(sp[i], sp[j]), my_atan2f(sp[i], sp[j]));
    report("atan2f");
    return 0;
}
