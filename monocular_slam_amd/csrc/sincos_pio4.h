// sincos_pio4.h -- cos and sin of a double in [0, 2 pi] (an ORB orientation in radians) by quadrant reduction and the
// classic |r| <= pi/4 minimax kernels (degree 13 / 14), every operation an explicit IEEE add / mul / fma, so the CPU
// checker (tests/trig_check.c) and the gfx950 kernel execute the same arithmetic bit for bit.  The descriptor needs
// (float)cos((double)a) and (float)sin((double)a) for FLOAT a; tests/trig_check.c compares this routine with libm's
// cos/sin after the cast to float for EVERY float in [0, 6.2831855] (1.09e9 values): no difference.  Plain C.
#ifndef MSLAM_SINCOS_PIO4_H
#define MSLAM_SINCOS_PIO4_H
#if defined(__HIPCC__)
#define MSLAM_HD __device__ __forceinline__
#else
#include <math.h>
#define MSLAM_HD static inline
#endif

MSLAM_HD void mslam_sincos_2pi(double a, double* sn, double* cs)
{
    // quadrant q = nearest integer to a / (pi/2), 0..4;  r = a - q pi/2 in two steps (q pi/2_hi is exact: q < 8, 33-bit hi)
    const double q = (double)(int)(a * 0.63661977236758138 + 0.5);
    double r = fma(-q, 1.57079632673412561417e+00, a);        /* pi/2, leading 33 bits */
    r = fma(-q, 6.07710050650619224932e-11, r);               /* pi/2 - hi */
    const double z = r * r;
    // sin(r) = r + r z (S1 + z (S2 + ... )),  cos(r) = 1 - z/2 + z^2 (C1 + z (C2 + ...))
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(z, ps, 2.75573137070700676789e-06);
    ps = fma(z, ps, -1.98412698298579493134e-04);
    ps = fma(z, ps, 8.33333333332248946124e-03);
    ps = fma(z, ps, -1.66666666666666324348e-01);
    const double s = fma(r * z, ps, r);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(z, pc, -2.75573143513906633035e-07);
    pc = fma(z, pc, 2.48015872894767294178e-05);
    pc = fma(z, pc, -1.38888888888741095749e-03);
    pc = fma(z, pc, 4.16666666666666019037e-02);
    const double hz = 0.5 * z, w = 1.0 - hz;
    const double c = w + (((1.0 - w) - hz) + z * z * pc);
    const int qi = (int)q & 3;
    *sn = (qi == 0) ? s : (qi == 1) ? c : (qi == 2) ? -s : -c;
    *cs = (qi == 0) ? c : (qi == 1) ? -s : (qi == 2) ? -c : s;
}
#endif
