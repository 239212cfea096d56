// detmath.h -- sin / cos in double with an explicit, platform-independent operation order (gfx950 device code).
//
// The reference reaches std::sin(float) / std::cos(float) (SURF descriptor window rotation, upstream surf.cpp SURFInvoker,
// called from ImageUtility.py:262) and cos(double) / sin(double) (ORB rBRIEF rotation, upstream orb.cpp computeOrbDescriptors,
// ImageUtility.py:260,262).  A correctly rounded result does not depend on the library, but OCML's and glibc's own routines
// differ in the last ulp, and one ulp of sin/cos can flip the u8 rounding of a window sample.  So the engine and the oracle
// both evaluate THIS algorithm (the CPU checker carries its own copy, written from the same description):
//   k = rint(x * 2/pi);  r = (x - k * PIO2_HI) - k * PIO2_LO      (Cody-Waite; PIO2_HI has 33 significant bits, so for the
//                                                                  float-valued |x| <= 2^10 of this path the first product and
//                                                                  difference are exact)
//   sin / cos of r on [-pi/4, pi/4] by the fdlibm minimax polynomials (degree 13 / 14), Horner form, plain mul / add
//   quadrant selection by k & 3.
// Error < 1e-16 absolute, i.e. the float rounding of the result equals the correctly rounded sinf / cosf except when the true
// value lies within ~2^-29 ulp of a rounding boundary.  Every operation is an IEEE double mul / add / sub / rint, compiled with
// -ffp-contract=off on both sides, so device and oracle agree bit for bit.
#pragma once
#include <hip/hip_runtime.h>

__device__ __forceinline__ void det_sincos(double x, double *s_out, double *c_out)
{
    const double INV_PIO2 = 6.36619772367581382433e-01;
    const double PIO2_HI = 1.57079632673412561417e+00;     // first 33 bits of pi/2
    const double PIO2_LO = 6.07710050650619224932e-11;     // pi/2 - PIO2_HI
    const double kd = rint(x * INV_PIO2);
    const int k = (int)kd;
    const double r = (x - kd * PIO2_HI) - kd * PIO2_LO;
    const double z = r * r;
    // fdlibm __kernel_sin / __kernel_cos coefficients
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double ps = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    const double sr = r + (z * r) * (S1 + z * ps);
    const double pc = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    const double cr = 1.0 - (0.5 * z - z * pc);
    double s, c;
    switch (k & 3) {
    case 0: s = sr; c = cr; break;
    case 1: s = cr; c = -sr; break;
    case 2: s = -sr; c = -cr; break;
    default: s = -cr; c = sr; break;
    }
    *s_out = s; *c_out = c;
}
