/* trig_check -- (float)cos((double)a), (float)sin((double)a) of csrc/sincos_pio4.h against libm, for every `stride`-th
 * float a in [0, hi] (stride 1 = exhaustive).  Prints the number of values checked and of differences. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../monocular_slam_amd/csrc/sincos_pio4.h"

int main(int argc, char** argv)
{
    const uint32_t stride = argc > 1 ? (uint32_t)strtoul(argv[1], 0, 10) : 1;
    const float hi = 6.2831855f; /* fl(360 deg) in radians and a little more */
    uint32_t hib; memcpy(&hib, &hi, 4);
    unsigned long long n = 0, bad = 0;
    for (uint64_t b = 0; b <= hib; b += stride) {
        const uint32_t bb = (uint32_t)b;
        float a; memcpy(&a, &bb, 4);
        double s, c;
        mslam_sincos_2pi((double)a, &s, &c);
        const float fs = (float)s, fc = (float)c, gs = (float)sin((double)a), gc = (float)cos((double)a);
        if (memcmp(&fs, &gs, 4) || memcmp(&fc, &gc, 4)) {
            if (bad < 10) printf("diff at a=%.9g: sin %.9g vs %.9g, cos %.9g vs %.9g\n", a, fs, gs, fc, gc);
            bad++;
        }
        n++;
    }
    printf("checked %llu differences %llu\n", n, bad);
    return bad != 0;
}
