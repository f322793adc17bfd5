/* Exhaustive proof-by-enumeration for the EMS kernel's division shortcut (cuda_ldpc_amd/csrc/nbldpc_kernel.hpp, nb_div12):
 *   reference (myNBLDPC/src/LDPC_Decoder.cpp:309):  (float)((double)x / 1.2)
 *   kernel:  r = RN(1/1.2); q0 = x*r; e = fma(-1.2, q0, x); q1 = fma(e, r, q0); (float)q1      (Markstein's correction step)
 * for EVERY finite non-zero float x (zeros and infinities take x itself in the kernel; NaN is undefined input).
 *   gcc -O2 -fopenmp -ffp-contract=off tests/c/div12_exhaustive.c -lm -o build/div12 && build/div12 [stride]
 * Prints the number of mismatches (0 expected); exit code 1 on any mismatch. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char **argv)
{
    const uint64_t stride = argc > 1 ? strtoull(argv[1], 0, 10) : 1;
    const volatile double one = 1.0, d12 = 1.2;
    const double r = one / d12;
    uint64_t bad = 0, n = 0;
#pragma omp parallel for reduction(+ : bad, n) schedule(static)
    for (uint64_t u = 0; u < (1ull << 32); u += stride) {
        const uint32_t b = (uint32_t)u;
        if ((b & 0x7f800000u) == 0x7f800000u || (b & 0x7fffffffu) == 0) continue; /* inf / NaN / zero */
        float x;
        memcpy(&x, &b, 4);
        const float ref = (float)((double)x / d12);
        const double q0 = (double)x * r;
        const double e = fma(-d12, q0, (double)x);
        const float alt = (float)fma(e, r, q0);
        uint32_t a1, a2;
        memcpy(&a1, &ref, 4);
        memcpy(&a2, &alt, 4);
        bad += a1 != a2;
        n++;
    }
    printf("checked %llu floats, %llu mismatches\n", (unsigned long long)n, (unsigned long long)bad);
    return bad != 0;
}
