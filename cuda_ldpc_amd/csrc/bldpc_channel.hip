// bldpc_channel.hip -- host-side input generator of the binary sweep.
//
// Restates AWGNChannel_CPU / RandomModule (bldpc_实习/LDPC_Encoder.cu:25-56) and the sigma formula
// of main.cu:120-127 so that a sweep driven through this library sees bit-identical channel
// samples.  The reference is C++: log/sqrt on float arguments are the float overloads, the
// sin argument is double (PI is the double literal 3.1415926, define.cuh:58).
#include <cmath>

#include "../../include/bldpc.h"
#include "common.hpp"

namespace {
inline float random_module(int *seed) // LDPC_Encoder.cu:45-56
{
    seed[0] = (seed[0] * 249) % 61967;
    seed[1] = (seed[1] * 251) % 63443;
    seed[2] = (seed[2] * 252) % 63599;
    float t = ((float)seed[0] / 61967.0f) + ((float)seed[1] / 63443.0f) + ((float)seed[2] / 63599.0f);
    t -= (int)t;
    return t;
}
} // namespace

extern "C" int bldpc_awgn_channel_host(int seed[3], float sigma, float *out, const int *cw, int N, int F)
{
    if (!seed || !out || N <= 0 || F <= 0) return cldpc::fail(BLDPC_EINVAL, "bldpc_awgn_channel_host: bad argument");
    const double two_pi = 2 * 3.1415926;
    for (int f = 0; f < F; f++)
        for (int n = 0; n < N; n++) {
            const float u1 = random_module(seed);
            const float u2 = random_module(seed);
            const float amp = std::sqrt(-2.0f * std::log(1.0f - u1));
            const int c = cw ? cw[(size_t)n * F + f] : 0;
            out[(size_t)n * F + f] = (float)((double)sigma * std::sin(two_pi * (double)u2) * (double)amp + 1.0 - (double)(2 * c));
        }
    return BLDPC_OK;
}

extern "C" float bldpc_sigma(float snr, int snrtype, float rate)
{
    if (snrtype == 0) return (float)std::sqrt(0.5 / (rate * std::pow(10.0, (double)(snr / 10.0))));
    return (float)std::sqrt(0.5 / std::pow(10.0, (double)(snr / 10.0)));
}
