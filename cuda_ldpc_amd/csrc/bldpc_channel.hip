// bldpc_channel.hip -- host-side input generator of the binary sweep.
//
// Restates AWGNChannel_CPU / RandomModule (bldpc_实习/LDPC_Encoder.cu:25-56) and the sigma formula
// of main.cu:120-127 so that a sweep driven through this library sees bit-identical channel
// samples.  The reference is C++: log/sqrt on float arguments are the float overloads, the
// sin argument is double (PI is the double literal 3.1415926, define.cuh:58).
#include <algorithm>
#include <cmath>
#include <thread>
#include <vector>

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

namespace {
constexpr unsigned kA[3] = {249u, 251u, 252u}, kM[3] = {61967u, 63443u, 63599u}; // LDPC_Encoder.cu:48-50

// a^k mod m for the three LCG moduli.  They are prime (61967, 63443, 63599), so a^(m-1) = 1 and the exponent reduces to
// k mod (m-1) < 2^16; operands stay below 2^16, so every product fits 32 bits: at most 16 squarings of 32-bit arithmetic
// per jump instead of up to 40 of 64-bit arithmetic.
__host__ __device__ inline unsigned powmod(unsigned a, unsigned long long k, unsigned m)
{
    unsigned e = (unsigned)(k % (unsigned long long)(m - 1));
    unsigned r = 1, b = a % m;
    while (e) {
        if (e & 1) r = (r * b) % m;
        b = (b * b) % m;
        e >>= 1;
    }
    return r;
}

// frames [f0, f1) of the serial loop (LDPC_Encoder.cu:30-38: frame outer, bit inner), seed = the stream state before frame f0
void awgn_frames(int *seed, float sigma, float *out, const int *cw, int N, int F, int f0, int f1)
{
    const double two_pi = 2 * 3.1415926;
    for (int f = f0; f < f1; f++)
        for (int n = 0; n < N; n++) {
            const float u1 = random_module(seed);
            const float u2 = random_module(seed);
            const float amp = std::sqrt(-2.0f * std::log(1.0f - u1));
            const int c = cw ? cw[(size_t)n * F + f] : 0;
            out[(size_t)n * F + f] = (float)((double)sigma * std::sin(two_pi * (double)u2) * (double)amp + 1.0 - (double)(2 * c));
        }
}
} // namespace

// The reference's generator is ONE serial stream; an LCG can be advanced k steps at once (seed * a^k mod m), so the frames are
// cut into contiguous ranges, one per host thread, each of which starts from the stream state the serial loop would have at its
// first frame: the same draws, the same host libm, the same bits as the serial loop (the tests compare with a serial CPU
// generator), in 1/threads of the time.
extern "C" int bldpc_awgn_channel_host(int seed[3], float sigma, float *out, const int *cw, int N, int F)
{
    if (!seed || !out || N <= 0 || F <= 0) return cldpc::fail(BLDPC_EINVAL, "bldpc_awgn_channel_host: bad argument");
    bool jumpable = true; // the jump needs canonical states; any other int triple takes the serial loop as the reference would
    for (int i = 0; i < 3; i++) jumpable = jumpable && seed[i] >= 0 && (unsigned)seed[i] < kM[i];
    const unsigned hw = std::thread::hardware_concurrency();
    const int T = (int)std::min<long long>({(long long)(hw ? hw : 1), 16LL, (long long)F / 16, (long long)((size_t)N * F / 65536)});
    if (!jumpable || T < 2) {
        awgn_frames(seed, sigma, out, cw, N, F, 0, F);
        return BLDPC_OK;
    }
    const int s0[3] = {seed[0], seed[1], seed[2]};
    auto jumped = [&](int f, int *s) {
        const unsigned long long k = 2ull * (unsigned long long)N * f;
        for (int i = 0; i < 3; i++) s[i] = (int)(((unsigned long long)s0[i] * powmod(kA[i], k, kM[i])) % kM[i]);
    };
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++) {
        const int f0 = (int)((long long)F * t / T), f1 = (int)((long long)F * (t + 1) / T);
        th.emplace_back([=]() {
            int s[3];
            jumped(f0, s);
            awgn_frames(s, sigma, out, cw, N, F, f0, f1);
        });
    }
    for (auto &x : th) x.join();
    jumped(F, seed);
    return BLDPC_OK;
}

namespace {
// One thread per (frame f, run of kRun consecutive bits): jump the three LCGs to the first draw of the run
// (draw index 2*(f*N + n0)), then step them as RandomModule does.  threadIdx.x runs along f: stores coalesce.
constexpr int kRun = 32;
__global__ __launch_bounds__(256) void k_awgn(unsigned s0, unsigned s1, unsigned s2, float sigma, float *out, const int *cw, int N, int F)
{
    const int f = blockIdx.x * 256 + threadIdx.x;
    const int n0 = blockIdx.y * kRun;
    if (f >= F) return;
    const unsigned long long k = 2ull * ((unsigned long long)f * N + n0);
    unsigned s[3] = {s0, s1, s2};
#pragma unroll
    for (int i = 0; i < 3; i++) s[i] = (unsigned)(((unsigned long long)s[i] * powmod(kA[i], k, kM[i])) % kM[i]);
    const double two_pi = 2 * 3.1415926;
    for (int n = n0; n < min(N, n0 + kRun); n++) {
        float u[2];
#pragma unroll
        for (int d = 0; d < 2; d++) {
#pragma unroll
            for (int i = 0; i < 3; i++) s[i] = (s[i] * kA[i]) % kM[i];
            // x / m for an integer 0 <= x < m, m an odd prime below 2^16: the correctly rounded float quotient equals the double
            // product x * (1/m) rounded to float (x/m is at least 2^-40 away, relatively, from every float rounding boundary;
            // all 3 x 63 599 cases checked in tests/test_host_cpu.py) -- three conversions and a multiply instead of a division
            float t = (float)((double)(int)s[0] * (1.0 / 61967.0)) + (float)((double)(int)s[1] * (1.0 / 63443.0)) + (float)((double)(int)s[2] * (1.0 / 63599.0));
            t -= (int)t;
            u[d] = t;
        }
        const float amp = sqrtf(-2.0f * logf(1.0f - u[0]));
        const int c = cw ? cw[(size_t)n * F + f] : 0;
        out[(size_t)n * F + f] = (float)((double)sigma * sin(two_pi * (double)u[1]) * (double)amp + 1.0 - (double)(2 * c));
    }
}
} // namespace

extern "C" int bldpc_awgn_channel_device(int seed[3], float sigma, float *out, const int *cw, int N, int F, void *stream)
{
    if (!seed || !out || N <= 0 || F <= 0) return cldpc::fail(BLDPC_EINVAL, "bldpc_awgn_channel_device: bad argument");
    for (int i = 0; i < 3; i++)
        if (seed[i] < 0 || (unsigned)seed[i] >= kM[i]) return cldpc::fail(BLDPC_EINVAL, "seed[%d]=%d outside [0,%u)", i, seed[i], kM[i]);
    hipLaunchKernelGGL(k_awgn, dim3((unsigned)((F + 255) / 256), (unsigned)((N + kRun - 1) / kRun)), dim3(256), 0, (hipStream_t)stream,
                       (unsigned)seed[0], (unsigned)seed[1], (unsigned)seed[2], sigma, out, cw, N, F);
    CLDPC_HIP(hipGetLastError(), BLDPC_EHIP);
    const unsigned long long draws = 2ull * (unsigned long long)N * F;
    for (int i = 0; i < 3; i++) seed[i] = (int)(((unsigned long long)seed[i] * powmod(kA[i], draws, kM[i])) % kM[i]);
    return BLDPC_OK;
}

extern "C" float bldpc_sigma(float snr, int snrtype, float rate)
{
    if (snrtype == 0) return (float)std::sqrt(0.5 / (rate * std::pow(10.0, (double)(snr / 10.0))));
    return (float)std::sqrt(0.5 / std::pow(10.0, (double)(snr / 10.0)));
}
