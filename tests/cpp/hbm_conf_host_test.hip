// hbm_conf_host_test.hip -- host-side check of nb_hbm_conf (the explicit-stack form of the reference's ConstructConf,
// myNBLDPC/src/LDPC_Decoder.cpp:319-359) against a plain recursion written the way the reference's is, on random sorted
// messages: every bit of the max array must agree for row weights 1 ... 22, Nm in {1, 2, 3, q}, Nc in {0 ... 3}, q in {4, 16, 256}.
// Runs on the CPU (no kernel launch): the function is __host__ __device__.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../cuda_ldpc_amd/csrc/nbldpc_hbm_kernel.hpp"

struct Rec {
    const float *pairs; const int *src; int e, end, q2, Nm, Nc; float *E;
    int sym; float s; int diff;
    void go(int begin)
    {
        if (begin > end) { if (s > E[sym >> 2]) E[sym >> 2] = s; return; } // symbols travel premultiplied by 4 (byte offsets into the max array)
        if (begin == e) { go(begin + 1); return; }
        for (int k = 0; k < Nm; k++) {
            const float v = pairs[(size_t)src[begin] * q2 + 2 * k];
            int m; memcpy(&m, &pairs[(size_t)src[begin] * q2 + 2 * k + 1], 4);
            sym ^= m; s = s + v; diff += (k != 0) ? 1 : 0;
            if (diff <= Nc) { go(begin + 1); sym ^= m; s = s - v; diff -= (k != 0) ? 1 : 0; }
            else { sym ^= m; s = s - v; diff -= (k != 0) ? 1 : 0; break; }
        }
    }
};

int main()
{
    unsigned long long st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (unsigned)(st >> 32); };
    long cases = 0;
    for (int q : {4, 16, 256})
        for (int w = 1; w <= 22; w += (w < 6 ? 1 : 5))
            for (int Nm : {1, 2, 3, q})
                for (int Nc = 0; Nc <= 3; Nc++) {
                    if (Nm == q && Nc > 1) continue;           // conf(q, 1) is the only full-width walk the decoder makes
                    if (w > 12 && Nm == 3 && Nc == 3) continue; // 3^Nc * C(21, 3) leaves: slow and nothing new
                    const int q2 = 2 * q;
                    std::vector<float> pairs((size_t)w * q2);
                    std::vector<int> src(w);
                    for (int i = 0; i < w; i++) {
                        src[i] = (i * 7 + 3) % w; // a permutation of the edge slots when gcd(7, w) == 1, a valid index anyway
                        float v = (float)(rnd() % 2000) * 0.37f;
                        for (int k = 0; k < q; k++) { // descending values, arbitrary symbols
                            pairs[(size_t)i * q2 + 2 * k] = v;
                            int m = (int)(rnd() % q) << 2;
                            memcpy(&pairs[(size_t)i * q2 + 2 * k + 1], &m, 4);
                            v -= (float)(rnd() % 1000) * 0.0131f;
                        }
                    }
                    for (int e = 0; e < w; e += (w > 8 ? 5 : 1)) {
                        std::vector<float> Ea(q, -__builtin_inff()), Eb(q, -__builtin_inff());
                        cldpc::nb_hbm_conf(pairs.data(), src.data(), e, w - 1, q2, Ea.data(), Nm, Nc);
                        Rec r{pairs.data(), src.data(), e, w - 1, q2, Nm, Nc, Eb.data(), 0, 0.0f, 0};
                        r.go(0);
                        if (memcmp(Ea.data(), Eb.data(), q * sizeof(float)) != 0) {
                            printf("MISMATCH q=%d w=%d Nm=%d Nc=%d e=%d\n", q, w, Nm, Nc, e);
                            return 1;
                        }
                        cases++;
                    }
                }
    printf("OK %ld cases\n", cases);
    return 0;
}
