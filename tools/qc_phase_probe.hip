// Experiment: where the two co-resident workgroups of the fused half-row kernel are in their iteration loops.
// Wave 0 of every workgroup stamps s_memtime after every barrier (k_qc2 built with -DQC_STAMPS=<n>); the host groups the
// workgroups by (XCC, SE, CU) and prints, per CU, the start offsets and barrier phases of workgroups that ran together.
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -DQC_STAMPS=128 -I include tools/qc_phase_probe.hip -o build/qc_phase_probe
#include "../cuda_ldpc_amd/csrc/bldpc_qc_kernel.hpp"

#include <cmath>
#include <map>
#include <random>
#include <tuple>

using namespace cldpc;

int main(int argc, char **argv)
{
    // usage: qc_phase_probe [F [stagger [J L Z path]]]
    const int J = argc > 3 ? atoi(argv[3]) : 4, L = argc > 4 ? atoi(argv[4]) : 24, Z = argc > 5 ? atoi(argv[5]) : 96;
    const int F = argc > 1 ? atoi(argv[1]) : 65536, iters = 50;
    g_qc_stagger = argc > 2 ? atoi(argv[2]) : 0;
    std::vector<int> H(J * L);
    FILE *fp = fopen(argc > 6 ? argv[6] : "data/bldpc/J4_L24_Z96_BlockH.txt", "r");
    if (!fp) return 1;
    for (int &h : H)
        if (fscanf(fp, "%d", &h) != 1) return 1;
    fclose(fp);
    QcPlan q;
    if (qc_plan_build(&q, J, L, Z, H.data()) || !q.frames_per_wg) { printf("no plan: %s\n", err_buf()); return 1; }
    printf("kernel %s, F = %d, stagger %d\n", q.name, F, g_qc_stagger);
    const int N = L * Z, nWG = F / 2;
    std::vector<float> y((size_t)N * F);
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(1.0f, 0.7f);
    for (float &v : y) v = nd(rng);
    float *dy, *dyg; int *dD; unsigned *dbits;
    (void)hipMalloc(&dy, y.size() * 4); (void)hipMalloc(&dyg, y.size() * 4); (void)hipMalloc(&dD, (size_t)(N + 1) * F * 4); (void)hipMalloc(&dbits, (size_t)F * N / 8);
    (void)hipMalloc(&g_qc_stamps, (size_t)nWG * QC_STAMPS * 8);
    (void)hipMemset(g_qc_stamps, 0, (size_t)nWG * QC_STAMPS * 8);
    (void)hipMemcpy(dy, y.data(), y.size() * 4, hipMemcpyHostToDevice);
    qc_regroup(&q, dy, dyg, F, 0);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    std::vector<float> tms;
    for (int rep = 0; rep < 12; rep++) {
        qc_launch(&q, dyg, F, iters, N - J * Z, dD, nullptr, nullptr, dbits, 0, e0, e1);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 6) tms.push_back(ms);
    }
    std::sort(tms.begin(), tms.end());
    printf("launches 7-12: median %.3f ms = %.2f M codewords/s (min %.3f ms)\n", tms[tms.size() / 2], F / tms[tms.size() / 2] * 1e-3, tms[0]);
    std::vector<unsigned long long> st((size_t)nWG * QC_STAMPS);
    (void)hipMemcpy(st.data(), g_qc_stamps, st.size() * 8, hipMemcpyDeviceToHost);
    // group by CU
    std::map<std::tuple<int, int, int, int>, std::vector<int>> cus;
    for (int w = 0; w < nWG; w++) {
        const unsigned hw = (unsigned)st[(size_t)w * QC_STAMPS], xcc = (unsigned)(st[(size_t)w * QC_STAMPS] >> 32);
        cus[{(int)(xcc & 15), (int)((hw >> 13) & 7), (int)((hw >> 12) & 1), (int)((hw >> 8) & 15)}].push_back(w);
    }
    printf("%zu distinct (xcc, se, sh, cu)\n", cus.size());
    // per workgroup: start, end, mean iteration period, split VN-barrier -> CN-barrier
    double sum_iter = 0, sum_cn = 0; long cnt = 0;
    for (int w = 0; w < nWG; w++) {
        const unsigned long long *s = &st[(size_t)w * QC_STAMPS];
        // s[1] start, s[2] after prologue barrier, then pairs (after VN barrier, after CN barrier) for it = 1..49
        for (int it = 5; it < 45; it++) { sum_iter += (double)(s[3 + 2 * it + 2] - s[3 + 2 * it]); sum_cn += (double)(s[3 + 2 * it + 1] - s[3 + 2 * it]); cnt++; }
    }
    printf("mean iteration period %.0f cycles, of which VN-barrier -> CN-barrier %.0f\n", sum_iter / cnt, sum_cn / cnt);
    {
        double pro = 0, loop = 0, epi = 0, first = 0;
        std::vector<double> per(49, 0.0);
        for (int w = 0; w < nWG; w++) {
            const unsigned long long *s = &st[(size_t)w * QC_STAMPS];
            pro += (double)(s[2] - s[1]); first += (double)(s[3] - s[2]); loop += (double)(s[100] - s[2]); epi += (double)(s[101] - s[100]);
            for (int it = 1; it < 49; it++) per[it] += (double)(s[3 + 2 * it] - s[3 + 2 * (it - 1)]);
        }
        printf("per workgroup: prologue %.0f cycles, first VN phase %.0f, loop (49 iterations) %.0f, final VN + outputs %.0f\n", pro / nWG, first / nWG, loop / nWG, epi / nWG);
        {
            double vnw = 0, cnw = 0; int n = 0; // wave 0's own arrival at the barriers of iteration 20 (k_qc only: slots 120, 121)
            for (int w = 0; w < nWG; w++) {
                const unsigned long long *s = &st[(size_t)w * QC_STAMPS];
                if (!s[120]) continue;
                vnw += (double)(s[3 + 2 * 19] - s[120]); cnw += (double)(s[4 + 2 * 19] - s[121]); n++;
            }
            if (n) printf("iteration 20: wave 0 waits %.0f cycles at the VN barrier, %.0f at the CN barrier\n", vnw / n, cnw / n);
        }
        printf("iteration period by iteration:");
        for (int it = 1; it < 49; it++) printf(" %.0f", per[it] / nWG);
        printf("\n");
    }
    // phase offset between workgroups that overlap in time on one CU: for each CU sort by start; for consecutive co-resident pairs
    // report (start difference) and the phase of B's VN barrier inside A's iteration at A's iteration 20
    std::vector<double> phase;
    int shown = 0;
    for (auto &kv : cus) {
        auto &v = kv.second;
        std::sort(v.begin(), v.end(), [&](int x, int y) { return st[(size_t)x * QC_STAMPS + 1] < st[(size_t)y * QC_STAMPS + 1]; });
        for (size_t i = 0; i + 1 < v.size(); i++) {
            const unsigned long long *A = &st[(size_t)v[i] * QC_STAMPS], *B = &st[(size_t)v[i + 1] * QC_STAMPS];
            const unsigned long long a20 = A[3 + 40], a21 = A[3 + 42]; // A's VN barriers of iterations 21 and 22
            if (B[1] > a20 || B[3 + 2 * 48] < a21) continue;             // B not running then
            for (int it = 0; it < 49; it++)
                if (B[3 + 2 * it] >= a20 && B[3 + 2 * it] < a21) { phase.push_back((double)(B[3 + 2 * it] - a20) / (double)(a21 - a20)); break; }
            if (shown < 12) {
                printf("cu(%d,%d,%d,%d) tg %u/%u start diff %lld cycles, A period %llu\n", std::get<0>(kv.first), std::get<1>(kv.first), std::get<2>(kv.first), std::get<3>(kv.first),
                       (unsigned)(A[0] >> 16) & 15, (unsigned)(B[0] >> 16) & 15, (long long)(B[1] - A[1]), a21 - a20);
                shown++;
            }
        }
    }
    int hist[10] = {};
    for (double p : phase) hist[std::min(9, (int)(p * 10))]++;
    printf("phase of the co-resident workgroup's VN barrier inside this workgroup's iteration (10 bins, %zu pairs):", phase.size());
    for (int h : hist) printf(" %d", h);
    printf("\n");
    return 0;
}
