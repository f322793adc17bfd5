// qc_local_assign_host_test.cpp -- host-side check of qc2_local_assign (csrc/bldpc_qc_assign.hpp): the matching that hands every
// block column to one block row containing it, L / J columns per row, for the fused kernels' local edges.  Random block patterns
// built around a hidden valid assignment (so one exists) plus random extra blocks: the function must find a valid one; patterns
// with a row too light to take its share, or with L not a multiple of J, must be refused.  Runs on the CPU (no kernel launch).
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../cuda_ldpc_amd/csrc/bldpc_qc_assign.hpp"

static unsigned long long st = 88172645463325252ull;
static unsigned rnd() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (unsigned)(st >> 11); }

static void lists(int J, int L, const std::vector<int> &H, std::vector<unsigned short> &rowptr, std::vector<cldpc::QcCnEdge> &cn)
{
    rowptr.assign(J + 1, 0);
    cn.clear();
    for (int j = 0; j < J; j++) {
        for (int l = 0; l < L; l++)
            if (H[j * L + l] >= 0) cn.push_back({(unsigned short)l, (unsigned short)H[j * L + l]});
        rowptr[j + 1] = (unsigned short)cn.size();
    }
}

int main()
{
    int checked = 0;
    const int shapes[][2] = {{4, 24}, {32, 64}, {8, 24}, {6, 24}, {12, 24}, {3, 9}, {5, 5}};
    for (auto &sh : shapes) {
        const int J = sh[0], L = sh[1], cap = L / J;
        for (int trial = 0; trial < 200; trial++) {
            std::vector<int> H(J * L, -1), hidden(L);
            std::vector<int> perm(L);
            for (int l = 0; l < L; l++) perm[l] = l;
            for (int l = L - 1; l > 0; l--) std::swap(perm[l], perm[rnd() % (l + 1)]);
            for (int k = 0; k < L; k++) { hidden[perm[k]] = k / cap; H[(k / cap) * L + perm[k]] = (int)(rnd() % 96); }
            const int extra = (int)(rnd() % (J * L));
            for (int k = 0; k < extra; k++) H[(rnd() % J) * L + rnd() % L] = (int)(rnd() % 96);
            if (trial % 2) // the hidden assignment is then the ONLY one for some columns: single-block columns
                for (int l = 0; l < L; l += 3)
                    for (int j = 0; j < J; j++)
                        if (j != hidden[l]) H[j * L + l] = -1;
            std::vector<unsigned short> rowptr;
            std::vector<cldpc::QcCnEdge> cn;
            lists(J, L, H, rowptr, cn);
            std::vector<int> owner;
            if (!cldpc::qc2_local_assign(J, L, rowptr, cn, owner)) { printf("FAIL: no assignment found, J %d L %d trial %d\n", J, L, trial); return 1; }
            std::vector<int> cnt(J, 0);
            for (int l = 0; l < L; l++) {
                if (owner[l] < 0 || owner[l] >= J || H[owner[l] * L + l] < 0) { printf("FAIL: column %d handed to a row without it\n", l); return 1; }
                cnt[owner[l]]++;
            }
            for (int j = 0; j < J; j++)
                if (cnt[j] != cap) { printf("FAIL: row %d owns %d columns, not %d\n", j, cnt[j], cap); return 1; }
            checked++;
        }
        { // a row with fewer blocks than its share: no assignment
            std::vector<int> H(J * L, 5);
            for (int l = cap - 1; l < L; l++) H[0 * L + l] = -1;
            std::vector<unsigned short> rowptr;
            std::vector<cldpc::QcCnEdge> cn;
            lists(J, L, H, rowptr, cn);
            std::vector<int> owner;
            if (cldpc::qc2_local_assign(J, L, rowptr, cn, owner)) { printf("FAIL: accepted a row of %d blocks (share %d)\n", cap - 1, cap); return 1; }
        }
        { // two columns that live in one and the same row only, share 1
            if (cap == 1 && J >= 2) {
                std::vector<int> H(J * L, 5);
                for (int j = 1; j < J; j++) H[j * L + 0] = H[j * L + 1] = -1;
                std::vector<unsigned short> rowptr;
                std::vector<cldpc::QcCnEdge> cn;
                lists(J, L, H, rowptr, cn);
                std::vector<int> owner;
                if (cldpc::qc2_local_assign(J, L, rowptr, cn, owner)) { printf("FAIL: two columns of one row only, share 1\n"); return 1; }
            }
        }
    }
    { // L not a multiple of J
        std::vector<int> H(4 * 22, 1);
        std::vector<unsigned short> rowptr;
        std::vector<cldpc::QcCnEdge> cn;
        lists(4, 22, H, rowptr, cn);
        std::vector<int> owner;
        if (cldpc::qc2_local_assign(4, 22, rowptr, cn, owner)) { printf("FAIL: L %% J != 0 accepted\n"); return 1; }
    }
    printf("OK %d assignments valid\n", checked);
    return 0;
}
