// bldpc_qc_assign.hpp -- host-only pieces of the fused QC kernels' plan: the block-list records and the matching behind the
// "local edges" (QcGeom / QcGeom2 LOC in bldpc_qc_kernel.hpp).  No HIP here: tests/cpp/qc_local_assign_host_test.cpp builds it with g++.
#pragma once
#include <algorithm>
#include <cstddef>
#include <vector>

namespace cldpc {

struct QcCnEdge { unsigned short col, shift; };  // block-row-major list of non-zero blocks
struct QcVnEdge { unsigned short e, shift; };    // per column, top->bottom: padded block index row*WC+position

// Local edges of the row / half-row kernels (LOC): hand every block column to ONE block row that contains it, 2*CPT columns per
// row (CPT per half-row).  A bipartite matching with row capacities, by augmenting paths (J, L are tens).  owner[l] = row or -1.
inline bool qc2_local_assign(int J, int L, const std::vector<unsigned short> &rowptr, const std::vector<QcCnEdge> &cn, std::vector<int> &owner)
{
    if (J <= 0 || L % J != 0) return false;
    const int cap = L / J;
    std::vector<std::vector<int>> rows_of(L);
    for (int j = 0; j < J; j++)
        for (int e = rowptr[j]; e < rowptr[j + 1]; e++) rows_of[cn[e].col].push_back(j);
    owner.assign(L, -1);
    std::vector<int> cnt(J, 0), seen(J, 0);
    struct Rec {
        static bool place(int l, const std::vector<std::vector<int>> &rows_of, std::vector<int> &owner, std::vector<int> &cnt, std::vector<int> &seen, int cap)
        {
            for (int j : rows_of[l]) {
                if (seen[j]) continue;
                seen[j] = 1;
                if (cnt[j] < cap) { owner[l] = j; cnt[j]++; return true; }
                for (size_t l2 = 0; l2 < owner.size(); l2++)
                    if (owner[l2] == j && place((int)l2, rows_of, owner, cnt, seen, cap)) { owner[l] = j; return true; } // l2 moved on (its new row counted it), l takes its place
            }
            return false;
        }
    };
    for (int l = 0; l < L; l++) {
        std::fill(seen.begin(), seen.end(), 0);
        if (!Rec::place(l, rows_of, owner, cnt, seen, cap)) return false;
    }
    return true;
}

} // namespace cldpc
