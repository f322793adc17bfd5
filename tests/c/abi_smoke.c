/* C-linkage smoke test of the C ABI (no GPU needed): graph builders, GF tables, channel generators.
 * Built and run by tests/test_host_cpu.py::test_c_program_links_against_the_abi with plain gcc. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bldpc.h"
#include "nbldpc.h"

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    enum { J = 4, L = 24, Z = 96 };
    int H[J * L], wc[J + 1], wv[L + 1];
    if (bldpc_read_blockh(argv[1], J, L, H, wc, wv) != BLDPC_OK) { printf("ERR %s\n", bldpc_last_error()); return 1; }
    int *addr = (int *)malloc(sizeof(int) * L * Z * wv[L]);
    if (bldpc_transform_h(H, J, L, Z, wc, wv, addr, 0) != BLDPC_OK) return 1;
    long long sum = 0;
    for (int i = 0; i < L * Z * wv[L]; i++) sum += addr[i];
    int seed[3] = {173, 173, 173};
    float y[8];
    if (bldpc_awgn_channel_host(seed, bldpc_sigma(3.0f, 1, 0.0f), y, NULL, 8, 1) != BLDPC_OK) return 1;
    static unsigned mul[64 * 64], add[64 * 64], inv[64];
    if (nbldpc_gf_generate(64, 67, mul, add, inv) != NBLDPC_OK) { printf("ERR %s\n", nbldpc_last_error()); return 1; }
    if (bldpc_read_blockh("/nonexistent", J, L, H, wc, wv) == BLDPC_OK) return 1; /* errors are codes, never exit() */
    printf("Wc=%d Wv=%d addrsum=%lld y0=%.9g mul[2][33]=%u inv[5]=%u err=\"%s\"\n", wc[J], wv[L], sum, y[0], mul[2 * 64 + 33], inv[5],
           bldpc_last_error());
    free(addr);
    return 0;
}
