/* placeholder: replaced by the permutation-test specification */
#include <stdint.h>
int lgo_perm_rows(uint64_t n_rows, const uint32_t* row_i, const uint32_t* row_j, const uint32_t* counts,
                  uint32_t n_shuffles, uint64_t seed, double* p_out, uint32_t* exceed_out, int n_threads)
{
    (void)n_rows; (void)row_i; (void)row_j; (void)counts; (void)n_shuffles; (void)seed; (void)p_out; (void)exceed_out; (void)n_threads;
    return -8;
}
