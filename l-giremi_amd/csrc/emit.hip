// emit.hip — validity scan, 3x3 table assembly, MI and ordered row emission.
//
// Output order is the reference's: block, then itertools.combinations order of the
// sorted positions (src/giremi/mutual_information.py:10-12), restricted to pairs
// with a het_snp side when het_only (src/giremi/mismatch.py:392-396).  One wave
// owns one site row i and walks its partners j > i 64 at a time; pass 1 counts
// the pairs with >= min_common common reads (:19), a scan turns the counts into
// row offsets, pass 2 assembles the table, computes MI exactly as
// sklearn.metrics.mutual_info_score does (_supervised.py:903-923) and writes the
// row at its final position (wave ballot + popcount prefix).
#include "lgmi_internal.h"

namespace lgmi {

__device__ __forceinline__ double mi_from_table(const uint32_t t[9]) {
    // _supervised.py:903-923.  Classes absent among the common reads do not exist
    // (np.unique); a single surviving row or column returns exactly 0.0 (:909).
    uint32_t R[3], C[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        R[a] = t[3 * a] + t[3 * a + 1] + t[3 * a + 2];
        C[a] = t[a] + t[3 + a] + t[6 + a];
    }
    const uint32_t n = R[0] + R[1] + R[2];
    const int nr = (R[0] != 0) + (R[1] != 0) + (R[2] != 0);
    const int nc = (C[0] != 0) + (C[1] != 0) + (C[2] != 0);
    if (nr <= 1 || nc <= 1) return 0.0;
    const double dn = (double)n, log_n = log(dn);
    double sum = 0.0;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const uint32_t nab = t[3 * a + b];
            if (nab) {
                const double frac = (double)nab / dn;
                const double outer = (double)((uint64_t)R[a] * (uint64_t)C[b]);
                double term = frac * (log((double)nab) - log_n) + frac * (-log(outer) + log_n + log_n);
                if (fabs(term) < 2.220446049250313e-16) term = 0.0;
                sum += term;
            }
        }
    }
    return sum > 0.0 ? sum : 0.0;
}

// one work item = up to EMIT_SEG consecutive partners of one site row, 64 per trip (x sites: their partners j > i
// are consecutive sites, so the slots of a trip are consecutive columns of one slot-matrix row)
template <int PASS>
__device__ __forceinline__ void emit_item(const EmitArgs& a, uint32_t item, uint32_t lane, uint32_t* __restrict__ cst)
{
    const uint32_t i = a.items[item].x, seg = a.items[item].y;
    const SiteMap mi_ = a.smap[i];
    const BlockPlan bp = a.plans[mi_.block];
    const bool i_in_x = (mi_.xrow != NONE);
    uint32_t ncand, q0 = 0;
    if (i_in_x) {
        ncand = bp.site_end - 1u - i;
    } else {
        q0 = mi_.xnext;
        ncand = bp.nxs - q0;
    }
    const Col ci = a.cols[i];
    const uint32_t i_w0 = ci.w0, i_w1 = ci.w0 + ci.nw;
    const bool tri_i = a.tri[i] != 0;

    uint32_t running = 0;
    unsigned long long wsum = 0ull;
    uint64_t base_row = 0;
    unsigned long long my_sum = 0ull;
    if (PASS == 2) base_row = a.row_start[item];
    const uint32_t q_end = min(ncand, (seg + 1u) * EMIT_SEG);

    for (uint32_t qb = seg * EMIT_SEG; qb < q_end; qb += 64u) {
        const uint32_t q = qb + lane;
        bool valid = false;
        uint32_t j = 0, n_common = 0;
        uint64_t slot = 0;
        uint4 sl = make_uint4(0u, 0u, 0u, 0u);
        SiteMap mj;
        if (q < q_end) {
            j = i_in_x ? (i + 1u + q) : a.xlist[bp.xl_off + q0 + q];
            const Col cj = a.cols[j];
            const uint32_t lo = max(i_w0, cj.w0), hi = min(i_w1, cj.w0 + cj.nw);
            mj = a.smap[j];
            if (PASS == 1 && lo < hi) wsum += hi - lo;
            if (lo < hi) {
                const uint32_t xr = i_in_x ? mi_.xrow : mj.xrow;
                const uint32_t yc = i_in_x ? mj.ycol : mi_.ycol;
                slot = bp.slot_base + (uint64_t)xr * bp.ny_pad + yc;
                sl = a.slots[slot];
                n_common = sl.x;
            }
            valid = (n_common >= a.min_common);
        }
        const unsigned long long ball = __ballot(valid);
        if (PASS == 1) {
            running += (uint32_t)__popcll(ball);
        } else {
            if (valid) {
                const uint32_t prefix = (uint32_t)__popcll(ball & ((1ull << lane) - 1ull));
                const uint64_t r = base_row + running + prefix;
                uint32_t T[9];
                if (n_common == 0) {  // only reachable with min_common == 0
                    *a.err_flag = 1;
#pragma unroll
                    for (int k = 0; k < 9; ++k) T[k] = 0;
                } else {
                    // x = row site, y = col site of the slot matrix
                    const bool tri_j = a.tri[j] != 0;
                    const bool tri_x = i_in_x ? tri_i : tri_j;
                    const bool tri_y = i_in_x ? tri_j : tri_i;
                    const SiteMap& mx = i_in_x ? mi_ : mj;
                    const SiteMap& my = i_in_x ? mj : mi_;
                    const uint32_t N = n_common, r2 = sl.y, c2 = sl.z, n22 = sl.w;
                    uint32_t R1 = N - r2, C1 = N - c2, t12 = c2 - n22, t21 = r2 - n22;
                    if (tri_x) {
                        const uint64_t sb = bp.slot_base + (uint64_t)mx.prow * bp.ny_pad + my.ycol;
                        const uint4 sp = a.slots[sb];
                        R1 = sp.y;
                        t12 = sp.w;
                    }
                    if (tri_y) {
                        const uint64_t sc = bp.slot_base + (uint64_t)mx.xrow * bp.ny_pad + my.pcol;
                        const uint4 sp = a.slots[sc];
                        C1 = sp.z;
                        t21 = sp.w;
                    }
                    uint32_t t11;
                    if (tri_x && tri_y) {
                        t11 = a.slots[bp.slot_base + (uint64_t)mx.prow * bp.ny_pad + my.pcol].w;
                    } else if (tri_y) {
                        t11 = C1 - t21;
                    } else {
                        t11 = R1 - t12;
                    }
                    const uint32_t R0 = N - r2 - R1;
                    const uint32_t t20 = r2 - n22 - t21, t10 = R1 - t12 - t11;
                    const uint32_t t02 = c2 - n22 - t12, t01 = C1 - t21 - t11;
                    const uint32_t t00 = R0 - t01 - t02;
                    // t[ax][by]; rows are (i, j) ordered: transpose when x is j
                    if (i_in_x) {
                        T[0] = t00; T[1] = t01; T[2] = t02;
                        T[3] = t10; T[4] = t11; T[5] = t12;
                        T[6] = t20; T[7] = t21; T[8] = n22;
                    } else {
                        T[0] = t00; T[1] = t10; T[2] = t20;
                        T[3] = t01; T[4] = t11; T[5] = t21;
                        T[6] = t02; T[7] = t12; T[8] = n22;
                    }
                }
                const double mi = (n_common == 0) ? 0.0 : mi_from_table(T);
                a.out_i[r] = i;
                a.out_j[r] = j;
                a.out_mi[r] = mi;
                if (a.out_counts) {     // staged through LDS below: 9 dwords per row at a 36-byte stride are partial-line stores
#pragma unroll
                    for (int k = 0; k < 9; ++k) cst[9u * prefix + k] = T[k];
                }
                const unsigned long long fx = (unsigned long long)__double2ll_rn(mi * MEAN_SCALE);
                my_sum += fx;
                atomicAdd(&a.site_sum[j], fx);
                atomicAdd(&a.site_cnt[j], 1u);
            }
            if (a.out_counts) {
                // the valid lanes' tables are consecutive rows: write them as one contiguous run, 64 dwords per store.
                // The stores above sit in divergent code and other lanes read them here: order them explicitly
                // (LDS operations of one wave execute in order, but the compiler must not move them across)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const uint32_t n_dw = 9u * (uint32_t)__popcll(ball);
                uint32_t* dst = a.out_counts + 9ull * (base_row + running);
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    const uint32_t idx = 64u * k + lane;
                    if (idx < n_dw) dst[idx] = cst[idx];
                }
                __builtin_amdgcn_wave_barrier();     // the next trip's stores stay behind these loads
            }
            running += (uint32_t)__popcll(ball);
        }
    }
    if (PASS == 1) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) wsum += __shfl_xor(wsum, o);
        if (lane == 0) {
            a.row_cnt[item] = running;
            if (wsum) atomicAdd(a.word_pairs, wsum);
        }
    } else {
        // i side of the mean: integer wave reduction (order-independent, deterministic)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) my_sum += __shfl_xor(my_sum, o);
        if (lane == 0 && running) {
            atomicAdd(&a.site_sum[i], my_sum);
            atomicAdd(&a.site_cnt[i], running);
        }
    }
}

// A quad = up to four sites that are NOT x sites and whose slot-matrix columns are the four slots of one 64-byte
// line (consecutive, aligned y columns).  Their partners are the x sites after them: a COLUMN walk of the row-major slot
// matrix.  One site per wave read one 16-byte slot per cache line; here lane = (site of the quad, x rank inside the trip),
// so the four slots of a line are read by four neighbouring lanes and every fetched line is used whole.  Rows of a site
// still come out in increasing partner order (lanes of a site are ordered by x rank) at that site's own row offset.
template <int PASS>
__device__ __forceinline__ void emit_quad(const EmitArgs& a, uint32_t first_item, uint32_t n_it, uint32_t lane,
                                          uint32_t* __restrict__ cst)
{
    const uint32_t sa = lane & 3u, t = lane >> 2;
    const bool has_site = sa < n_it;
    const uint32_t item = first_item + (has_site ? sa : 0u);
    const uint32_t i = a.items[item].x;
    const SiteMap mi_ = a.smap[i];
    const BlockPlan bp = a.plans[mi_.block];                 // one block per unit (the planner never mixes blocks)
    const uint32_t q0 = mi_.xnext;                           // the first x rank after site i
    const Col ci = a.cols[i];
    const uint32_t i_w0 = ci.w0, i_w1 = ci.w0 + ci.nw;
    const bool tri_i = a.tri[i] != 0;
    uint32_t rmin = has_site ? q0 : 0xFFFFFFFFu;             // the quad's first x rank: every group of four lanes holds all sites
    rmin = min(rmin, (uint32_t)__shfl_xor((int)rmin, 1));
    rmin = min(rmin, (uint32_t)__shfl_xor((int)rmin, 2));
    const unsigned long long site_mask = 0x1111111111111111ull << sa;
    uint64_t base_row = 0;
    if (PASS == 2 && has_site) base_row = a.row_start[item];
    uint32_t running = 0;                                    // rows of this lane's site so far (equal in all its lanes)
    unsigned long long wsum = 0ull, my_sum = 0ull;

    for (uint32_t rb = rmin; rb < bp.nxs; rb += 16u) {       // wave-uniform bounds
        const uint32_t r = rb + t;
        bool valid = false;
        uint32_t j = 0, n_common = 0;
        uint64_t slot = 0;
        uint4 sl = make_uint4(0u, 0u, 0u, 0u);
        SiteMap mj = mi_;
        if (has_site && r >= q0 && r < bp.nxs) {
            j = a.xlist[bp.xl_off + r];                      // real x sites are the first nxs rows of the x list: xrow(j) == r
            const Col cj = a.cols[j];
            const uint32_t lo = max(i_w0, cj.w0), hi = min(i_w1, cj.w0 + cj.nw);
            mj = a.smap[j];
            if (PASS == 1 && lo < hi) wsum += hi - lo;
            if (lo < hi) {
                slot = bp.slot_base + (uint64_t)r * bp.ny_pad + mi_.ycol;
                sl = a.slots[slot];
                n_common = sl.x;
            }
            valid = (n_common >= a.min_common);
        }
        const unsigned long long ball = __ballot(valid);
        const unsigned long long mine = ball & site_mask;
        const uint32_t cnt_mine = (uint32_t)__popcll(mine);
        if (PASS == 2) {
            unsigned long long fx = 0ull;
            if (valid) {
                const uint32_t prefix = (uint32_t)__popcll(mine & ((1ull << lane) - 1ull));
                const uint64_t ro = base_row + running + prefix;
                uint32_t T[9];
                if (n_common == 0) {  // only reachable with min_common == 0
                    *a.err_flag = 1;
#pragma unroll
                    for (int k = 0; k < 9; ++k) T[k] = 0;
                } else {
                    // x = row site = j, y = col site = i (see emit_item for the derivation of the nine cells)
                    const bool tri_x = a.tri[j] != 0, tri_y = tri_i;
                    const uint32_t N = n_common, r2 = sl.y, c2 = sl.z, n22 = sl.w;
                    uint32_t R1 = N - r2, C1 = N - c2, t12 = c2 - n22, t21 = r2 - n22;
                    if (tri_x) {
                        const uint4 sp = a.slots[bp.slot_base + (uint64_t)mj.prow * bp.ny_pad + mi_.ycol];
                        R1 = sp.y;
                        t12 = sp.w;
                    }
                    if (tri_y) {
                        const uint4 sp = a.slots[bp.slot_base + (uint64_t)r * bp.ny_pad + mi_.pcol];
                        C1 = sp.z;
                        t21 = sp.w;
                    }
                    uint32_t t11;
                    if (tri_x && tri_y) t11 = a.slots[bp.slot_base + (uint64_t)mj.prow * bp.ny_pad + mi_.pcol].w;
                    else if (tri_y) t11 = C1 - t21;
                    else t11 = R1 - t12;
                    const uint32_t R0 = N - r2 - R1;
                    const uint32_t t20 = r2 - n22 - t21, t10 = R1 - t12 - t11;
                    const uint32_t t02 = c2 - n22 - t12, t01 = C1 - t21 - t11;
                    const uint32_t t00 = R0 - t01 - t02;
                    // rows are (i, j) ordered and x is j: the transpose
                    T[0] = t00; T[1] = t10; T[2] = t20;
                    T[3] = t01; T[4] = t11; T[5] = t21;
                    T[6] = t02; T[7] = t12; T[8] = n22;
                }
                const double mi = (n_common == 0) ? 0.0 : mi_from_table(T);
                a.out_i[ro] = i;
                a.out_j[ro] = j;
                a.out_mi[ro] = mi;
                if (a.out_counts) {
#pragma unroll
                    for (int k = 0; k < 9; ++k) cst[144u * sa + 9u * prefix + k] = T[k];
                }
                fx = (unsigned long long)__double2ll_rn(mi * MEAN_SCALE);
                my_sum += fx;
            }
            {
                // the partner's side of the mean: the four lanes of an x rank hold the same partner j — one atomic for
                // the four (same-address atomics of one wave serialise), integer sums so the total is the same
                unsigned long long f4 = fx;
                uint32_t c4 = valid ? 1u : 0u;
                f4 += __shfl_xor(f4, 1); c4 += (uint32_t)__shfl_xor((int)c4, 1);
                f4 += __shfl_xor(f4, 2); c4 += (uint32_t)__shfl_xor((int)c4, 2);
                if (sa == 0u && c4) {
                    const uint32_t jq = a.xlist[bp.xl_off + r];
                    atomicAdd(&a.site_sum[jq], f4);
                    atomicAdd(&a.site_cnt[jq], c4);
                }
            }
            if (a.out_counts) {
                // each site's tables of this trip are consecutive rows of that site: four contiguous runs
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const unsigned long long row_here = base_row + running;      // of this lane's site
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const uint32_t n_dw = 9u * (uint32_t)__popcll(ball & (0x1111111111111111ull << s));
                    const uint32_t lo32 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)row_here, s);
                    const uint32_t hi32 = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(row_here >> 32), s);
                    uint32_t* dst = a.out_counts + 9ull * (((unsigned long long)hi32 << 32) | lo32);
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const uint32_t idx = 64u * k + lane;
                        if (idx < n_dw) dst[idx] = cst[144u * s + idx];
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        running += cnt_mine;
    }
    if (PASS == 1) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) wsum += __shfl_xor(wsum, o);
        if (t == 0u && has_site) a.row_cnt[item] = running;
        if (lane == 0 && wsum) atomicAdd(a.word_pairs, wsum);
    } else {
        // the i side of the mean: sum over the lanes of a site (xor 4, 8, 16, 32 keeps the site)
#pragma unroll
        for (int o = 32; o >= 4; o >>= 1) my_sum += __shfl_xor(my_sum, o);
        if (t == 0u && has_site && running) {
            atomicAdd(&a.site_sum[i], my_sum);
            atomicAdd(&a.site_cnt[i], running);
        }
    }
}

// one wave per work unit: an x site's item, or a quad of other sites' items (units are built on the host, plan.cpp)
template <int PASS>
__global__ __launch_bounds__(256) void k_emit(EmitArgs a)
{
    __shared__ uint32_t cstage[4][64 * 9];          // pass 2: one wave's 3 x 3 tables, row-major, before the coalesced store
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t u = blockIdx.x * 4u + wv;
    if (u >= a.n_units) return;
    const uint2 unit = a.units[u];                  // (first item, items in the unit | kind << 16)
    if (unit.y >> 16) emit_quad<PASS>(a, unit.x, unit.y & 0xFFFFu, lane, cstage[wv]);
    else emit_item<PASS>(a, unit.x, lane, cstage[wv]);
}

void launch_emit_count(hipStream_t st, const EmitArgs& a) {
    if (!a.n_units) return;
    hipLaunchKernelGGL(k_emit<1>, dim3((a.n_units + 3) / 4), dim3(256), 0, st, a);
}
void launch_emit_write(hipStream_t st, const EmitArgs& a) {
    if (!a.n_units) return;
    hipLaunchKernelGGL(k_emit<2>, dim3((a.n_units + 3) / 4), dim3(256), 0, st, a);
}

// exclusive scan of n u32 counts into n+1 u64 offsets; one workgroup, each thread
// sums a contiguous segment (n is the number of sites: at most a few million)
__global__ __launch_bounds__(1024) void k_scan(const uint32_t* __restrict__ cnt,
                                               uint64_t* __restrict__ start, uint32_t n)
{
    __shared__ uint64_t part[1024];
    const uint32_t t = threadIdx.x;
    const uint32_t seg = (n + 1023u) / 1024u;
    const uint32_t b = min(t * seg, n), e = min(b + seg, n);
    uint64_t s = 0;
    for (uint32_t k = b; k < e; ++k) s += cnt[k];
    part[t] = s;
    __syncthreads();
    for (uint32_t off = 1; off < 1024u; off <<= 1) {  // Hillis-Steele inclusive scan
        uint64_t v = (t >= off) ? part[t - off] : 0ull;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint64_t run = part[t] - s;
    for (uint32_t k = b; k < e; ++k) { start[k] = run; run += cnt[k]; }
    if (t == 1023u) start[n] = part[1023];
}

void launch_scan(hipStream_t st, const uint32_t* cnt, uint64_t* start, uint32_t n) {
    hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, st, cnt, start, n);
}

__global__ void k_site_mean(uint32_t n, const unsigned long long* __restrict__ sum,
                            const uint32_t* __restrict__ cnt, double* __restrict__ mean)
{
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const uint32_t c = cnt[s];
    mean[s] = c ? ((double)sum[s] / MEAN_SCALE) / (double)c : __longlong_as_double(0x7ff8000000000000ll);
}

void launch_site_mean(hipStream_t st, uint32_t n_sites, const unsigned long long* sum,
                      const uint32_t* cnt, double* mean) {
    if (!n_sites) return;
    hipLaunchKernelGGL(k_site_mean, dim3((n_sites + 255) / 256), dim3(256), 0, st, n_sites, sum, cnt, mean);
}

__global__ void k_rows_mean(uint64_t n_rows, const uint32_t* __restrict__ ri,
                            const uint32_t* __restrict__ rj, const double* __restrict__ mi,
                            unsigned long long* sum, uint32_t* cnt)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const unsigned long long fx = (unsigned long long)__double2ll_rn(mi[r] * MEAN_SCALE);
    atomicAdd(&sum[ri[r]], fx);
    atomicAdd(&cnt[ri[r]], 1u);
    atomicAdd(&sum[rj[r]], fx);
    atomicAdd(&cnt[rj[r]], 1u);
}

void launch_rows_mean(hipStream_t st, uint64_t n_rows, const uint32_t* ri, const uint32_t* rj,
                      const double* mi, unsigned long long* sum, uint32_t* cnt) {
    if (!n_rows) return;
    hipLaunchKernelGGL(k_rows_mean, dim3((uint32_t)((n_rows + 255) / 256)), dim3(256), 0, st, n_rows, ri, rj, mi, sum, cnt);
}

}  // namespace lgmi
