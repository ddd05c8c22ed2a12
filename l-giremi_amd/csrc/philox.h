// philox.h — Philox4x32-10 counter-based generator (Salmon et al., SC'11), usable
// from host and device code, plus the per-site parameters of the dense synthetic
// chromosome (include/lgmi.h: lgmi_synth_spec).
#pragma once
#include <stdint.h>

#ifdef __HIPCC__
#define LGMI_HD __host__ __device__ __forceinline__
#else
#define LGMI_HD inline
#endif

namespace lgmi {

struct U4 { uint32_t x, y, z, w; };

LGMI_HD void mulhilo(uint32_t a, uint32_t b, uint32_t& hi, uint32_t& lo) {
    uint64_t p = (uint64_t)a * (uint64_t)b;
    hi = (uint32_t)(p >> 32);
    lo = (uint32_t)p;
}

template <int ROUNDS>
LGMI_HD U4 philox4x32_r(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        uint32_t hi0, lo0, hi1, lo1;
        mulhilo(0xD2511F53u, c0, hi0, lo0);
        mulhilo(0xCD9E8D57u, c2, hi1, lo1);
        uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    U4 o; o.x = c0; o.y = c1; o.z = c2; o.w = c3;
    return o;
}

LGMI_HD U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    return philox4x32_r<10>(c0, c1, c2, c3, k0, k1);
}

// stream tags (counter word 2)
static const uint32_t TAG_CELL = 0x5eed0001u, TAG_HAP = 0x5eed0002u, TAG_SITE = 0x5eed0003u;
static const uint32_t TAG_PERM = 0x5eed0004u;

struct SynthSite { uint32_t e16; bool het, tri, snp; };

LGMI_HD SynthSite synth_site(const lgmi_synth_spec& sp, uint32_t s) {
    U4 h = philox4x32_10(s, 0xFFFFFFFEu, TAG_SITE, 0u, (uint32_t)sp.seed, (uint32_t)(sp.seed >> 32));
    SynthSite o;
    o.e16 = 3277u + h.x % 29491u;                    // alt fraction ~ U(0.05, 0.5)
    o.het = sp.het_every ? (s % sp.het_every == 0u) : false;
    o.tri = (h.y % 1024u) < sp.tri_per_1024;
    o.snp = !o.het && (h.z % 1024u) < sp.snp_per_1024;
    return o;
}

}  // namespace lgmi
