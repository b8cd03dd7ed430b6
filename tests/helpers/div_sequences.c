/* The short division sequences of the grid walk (openclraytracer_amd/csrc/tlg_kernels.hip: div_by, div_by_res) in plain C with fmaf, compared with the
 * IEEE quotient on this CPU: the arithmetic argument, independent of the GPU (tools/micro/div_check.hip runs the exhaustive comparison there).
 *   returns the number of mismatches over `count` pseudo-random operand pairs; *first receives the first mismatch (x bits, y bits)          */
#include <math.h>
#include <stdint.h>
#include <string.h>

static inline float bits_f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline uint32_t mix(uint32_t a) { a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16; return a; }

static float div3(float x, float y, float r) { const float q0 = x * r; const float e0 = fmaf(-y, q0, x); return fmaf(e0, r, q0); }
static float div5(float x, float y, float r) { const float q1 = div3(x, y, r); const float e1 = fmaf(-y, q1, x); return fmaf(e1, r, q1); }

/* x: sign random, exponent in [-90, 90), mantissa random */
static float operand(uint32_t h) { return bits_f((h & 0x807FFFFFu) | ((127u - 90u + (mix(h) % 180u)) << 23)); }

/* mode 0: y = (float)res, res = 1..res_max, three operations; mode 1: arbitrary y in [2^-24, 2^24) (every 256th with an all-ones significand), five operations */
uint64_t div_sequences_mismatches(int mode, uint32_t res_max, uint64_t count, uint32_t seed, uint32_t first[2])
{
    uint64_t bad = 0;
    for (uint64_t i = 0; i < count; i++) {
        const uint32_t h = mix((uint32_t)i * 2654435761u + seed);
        float y;
        if (mode == 0) y = (float)(1u + (h >> 8) % res_max);
        else {
            uint32_t ym = mix(h ^ 0x9e3779b9u) & 0x7FFFFFu;
            if ((i & 255u) == 255u) ym = 0x7FFFFFu;
            y = bits_f(((127u - 24u + (mix(h + 17u) % 48u)) << 23) | ym);
        }
        const float x = operand(mix(h + 0x51ed270bu));
        const float r = 1.0f / y;
        const float q = x / y;
        const float got = mode == 0 ? div3(x, y, r) : div5(x, y, r);
        if (f_bits(got) != f_bits(q)) { if (!bad) { first[0] = f_bits(x); first[1] = f_bits(y); } bad++; }
    }
    return bad;
}
