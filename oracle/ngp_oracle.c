/*
 * TEST INFRASTRUCTURE ONLY (oracle/).  Plain-C, fp32, CPU restatement of the Instant-NGP hot
 * path of openxrlab/xrnerf.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load this library, and only as the checker -- never as the thing measured or shipped.
 *
 * Pinned against: (1) the reference's own kernels compiled for the CPU (oracle/_ref, built by
 * oracle/build.py from /root/reference/extensions/ngp_raymarch/src/ *.cu through oracle/shim) --
 * bit-exact for K1/K2/K6/K7/K11 indices & counts, <=1e-6 for the fp32 compositor; (2) the PCG32
 * known-answer vector of SURVEY.md section 8c; (3) committed golden fixtures tests/golden/ *.npz that
 * were generated from (1) by tests/golden/make_golden.py.
 * The tiny-cuda-nn half (hash grid, SH-4, fully fused MLP) is NOT in the reference tree
 * (requirements.txt:11, un-pinned git HEAD) and no reference test pins its numerics:
 *   ==> PARITY UNPINNED for xo_hashgrid_*, xo_sh4, xo_mlp_* : they restate the published
 *       Instant-NGP / tiny-cuda-nn algorithm and are anchored only on the
 *       reference's call sites (xrnerf/models/mlps/hashnerf_mlp.py:34-45,55-79,107-111).
 *
 * Compile: gcc -std=c11 -O2 -ffp-contract=off -fno-fast-math (x86-64 baseline: no FMA), so
 * `o + t*d` is mul-then-add exactly like the g++-compiled reference.
 *
 * Citations are relative to /root/reference/extensions/ngp_raymarch/ unless a path is given.
 */
#include <math.h>
#include <float.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---------------------------------------------------------------- constants (raymarch_shared.h:41-56) */
#define XO_STEPS 1024u
#define XO_CASCADES 8u
#define XO_GRID 128u
#define XO_GRID3 (128u * 128u * 128u)
static const float XO_SQRT3 = 1.73205080757f;
static inline float xo_min_step(void) { return XO_SQRT3 / (float)XO_STEPS; }      /* STEPSIZE()          :45 */
static inline float xo_max_step(void) {                                            /* MAX_CONE_STEPSIZE() :51 */
    return xo_min_step() * (float)(1 << (XO_CASCADES - 1)) * (float)XO_STEPS / (float)XO_GRID;
}
static inline float xo_max_warp_step(void) { return xo_min_step() * (float)(1 << (XO_CASCADES - 1)); } /* :114 */

static int g_threads = 1;
void xo_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int xo_get_threads(void) { return g_threads; }

/* ---------------------------------------------------------------- PCG32 (include/op_include/pcg32/pcg32.h:39-166) */
typedef struct { uint64_t state, inc; } xo_pcg32;
#define XO_PCG_MULT 0x5851f42d4c957f2dULL
static inline uint32_t xo_pcg_next(xo_pcg32* r) {                                  /* :62-68 */
    uint64_t old = r->state;
    r->state = old * XO_PCG_MULT + r->inc;
    uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    uint32_t rot = (uint32_t)(old >> 59u);
    return (xs >> rot) | (xs << ((~rot + 1u) & 31));
}
static inline void xo_pcg_seed(xo_pcg32* r, uint64_t initstate, uint64_t initseq) { /* :53-59 */
    r->state = 0u; r->inc = (initseq << 1u) | 1u;
    xo_pcg_next(r); r->state += initstate; xo_pcg_next(r);
}
static inline float xo_pcg_float(xo_pcg32* r) {                                    /* :103-112 */
    union { uint32_t u; float f; } x;
    x.u = (xo_pcg_next(r) >> 9) | 0x3f800000u;
    return x.f - 1.0f;
}
static inline void xo_pcg_advance(xo_pcg32* r, uint64_t delta) {                   /* :145-166 */
    uint64_t cur_mult = XO_PCG_MULT, cur_plus = r->inc, acc_mult = 1u, acc_plus = 0u;
    while (delta > 0) {
        if (delta & 1) { acc_mult *= cur_mult; acc_plus = acc_plus * cur_mult + cur_plus; }
        cur_plus = (cur_mult + 1) * cur_plus;
        cur_mult *= cur_mult;
        delta /= 2;
    }
    r->state = acc_mult * r->state + acc_plus;
}
/* state of the reference's per-TU `static pcg32 rng{9121}` (raymarch_shared.h:38) after
 * `ncalls` launches, each of which ends with rng.advance() = 2^32 (ray_sampler.cu:198) */
void xo_pcg32_host_state(uint64_t seed, uint64_t ncalls, uint64_t* state, uint64_t* inc) {
    xo_pcg32 r; xo_pcg_seed(&r, seed, 1u);
    for (uint64_t c = 0; c < ncalls; ++c) xo_pcg_advance(&r, 1ull << 32);
    *state = r.state; *inc = r.inc;
}
void xo_pcg32_probe(uint64_t seed, uint64_t advance_by, uint64_t* state, uint64_t* inc,
                    uint32_t* u5, float* f3) {
    xo_pcg32 r; xo_pcg_seed(&r, seed, 1u); xo_pcg_advance(&r, advance_by);
    *state = r.state; *inc = r.inc;
    xo_pcg32 a = r; for (int i = 0; i < 5; ++i) u5[i] = xo_pcg_next(&a);
    xo_pcg32 b = r; for (int i = 0; i < 3; ++i) f3[i] = xo_pcg_float(&b);
}

/* ---------------------------------------------------------------- Morton (raymarch_shared.h:128-136,753-768) */
static inline uint32_t xo_expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu; v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u; v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
static inline uint32_t xo_morton3d(uint32_t x, uint32_t y, uint32_t z) {
    return xo_expand_bits(x) | (xo_expand_bits(y) << 1) | (xo_expand_bits(z) << 2);
}
static inline uint32_t xo_morton3d_invert(uint32_t x) {
    x = x & 0x49249249; x = (x | (x >> 2)) & 0xc30c30c3; x = (x | (x >> 4)) & 0x0f00f00f;
    x = (x | (x >> 8)) & 0xff0000ff; x = (x | (x >> 16)) & 0x0000ffff;
    return x;
}
uint32_t xo_morton3d_api(uint32_t x, uint32_t y, uint32_t z) { return xo_morton3d(x, y, z); }
uint32_t xo_morton3d_invert_api(uint32_t m) { return xo_morton3d_invert(m); }

/* ---------------------------------------------------------------- march helpers (ray_sampler_header.h) */
static inline float xo_clamp(float v, float lo, float hi) { return v < lo ? lo : (hi < v ? hi : v); } /* :24 */
static inline float xo_calc_dt(float t, float cone) { return xo_clamp(t * cone, xo_min_step(), xo_max_step()); } /* :25 */
static inline float xo_warp_dt(float dt) {                                         /* raymarch_shared.h:112-116 */
    float mx = xo_max_warp_step();
    return (dt - xo_min_step()) / (mx - xo_min_step());
}
static inline float xo_unwarp_dt(float dt) {                                       /* ray_sampler_header.h:388-392 */
    float mx = xo_max_warp_step();
    return dt * (mx - xo_min_step()) + xo_min_step();
}
/* the shim's (and g++'s) `min(a,b)` is `a < b ? a : b` -- kept verbatim because the NaN
 * behaviour (0*inf when a direction component is exactly 0) is order dependent */
static inline float xo_fmin_lt(float a, float b) { return a < b ? a : b; }
static inline int xo_imin(int a, int b) { return a < b ? a : b; }
static inline int xo_imax(int a, int b) { return a > b ? a : b; }

static inline int xo_mip_from_pos(const float p[3]) {                              /* :37-43 */
    float m = fabsf(p[0] - 0.5f);
    float b = fabsf(p[1] - 0.5f); if (b > m) m = b;      /* Eigen maxCoeff: strict > scan */
    float c = fabsf(p[2] - 0.5f); if (c > m) m = c;
    int e; frexpf(m, &e);
    return xo_imin((int)XO_CASCADES - 1, xo_imax(0, e + 1));
}
static inline int xo_mip_from_dt(float dt, const float p[3]) {                     /* :45-54 */
    int mip = xo_mip_from_pos(p);
    dt *= (float)(2 * XO_GRID);
    if (dt < 1.f) return mip;
    int e; frexpf(dt, &e);
    return xo_imin((int)XO_CASCADES - 1, xo_imax(e, mip));
}
static inline uint32_t xo_cascaded_idx(const float pos[3], uint32_t mip) {         /* :298-313 */
    float s = scalbnf(1.0f, -(int)mip);
    int c[3];
    for (int a = 0; a < 3; ++a) {
        float q = pos[a] - 0.5f; q = q * s; q = q + 0.5f;
        int i = (int)(q * (float)XO_GRID);
        c[a] = i < 0 ? 0 : (i > (int)XO_GRID - 1 ? (int)XO_GRID - 1 : i);
    }
    return xo_morton3d((uint32_t)c[0], (uint32_t)c[1], (uint32_t)c[2]);
}
static inline int xo_occupied(const float pos[3], const uint8_t* bf, uint32_t mip) { /* :315-319 */
    uint32_t idx = xo_cascaded_idx(pos, mip);
    return bf[idx / 8 + (XO_GRID3 * mip) / 8] & (1 << (idx % 8));
}
static inline float xo_sign(float x) { return copysignf(1.0f, x); }                /* raymarch_shared.h:172-175 */
static inline float xo_dist_next_voxel(const float pos[3], const float d[3], const float id[3],
                                       uint32_t res) {                              /* :271-280 */
    float r = (float)res;
    float px = r * pos[0], py = r * pos[1], pz = r * pos[2];
    float tx = (floorf(px + 0.5f + 0.5f * xo_sign(d[0])) - px) * id[0];
    float ty = (floorf(py + 0.5f + 0.5f * xo_sign(d[1])) - py) * id[1];
    float tz = (floorf(pz + 0.5f + 0.5f * xo_sign(d[2])) - pz) * id[2];
    float t = xo_fmin_lt(xo_fmin_lt(tx, ty), tz);
    return fmaxf(t / r, 0.0f);
}
static inline float xo_advance_next_voxel(float t, float cone, const float pos[3], const float d[3],
                                          const float id[3], uint32_t res) {        /* :282-296 */
    float target = t + xo_dist_next_voxel(pos, d, id, res);
    do { t += xo_calc_dt(t, cone); } while (t < target);
    return t;
}
/* BoundingBox::ray_intersect (raymarch_shared.h:506-563): returns tmin (tmax unused by K1) */
static inline float xo_aabb_tmin(float lo, float hi, const float o[3], const float d[3]) {
    float tmin = (lo - o[0]) / d[0], tmax = (hi - o[0]) / d[0];
    if (tmin > tmax) { float s = tmin; tmin = tmax; tmax = s; }
    float tymin = (lo - o[1]) / d[1], tymax = (hi - o[1]) / d[1];
    if (tymin > tymax) { float s = tymin; tymin = tymax; tymax = s; }
    if (tmin > tymax || tymin > tmax) return FLT_MAX;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    float tzmin = (lo - o[2]) / d[2], tzmax = (hi - o[2]) / d[2];
    if (tzmin > tzmax) { float s = tzmin; tzmin = tzmax; tzmax = s; }
    if (tmin > tzmax || tzmin > tmax) return FLT_MAX;
    if (tzmin > tmin) tmin = tzmin;
    return tmin;
}
static inline int xo_contains(float lo, float hi, const float p[3]) {              /* :570-575 */
    return p[0] >= lo && p[0] <= hi && p[1] >= lo && p[1] <= hi && p[2] >= lo && p[2] <= hi;
}

/* ---------------------------------------------------------------- K1 rays_sampler (src/ray_sampler.cu:5-116)
 * Serial in ray order => `base` is the exclusive prefix sum of the per-ray counts, which is one
 * valid schedule of the reference's atomicAdd (and the one the CPU-shim build produces).
 * metadata / xforms are loaded-but-dead in the reference (:34-38) and therefore not taken here;
 * img_ids only has to be a valid index there and is ignored here. */
void xo_rays_sampler(const float* rays_o, const float* rays_d, const uint8_t* bitfield, int n_rays,
                     float aabb0, float aabb1, float near_distance, float cone, uint32_t max_samples,
                     uint64_t rng_state, uint64_t rng_inc, float* coords_out, int32_t* rays_index,
                     int32_t* numsteps_out, uint32_t* counter2 /* [rays, samples] */) {
    for (int i = 0; i < n_rays; ++i) {
        xo_pcg32 rng = { rng_state, rng_inc };
        xo_pcg_advance(&rng, (uint64_t)((uint32_t)i * 8u));                        /* :31 */
        const float* o = rays_o + 3 * i; const float* d = rays_d + 3 * i;
        float tmin = xo_aabb_tmin(aabb0, aabb1, o, d);
        tmin = fmaxf(tmin, near_distance);                                         /* :46 */
        float startt = tmin;
        startt += xo_calc_dt(startt, cone) * xo_pcg_float(&rng);                   /* :50 */
        float id[3] = { 1.0f / d[0], 1.0f / d[1], 1.0f / d[2] };
        uint32_t j = 0; float t = startt; float pos[3];
        for (;;) {                                                                 /* :58-72 */
            pos[0] = o[0] + t * d[0]; pos[1] = o[1] + t * d[1]; pos[2] = o[2] + t * d[2];
            if (!(xo_contains(aabb0, aabb1, pos) && j < XO_STEPS)) break;
            float dt = xo_calc_dt(t, cone);
            uint32_t mip = (uint32_t)xo_mip_from_dt(dt, pos);
            if (xo_occupied(pos, bitfield, mip)) { ++j; t += dt; }
            else t = xo_advance_next_voxel(t, cone, pos, d, id, XO_GRID >> mip);
        }
        uint32_t numsteps = j;
        uint32_t base = counter2[1]; counter2[1] += numsteps;                      /* :75 */
        if (base + numsteps > max_samples) {                                       /* :76-82 */
            numsteps_out[2 * i + 0] = 0; numsteps_out[2 * i + 1] = (int32_t)base;
            continue;
        }
        uint32_t ray_idx = counter2[0]; counter2[0] += 1;                          /* :86 */
        rays_index[i] = (int32_t)ray_idx;
        numsteps_out[2 * i + 0] = (int32_t)numsteps; numsteps_out[2 * i + 1] = (int32_t)base;
        if (j == 0) { rays_index[i] = -1; continue; }
        float wd[3] = { (d[0] + 1.0f) * 0.5f, (d[1] + 1.0f) * 0.5f, (d[2] + 1.0f) * 0.5f }; /* warp_direction */
        float diag = aabb1 - aabb0;
        t = startt; j = 0;
        for (;;) {                                                                 /* :99-115 */
            pos[0] = o[0] + t * d[0]; pos[1] = o[1] + t * d[1]; pos[2] = o[2] + t * d[2];
            if (!(xo_contains(aabb0, aabb1, pos) && j < numsteps)) break;
            float dt = xo_calc_dt(t, cone);
            uint32_t mip = (uint32_t)xo_mip_from_dt(dt, pos);
            if (xo_occupied(pos, bitfield, mip)) {
                float* c = coords_out + 7 * (size_t)(base + j);
                c[0] = (pos[0] - aabb0) / diag; c[1] = (pos[1] - aabb0) / diag; c[2] = (pos[2] - aabb0) / diag;
                c[3] = xo_warp_dt(dt);
                c[4] = wd[0]; c[5] = wd[1]; c[6] = wd[2];
                ++j; t += dt;
            } else t = xo_advance_next_voxel(t, cone, pos, d, id, XO_GRID >> mip);
        }
    }
}

/* ---------------------------------------------------------------- K2 compacted_coord (src/compacted_coord.cu:6-77)
 * The transmittance loop (:39-58) has no effect on any output (its `break` is commented out,
 * :41-44), so network_output is not an input of this restatement. */
void xo_compacted_coord(const float* coords_in, const int32_t* numsteps_in, int n_rays,
                        uint32_t max_compacted, float* coords_out, int32_t* numsteps_out,
                        uint32_t* rays_counter, uint32_t* numstep_counter) {
    for (int i = 0; i < n_rays; ++i) {
        uint32_t n = (uint32_t)numsteps_in[2 * i], base = (uint32_t)numsteps_in[2 * i + 1];
        uint32_t cbase = *numstep_counter; *numstep_counter += n;                  /* :63 */
        uint32_t room = max_compacted - (max_compacted < cbase ? max_compacted : cbase);
        uint32_t nc = room < n ? room : n;                                         /* :64 */
        numsteps_out[2 * i] = (int32_t)nc; numsteps_out[2 * i + 1] = (int32_t)cbase;
        if (nc == 0) continue;
        *rays_counter += 1;
        memcpy(coords_out + 7 * (size_t)cbase, coords_in + 7 * (size_t)base, sizeof(float) * 7 * nc);
    }
}

/* ---------------------------------------------------------------- activations
 * ENerfActivation {None,ReLU,Logistic,Exponential}=0..3 (raymarch_shared.h:619-625) */
static inline float xo_logistic(float x) { return 1.0f / (1.0f + expf(-x)); }      /* :615-618 */
static inline float xo_act_rgb(float v, int a) {                                   /* ray_sampler_header.h:440-456 */
    switch (a) { case 0: return v; case 1: return v > 0.f ? v : 0.f; case 2: return xo_logistic(v);
                 default: return expf(xo_clamp(v, -10.f, 10.f)); }
}
static inline float xo_act_density(float v, int a) {                               /* raymarch_shared.h:626-642 */
    switch (a) { case 0: return v; case 1: return v > 0.f ? v : 0.f; case 2: return xo_logistic(v);
                 default: return expf(v); }
}
static inline float xo_dact_rgb(float v, int a) {                                  /* ray_sampler_header.h:534-553 */
    switch (a) { case 0: return 1.f; case 1: return v > 0.f ? 1.f : 0.f;
                 case 2: { float s = xo_logistic(v); return s * (1 - s); }
                 default: return expf(xo_clamp(v, -10.f, 10.f)); }
}
static inline float xo_dact_density(float v, int a) {                              /* :555-574 */
    switch (a) { case 0: return 1.f; case 1: return v > 0.f ? 1.f : 0.f;
                 case 2: { float s = xo_logistic(v); return s * (1 - s); }
                 default: return expf(xo_clamp(v, -15.f, 15.f)); }
}

/* ---------------------------------------------------------------- K3 compute_rgbs (src/calc_rgb.cu:6-67) */
void xo_calc_rgb_forward(const float* raw, const float* coords, const int32_t* numsteps,
                         const int32_t* numsteps_c, const float* bg, int n_rays, int rgb_act,
                         int density_act, float* rgb_out) {
    for (int i = 0; i < n_rays; ++i) {
        const float* b = bg + 3 * i;
        uint32_t n = (uint32_t)numsteps_c[2 * i], base = (uint32_t)numsteps_c[2 * i + 1];
        if (n == 0) { rgb_out[3 * i] = b[0]; rgb_out[3 * i + 1] = b[1]; rgb_out[3 * i + 2] = b[2]; continue; }
        float T = 1.f, c[3] = { 0, 0, 0 };
        uint32_t k = 0;
        for (; k < n; ++k) {
            const float* r = raw + 4 * (size_t)(base + k);
            float dt = xo_unwarp_dt(coords[7 * (size_t)(base + k) + 3]);
            float density = xo_act_density(r[3], density_act);
            float alpha = 1.f - expf(-density * dt);
            float w = alpha * T;
            for (int ch = 0; ch < 3; ++ch) c[ch] += w * xo_act_rgb(r[ch], rgb_act);
            T *= (1.f - alpha);
        }
        if (k == (uint32_t)numsteps[2 * i]) for (int ch = 0; ch < 3; ++ch) c[ch] += T * b[ch];  /* :61-64 */
        rgb_out[3 * i] = c[0]; rgb_out[3 * i + 1] = c[1]; rgb_out[3 * i + 2] = c[2];
    }
}

/* ---------------------------------------------------------------- K5 compute_rgbs_inference (:144-206) */
void xo_calc_rgb_inference(const float* raw, const float* coords, const int32_t* numsteps,
                           const float* bg3, int n_rays, int rgb_act, int density_act,
                           float* rgb_out, float* alpha_out) {
    for (int i = 0; i < n_rays; ++i) {
        uint32_t n = (uint32_t)numsteps[2 * i], base = (uint32_t)numsteps[2 * i + 1];
        if (n == 0) { for (int ch = 0; ch < 3; ++ch) rgb_out[3 * i + ch] = bg3[ch]; alpha_out[i] = 0; continue; }
        float T = 1.f, c[3] = { 0, 0, 0 };
        for (uint32_t k = 0; k < n; ++k) {
            const float* r = raw + 4 * (size_t)(base + k);
            float dt = xo_unwarp_dt(coords[7 * (size_t)(base + k) + 3]);
            float density = xo_act_density(r[3], density_act);
            float alpha = 1.f - expf(-density * dt);
            float w = alpha * T;
            for (int ch = 0; ch < 3; ++ch) c[ch] += w * xo_act_rgb(r[ch], rgb_act);
            T *= (1.f - alpha);
        }
        for (int ch = 0; ch < 3; ++ch) rgb_out[3 * i + ch] = c[ch] + T * bg3[ch];
        alpha_out[i] = 1 - T;
    }
}

/* ---------------------------------------------------------------- K4 compute_rgbs_grad (:71-140)
 * dloss_doutput rows that no ray covers are left untouched (caller zero-fills,
 * xrnerf/models/renders/hashnerf_render.py:121-123). */
void xo_calc_rgb_backward(const float* raw, const int32_t* numsteps_c, const float* coords,
                          const float* grad_rgb, const float* rgb_final, const float* density_grid_mean,
                          int n_rays, int rgb_act, int density_act, float* dloss_doutput) {
    float loss_scale = 128; loss_scale /= (float)(uint32_t)n_rays;                 /* :92-93 */
    const float l2 = rgb_act == 3 ? 1e-4f : 0.0f;                                  /* :103 */
    const float l1 = density_grid_mean[0] < 0.01f ? 1e-4f : 0.0f;                  /* :104 */
    for (int i = 0; i < n_rays; ++i) {
        uint32_t n = (uint32_t)numsteps_c[2 * i], base = (uint32_t)numsteps_c[2 * i + 1];
        const float* g = grad_rgb + 3 * i; const float* cf = rgb_final + 3 * i;
        float T = 1.f, c2[3] = { 0, 0, 0 };
        for (uint32_t k = 0; k < n; ++k) {
            const float* r = raw + 4 * (size_t)(base + k);
            float* o = dloss_doutput + 4 * (size_t)(base + k);
            float rgb[3] = { xo_act_rgb(r[0], rgb_act), xo_act_rgb(r[1], rgb_act), xo_act_rgb(r[2], rgb_act) };
            float dt = xo_unwarp_dt(coords[7 * (size_t)(base + k) + 3]);
            float density = xo_act_density(r[3], density_act);
            float alpha = 1.f - expf(-density * dt);
            float w = alpha * T;
            for (int ch = 0; ch < 3; ++ch) c2[ch] += w * rgb[ch];
            T *= (1.f - alpha);
            float suffix[3] = { cf[0] - c2[0], cf[1] - c2[1], cf[2] - c2[2] };
            for (int ch = 0; ch < 3; ++ch)
                o[ch] = loss_scale * ((w * g[ch]) * xo_dact_rgb(r[ch], rgb_act) + fmaxf(0.0f, l2 * r[ch]));
            float dd = xo_dact_density(r[3], density_act);
            float dot = g[0] * (T * rgb[0] - suffix[0]) + g[1] * (T * rgb[1] - suffix[1]) + g[2] * (T * rgb[2] - suffix[2]);
            float dmlp = dd * (dt * dot);
            o[3] = loss_scale * dmlp + (r[3] < 0 ? -l1 : 0.0f);
        }
    }
}

/* ---------------------------------------------------------------- K6 (src/generate_grid_samples_nerf_nonuniform.cu:6-42) */
void xo_generate_grid_samples(const float* grid, uint32_t step, uint32_t n_elements, uint32_t n_cascades,
                              float thresh, float aabb0, float aabb1, uint64_t rng_state, uint64_t rng_inc,
                              float* positions, int32_t* indices) {
    float diag = aabb1 - aabb0;
    for (uint32_t i = 0; i < n_elements; ++i) {
        xo_pcg32 rng = { rng_state, rng_inc };
        xo_pcg_advance(&rng, (uint64_t)(i * 4u));
        uint32_t level = (uint32_t)(xo_pcg_float(&rng) * (float)n_cascades) % n_cascades;
        uint32_t idx = 0;
        for (uint32_t j = 0; j < 10; ++j) {
            idx = ((i + step * n_elements) * 56924617u + j * 19349663u + 96925573u) % XO_GRID3;
            idx += level * XO_GRID3;
            if (grid[idx] > thresh) break;
        }
        uint32_t pidx = idx % XO_GRID3;
        float xyz[3] = { (float)xo_morton3d_invert(pidx >> 0), (float)xo_morton3d_invert(pidx >> 1),
                         (float)xo_morton3d_invert(pidx >> 2) };
        float sc = scalbnf(1.0f, (int)level);
        for (int a = 0; a < 3; ++a) {
            float u = xo_pcg_float(&rng);
            float p = ((xyz[a] + u) / (float)XO_GRID - 0.5f) * sc + 0.5f;
            positions[3 * (size_t)i + a] = (p - aabb0) / diag;
        }
        indices[i] = (int32_t)idx;
    }
}

/* ---------------------------------------------------------------- K7 (src/mark_untrained_density_grid.cu:6-52)
 * Restated as "0 if visible from any training camera else -1": the reference only rewrites on a
 * sign mismatch of an UNINITIALISED buffer (xrnerf/models/samplers/utils/mark_untrained_density_grid.py:18);
 * on a zero-initialised buffer the two coincide, which is how the _ref comparison is run.
 * res0/res1 are passed as the reference passes them: (H, W) -> resolution.x()=H, .y()=W. */
void xo_mark_untrained(const float* focal /*[n,2]*/, const float* xforms /*[n,4,3]*/, uint32_t n_elements,
                       int n_img, int res0, int res1, float* grid) {
    float hx = res0 * 0.5f, hy = res1 * 0.5f;
    for (uint32_t i = 0; i < n_elements; ++i) {
        uint32_t level = i / XO_GRID3, pidx = i % XO_GRID3;
        float sc = scalbnf(1.0f, (int)level);
        float c[3] = { (float)xo_morton3d_invert(pidx >> 0) + 0.5f, (float)xo_morton3d_invert(pidx >> 1) + 0.5f,
                       (float)xo_morton3d_invert(pidx >> 2) + 0.5f };
        float pos[3];
        for (int a = 0; a < 3; ++a) pos[a] = (c[a] / (float)XO_GRID - 0.5f) * sc + 0.5f;
        float radius = 0.5f * XO_SQRT3 * sc / (float)XO_GRID;
        int seen = 0;
        for (int j = 0; j < n_img && !seen; ++j) {
            const float* m = xforms + 12 * j;     /* Matrix<float,3,4> column major: col c = m[3c..3c+2] */
            float pl[3] = { pos[0] - m[9], pos[1] - m[10], pos[2] - m[11] };
            float x = pl[0] * m[0] + pl[1] * m[1] + pl[2] * m[2];
            float y = pl[0] * m[3] + pl[1] * m[4] + pl[2] * m[5];
            float z = pl[0] * m[6] + pl[1] * m[7] + pl[2] * m[8];
            if (z > 0.f) {
                if (fabsf(x) - radius < z / focal[2 * j] * hx && fabsf(y) - radius < z / focal[2 * j + 1] * hy) seen = 1;
            }
        }
        grid[i] = seen ? 0.f : -1.f;
    }
}

/* ---------------------------------------------------------------- K8 (src/splat_grid_samples_nerf_max_nearest_neighbor.cu:7-28) */
void xo_splat(const float* mlp_out, const int32_t* indices, int padded_width, uint32_t n, float* grid_tmp) {
    for (uint32_t i = 0; i < n; ++i) {
        float thick = expf(mlp_out[(size_t)i * padded_width]) * scalbnf(xo_min_step(), 0);
        uint32_t u, *slot = (uint32_t*)&grid_tmp[(uint32_t)indices[i]];
        memcpy(&u, &thick, 4);
        if (u > *slot) *slot = u;
    }
}
/* ---------------------------------------------------------------- K9 (src/ema_grid_samples_nerf.cu:4-27) */
void xo_ema(const float* grid_tmp, uint32_t n, float decay, float* grid) {
    for (uint32_t i = 0; i < n; ++i) {
        float prev = grid[i];
        grid[i] = (prev < 0.f) ? prev : fmaxf(prev * decay, grid_tmp[i]);
    }
}
/* ---------------------------------------------------------------- K10 mean (src/update_bitfield.cu:3-22,98-101)
 * Sum of max(v,0)/G^3 over level 0. The reference's own value depends on its float-atomic
 * order; here a float4-lane-major serial sum (same as the CPU-shim build). Tested to rel 1e-5. */
float xo_density_mean(const float* grid) {
    float s = 0.f;
    for (uint32_t e = 0; e < XO_GRID3; ++e) s += fmaxf(grid[e], 0.f) / (float)XO_GRID3;
    return s;
}
/* ---------------------------------------------------------------- K11 (src/update_bitfield.cu:24-71,103-112) */
void xo_bitfield(const float* grid, float mean, uint8_t* bitfield) {
    float thresh = 0.01f < mean ? 0.01f : mean;                                    /* :35 */
    for (uint32_t i = 0; i < XO_GRID3 / 8 * XO_CASCADES; ++i) {
        uint8_t bits = 0;
        for (int j = 0; j < 8; ++j) bits |= grid[(size_t)i * 8 + j] > thresh ? (uint8_t)(1 << j) : 0;
        bitfield[i] = bits;
    }
    for (uint32_t level = 1; level < XO_CASCADES; ++level) {
        const uint8_t* prev = bitfield + (size_t)(XO_GRID3 * (level - 1)) / 8;
        uint8_t* next = bitfield + (size_t)(XO_GRID3 * level) / 8;
        for (uint32_t i = 0; i < XO_GRID3 / 64; ++i) {
            uint8_t bits = 0;
            for (int j = 0; j < 8; ++j) bits |= prev[i * 8 + j] > 0 ? (uint8_t)(1 << j) : 0;
            uint32_t x = xo_morton3d_invert(i >> 0) + XO_GRID / 8, y = xo_morton3d_invert(i >> 1) + XO_GRID / 8,
                     z = xo_morton3d_invert(i >> 2) + XO_GRID / 8;
            next[xo_morton3d(x, y, z)] |= bits;
        }
    }
}

/* ================================================================ tiny-cuda-nn half  (PARITY UNPINNED)
 * Restated from the published algorithm; call sites
 * xrnerf/models/mlps/hashnerf_mlp.py:34-45 (construction) and :55-79,107-111 (forward). */

/* Per-level table geometry.  scale_l = 2^(l*log2(b))*N_min - 1 ; res_l = ceil(scale_l)+1 ;
 * T_l = min(roundup8(res_l^3), 2^log2T) ; offsets = prefix sum (in ENTRIES, each of F floats).
 * Computed once on the host and handed to BOTH the oracle and the HIP kernels, so no
 * device-side exp2f can make them disagree on an index. */
void xo_hashgrid_meta(int n_levels, int log2_hashmap_size, int base_resolution, double per_level_scale,
                      float* scale /*[L]*/, uint32_t* resolution /*[L]*/, uint32_t* offset /*[L+1]*/) {
    float log2b = log2f((float)per_level_scale);   /* tcnn keeps per_level_scale and its log2 as float */
    uint32_t off = 0;
    for (int l = 0; l < n_levels; ++l) {
        float s = exp2f((float)l * log2b) * (float)base_resolution - 1.0f;
        uint32_t res = (uint32_t)ceilf(s) + 1u;
        scale[l] = s; resolution[l] = res; offset[l] = off;
        double cube = (double)res * res * res;
        uint32_t n = cube > 2147483647.0 ? 2147483647u : (uint32_t)cube;
        n = (n + 7u) / 8u * 8u;
        uint32_t cap = 1u << log2_hashmap_size;
        if (n > cap) n = cap;
        off += n;
    }
    offset[n_levels] = off;
}
static inline uint32_t xo_grid_index(uint32_t cx, uint32_t cy, uint32_t cz, uint32_t res, uint32_t hsize) {
    uint32_t stride = 1, index = 0;
    uint32_t c[3] = { cx, cy, cz };
    for (int d = 0; d < 3 && stride <= hsize; ++d) { index += c[d] * stride; stride *= res; }
    if (hsize < stride) index = (cx * 1u) ^ (cy * 2654435761u) ^ (cz * 805459861u);
    return index % hsize;
}
/* forward: x [N,3] in [0,1]  ->  y [N, L*F] (level-major pairs), F = 2 */
void xo_hashgrid_fwd(const float* params, const float* x, int n, int n_levels, const float* scale,
                     const uint32_t* resolution, const uint32_t* offset, float* y) {
#pragma omp parallel for num_threads(g_threads) if (g_threads > 1)
    for (int i = 0; i < n; ++i) {
        for (int l = 0; l < n_levels; ++l) {
            uint32_t hsize = offset[l + 1] - offset[l];
            const float* tab = params + 2 * (size_t)offset[l];
            float w[3]; uint32_t g[3];
            for (int d = 0; d < 3; ++d) {
                float p = x[3 * (size_t)i + d] * scale[l] + 0.5f;
                float f = floorf(p);
                g[d] = (uint32_t)(int)f; w[d] = p - f;
            }
            float r0 = 0.f, r1 = 0.f;
            for (int c = 0; c < 8; ++c) {
                float wt = 1.f; uint32_t q[3];
                for (int d = 0; d < 3; ++d) {
                    if ((c & (1 << d)) == 0) { wt *= 1.f - w[d]; q[d] = g[d]; }
                    else { wt *= w[d]; q[d] = g[d] + 1u; }
                }
                uint32_t idx = xo_grid_index(q[0], q[1], q[2], resolution[l], hsize);
                r0 += wt * tab[2 * (size_t)idx]; r1 += wt * tab[2 * (size_t)idx + 1];
            }
            y[(size_t)i * (2 * n_levels) + 2 * l] = r0;
            y[(size_t)i * (2 * n_levels) + 2 * l + 1] = r1;
        }
    }
}
/* backward: grad_params[idx,f] += w_corner * dy[i, l, f]  (accumulates; caller zero-fills).
 * No input gradient (the reference detaches pts, hashnerf_mlp.py:58-59). */
void xo_hashgrid_bwd(const float* x, const float* dy, int n, int n_levels, const float* scale,
                     const uint32_t* resolution, const uint32_t* offset, float* grad_params) {
    for (int i = 0; i < n; ++i) {
        for (int l = 0; l < n_levels; ++l) {
            uint32_t hsize = offset[l + 1] - offset[l];
            float* tab = grad_params + 2 * (size_t)offset[l];
            float w[3]; uint32_t g[3];
            for (int d = 0; d < 3; ++d) {
                float p = x[3 * (size_t)i + d] * scale[l] + 0.5f;
                float f = floorf(p);
                g[d] = (uint32_t)(int)f; w[d] = p - f;
            }
            float d0 = dy[(size_t)i * (2 * n_levels) + 2 * l], d1 = dy[(size_t)i * (2 * n_levels) + 2 * l + 1];
            for (int c = 0; c < 8; ++c) {
                float wt = 1.f; uint32_t q[3];
                for (int d = 0; d < 3; ++d) {
                    if ((c & (1 << d)) == 0) { wt *= 1.f - w[d]; q[d] = g[d]; }
                    else { wt *= w[d]; q[d] = g[d] + 1u; }
                }
                uint32_t idx = xo_grid_index(q[0], q[1], q[2], resolution[l], hsize);
                tab[2 * (size_t)idx] += wt * d0; tab[2 * (size_t)idx + 1] += wt * d1;
            }
        }
    }
}

/* SH degree 4 on d' = 2*x - 1 (x is the reference's warp_direction output in [0,1]) -> 16 values */
void xo_sh4(const float* dirs, int n, float* out) {
#pragma omp parallel for num_threads(g_threads) if (g_threads > 1)
    for (int i = 0; i < n; ++i) {
        float x = dirs[3 * (size_t)i] * 2.f - 1.f, y = dirs[3 * (size_t)i + 1] * 2.f - 1.f, z = dirs[3 * (size_t)i + 2] * 2.f - 1.f;
        float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
        float* o = out + 16 * (size_t)i;
        o[0] = 0.28209479177387814f;
        o[1] = -0.48860251190291987f * y;
        o[2] = 0.48860251190291987f * z;
        o[3] = -0.48860251190291987f * x;
        o[4] = 1.0925484305920792f * xy;
        o[5] = -1.0925484305920792f * yz;
        o[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
        o[7] = -1.0925484305920792f * xz;
        o[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
        o[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
        o[10] = 2.8906114426405538f * xy * z;
        o[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
        o[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
        o[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
        o[14] = 1.4453057213202769f * z * (x2 - y2);
        o[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
    }
}

/* Bias-free MLP, ReLU hidden, linear output. Weights are one flat buffer of row-major
 * [out,in] matrices in layer order: W0 [width, in_pad], (n_hidden-1) x [width,width],
 * Wlast [out_pad, width].  in_pad/out_pad are multiples of 16.  x is [N,in_pad], y [N,out_pad].
 * If `acts` != NULL it receives the post-ReLU hidden activations [n_hidden][N][width]. */
void xo_mlp_fwd(const float* W, const float* x, int n, int in_pad, int width, int n_hidden, int out_pad,
                float* y, float* acts) {
#pragma omp parallel for num_threads(g_threads) if (g_threads > 1)
    for (int i = 0; i < n; ++i) {
        float a[2][256];
        const float* in = x + (size_t)i * in_pad; int nin = in_pad; const float* w = W; int cur = 0;
        for (int l = 0; l < n_hidden; ++l) {
            for (int o = 0; o < width; ++o) {
                float s = 0.f;
                for (int k = 0; k < nin; ++k) s += w[(size_t)o * nin + k] * in[k];
                a[cur][o] = s > 0.f ? s : 0.f;
            }
            if (acts) memcpy(acts + ((size_t)l * n + i) * width, a[cur], sizeof(float) * width);
            w += (size_t)width * nin; in = a[cur]; nin = width; cur ^= 1;
        }
        for (int o = 0; o < out_pad; ++o) {
            float s = 0.f;
            for (int k = 0; k < nin; ++k) s += w[(size_t)o * nin + k] * in[k];
            y[(size_t)i * out_pad + o] = s;
        }
    }
}
/* backward: given x, acts (from fwd) and dy [N,out_pad]: dW (accumulated; caller zero-fills)
 * and dx [N,in_pad] (may be NULL). Serial => deterministic summation order over samples. */
void xo_mlp_bwd(const float* W, const float* x, const float* acts, const float* dy, int n, int in_pad,
                int width, int n_hidden, int out_pad, float* dW, float* dx) {
    size_t woff[16]; size_t off = 0;
    for (int l = 0; l <= n_hidden; ++l) {
        woff[l] = off;
        int nin = l == 0 ? in_pad : width, nout = l == n_hidden ? out_pad : width;
        off += (size_t)nin * nout;
    }
    for (int i = 0; i < n; ++i) {
        float g[2][256]; int cur = 0;
        /* output layer */
        const float* hin = acts + ((size_t)(n_hidden - 1) * n + i) * width;
        const float* wl = W + woff[n_hidden]; float* dwl = dW + woff[n_hidden];
        for (int k = 0; k < width; ++k) g[cur][k] = 0.f;
        for (int o = 0; o < out_pad; ++o) {
            float d = dy[(size_t)i * out_pad + o];
            for (int k = 0; k < width; ++k) { dwl[(size_t)o * width + k] += d * hin[k]; g[cur][k] += wl[(size_t)o * width + k] * d; }
        }
        for (int l = n_hidden - 1; l >= 0; --l) {
            const float* hout = acts + ((size_t)l * n + i) * width;
            for (int k = 0; k < width; ++k) if (!(hout[k] > 0.f)) g[cur][k] = 0.f;   /* ReLU' */
            int nin = l == 0 ? in_pad : width;
            const float* in = l == 0 ? x + (size_t)i * in_pad : acts + ((size_t)(l - 1) * n + i) * width;
            const float* w = W + woff[l]; float* dw = dW + woff[l];
            float* gn = g[cur ^ 1];
            for (int k = 0; k < nin; ++k) gn[k] = 0.f;
            for (int o = 0; o < width; ++o) {
                float d = g[cur][o];
                for (int k = 0; k < nin; ++k) { dw[(size_t)o * nin + k] += d * in[k]; gn[k] += w[(size_t)o * nin + k] * d; }
            }
            cur ^= 1;
        }
        if (dx) memcpy(dx + (size_t)i * in_pad, g[cur], sizeof(float) * in_pad);
    }
}

/* HashNerfMLP.run_mlp (xrnerf/models/mlps/hashnerf_mlp.py:55-79): pts [N,3], dirs [N,3] (both the
 * sampler's warped values in [0,1]) -> raw [N,4] = [r,g,b,sigma] pre-activation.
 * density_net: 32 -> n_hidden_d x 64 -> 16 ; color_net: cat(density_out[1:16], sh16) = 31, padded
 * to 32 with `pad_value` (tcnn pads the Identity-encoded input with 1.0) -> n_hidden_c x 64 -> 16. */
void xo_nerf_mlp_fwd(const float* table, const float* Wd, const float* Wc, const float* pts, const float* dirs,
                     int n, int n_levels, const float* scale, const uint32_t* resolution, const uint32_t* offset,
                     int n_hidden_d, int n_hidden_c, float pad_value, float* raw) {
    const int B = 4096;
    float* enc = (float*)malloc(sizeof(float) * B * 32);
    float* sh = (float*)malloc(sizeof(float) * B * 16);
    float* dout = (float*)malloc(sizeof(float) * B * 16);
    float* cin = (float*)malloc(sizeof(float) * B * 32);
    float* cout = (float*)malloc(sizeof(float) * B * 16);
    for (int s = 0; s < n; s += B) {
        int m = n - s < B ? n - s : B;
        xo_hashgrid_fwd(table, pts + 3 * (size_t)s, m, n_levels, scale, resolution, offset, enc);
        xo_mlp_fwd(Wd, enc, m, 2 * n_levels, 64, n_hidden_d, 16, dout, NULL);
        if (dirs) {
            xo_sh4(dirs + 3 * (size_t)s, m, sh);
            for (int i = 0; i < m; ++i) {
                for (int k = 0; k < 15; ++k) cin[i * 32 + k] = dout[i * 16 + 1 + k];
                for (int k = 0; k < 16; ++k) cin[i * 32 + 15 + k] = sh[i * 16 + k];
                cin[i * 32 + 31] = pad_value;
            }
            xo_mlp_fwd(Wc, cin, m, 32, 64, n_hidden_c, 16, cout, NULL);
        }
        for (int i = 0; i < m; ++i) {
            float* r = raw + 4 * (size_t)(s + i);
            if (dirs) { r[0] = cout[i * 16]; r[1] = cout[i * 16 + 1]; r[2] = cout[i * 16 + 2]; }
            else { r[0] = r[1] = r[2] = 0.f; }
            r[3] = dout[i * 16];
        }
    }
    free(enc); free(sh); free(dout); free(cin); free(cout);
}
/* Test infrastructure for the gradient tests: how close does a sample come to a ReLU kink?  margin[i] = min over the hidden
 * units of this network of |z| / sum_k |w_k x_k| (z = the unit's pre-activation, evaluated as xo_mlp_fwd does), MIN-merged into
 * margin (caller fills it with a large value).  A backward that recomputes z in another arithmetic can only disagree with this
 * one about relu'(z) on samples whose margin is below that arithmetic's relative error. */
void xo_mlp_kink_margin(const float* W, const float* x, int n, int in_pad, int width, int n_hidden, float* margin) {
    for (int i = 0; i < n; ++i) {
        float a[2][256];
        const float* in = x + (size_t)i * in_pad; int nin = in_pad; const float* w = W; int cur = 0;
        for (int l = 0; l < n_hidden; ++l) {
            for (int o = 0; o < width; ++o) {
                float s = 0.f; double mag = 0.0;
                for (int k = 0; k < nin; ++k) { s += w[(size_t)o * nin + k] * in[k]; mag += fabs((double)w[(size_t)o * nin + k] * (double)in[k]); }
                a[cur][o] = s > 0.f ? s : 0.f;
                if (mag > 0.0) { const float r = (float)(fabs((double)s) / mag); if (r < margin[i]) margin[i] = r; }
            }
            w += (size_t)width * nin; in = a[cur]; nin = width; cur ^= 1;
        }
    }
}
void xo_nerf_mlp_kink_margin(const float* table, const float* Wd, const float* Wc, const float* pts, const float* dirs,
                             int n, int n_levels, const float* scale, const uint32_t* resolution, const uint32_t* offset,
                             int n_hidden_d, int n_hidden_c, float pad_value, float* margin) {
    float* enc = (float*)malloc(sizeof(float) * 32);
    float sh[16], dout[16], cin[32];
    for (int i = 0; i < n; ++i) {
        margin[i] = 1e30f;
        xo_hashgrid_fwd(table, pts + 3 * (size_t)i, 1, n_levels, scale, resolution, offset, enc);
        xo_mlp_kink_margin(Wd, enc, 1, 2 * n_levels, 64, n_hidden_d, margin + i);
        xo_mlp_fwd(Wd, enc, 1, 2 * n_levels, 64, n_hidden_d, 16, dout, NULL);
        xo_sh4(dirs + 3 * (size_t)i, 1, sh);
        for (int k = 0; k < 15; ++k) cin[k] = dout[1 + k];
        for (int k = 0; k < 16; ++k) cin[15 + k] = sh[k];
        cin[31] = pad_value;
        xo_mlp_kink_margin(Wc, cin, 1, 32, 64, n_hidden_c, margin + i);
    }
    free(enc);
}
/* backward of the above w.r.t. table, Wd, Wc given dL/draw [N,4]; gradients ACCUMULATE. */
void xo_nerf_mlp_bwd(const float* table, const float* Wd, const float* Wc, const float* pts, const float* dirs,
                     const float* draw, int n, int n_levels, const float* scale, const uint32_t* resolution,
                     const uint32_t* offset, int n_hidden_d, int n_hidden_c, float pad_value,
                     float* g_table, float* g_Wd, float* g_Wc) {
    const int B = 4096;
    float* enc = (float*)malloc(sizeof(float) * B * 32);
    float* sh = (float*)malloc(sizeof(float) * B * 16);
    float* dout = (float*)malloc(sizeof(float) * B * 16);
    float* cin = (float*)malloc(sizeof(float) * B * 32);
    float* cout = (float*)malloc(sizeof(float) * B * 16);
    float* actd = (float*)malloc(sizeof(float) * B * 64 * (n_hidden_d > 0 ? n_hidden_d : 1));
    float* actc = (float*)malloc(sizeof(float) * B * 64 * (n_hidden_c > 0 ? n_hidden_c : 1));
    float* dcout = (float*)malloc(sizeof(float) * B * 16);
    float* dcin = (float*)malloc(sizeof(float) * B * 32);
    float* ddout = (float*)malloc(sizeof(float) * B * 16);
    float* denc = (float*)malloc(sizeof(float) * B * 32);
    for (int s = 0; s < n; s += B) {
        int m = n - s < B ? n - s : B;
        xo_hashgrid_fwd(table, pts + 3 * (size_t)s, m, n_levels, scale, resolution, offset, enc);
        xo_mlp_fwd(Wd, enc, m, 2 * n_levels, 64, n_hidden_d, 16, dout, actd);
        xo_sh4(dirs + 3 * (size_t)s, m, sh);
        for (int i = 0; i < m; ++i) {
            for (int k = 0; k < 15; ++k) cin[i * 32 + k] = dout[i * 16 + 1 + k];
            for (int k = 0; k < 16; ++k) cin[i * 32 + 15 + k] = sh[i * 16 + k];
            cin[i * 32 + 31] = pad_value;
        }
        xo_mlp_fwd(Wc, cin, m, 32, 64, n_hidden_c, 16, cout, actc);
        for (int i = 0; i < m; ++i) {
            const float* d = draw + 4 * (size_t)(s + i);
            for (int k = 0; k < 16; ++k) dcout[i * 16 + k] = k < 3 ? d[k] : 0.f;
        }
        xo_mlp_bwd(Wc, cin, actc, dcout, m, 32, 64, n_hidden_c, 16, g_Wc, dcin);
        for (int i = 0; i < m; ++i) {
            ddout[i * 16] = draw[4 * (size_t)(s + i) + 3];
            for (int k = 0; k < 15; ++k) ddout[i * 16 + 1 + k] = dcin[i * 32 + k];
        }
        xo_mlp_bwd(Wd, enc, actd, ddout, m, 2 * n_levels, 64, n_hidden_d, 16, g_Wd, denc);
        xo_hashgrid_bwd(pts + 3 * (size_t)s, denc, m, n_levels, scale, resolution, offset, g_table);
    }
    free(enc); free(sh); free(dout); free(cin); free(cout); free(actd); free(actc);
    free(dcout); free(dcin); free(ddout); free(denc);
}

/* ---------------------------------------------------------------- ray generation
 * get_rays_np_hash (xrnerf/datasets/load_data/get_rays.py:35-69), arithmetic pinned to fp32:
 * pixel centre (+0.5), dir = ((i-cx)/fx, (j-cy)/fy, 1), d = R*dir (row k of R = column k of the
 * python [4,3] pose), normalise; o = translation row. pose is the python row-major [4,3]. */
void xo_gen_rays(const float* pose43, int H, int W, float fx, float fy, float cx, float cy,
                 int row0, int nrows, float* rays_o, float* rays_d) {
    for (int r = 0; r < nrows; ++r) for (int c = 0; c < W; ++c) {
        float i = (float)c + 0.5f, j = (float)(row0 + r) + 0.5f;
        float dx = (i - cx) / fx, dy = (j - cy) / fy, dz = 1.0f;
        float v[3];
        for (int k = 0; k < 3; ++k) {      /* c2w = pose^T ; c2w[k][m] = pose[m][k] */
            v[k] = pose43[0 * 3 + k] * dx + pose43[1 * 3 + k] * dy;
            v[k] = v[k] + pose43[2 * 3 + k] * dz;
        }
        float nrm = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        size_t q = (size_t)r * W + c;
        for (int k = 0; k < 3; ++k) { rays_d[3 * q + k] = v[k] / nrm; rays_o[3 * q + k] = pose43[3 * 3 + k]; }
    }
}

/* ---------------------------------------------------------------- loss
 * HuberLoss (xrnerf/models/networks/utils/metrics.py:8-16): rel=|x-y|; rel > delta ? rel - delta/2
 * : 0.5/delta*rel^2, summed; times 5 in train_step (networks/hashnerf.py:37-44). Returns dL/drgb. */
float xo_huber_loss_grad(const float* rgb, const float* target, int n3, float delta, float scale, float* grad) {
    double loss = 0;
    for (int i = 0; i < n3; ++i) {
        float r = rgb[i] - target[i], a = fabsf(r);
        if (a > delta) { loss += a - 0.5f * delta; grad[i] = scale * (r > 0 ? 1.f : -1.f); }
        else { loss += 0.5f / delta * a * a; grad[i] = scale * (r / delta); }
    }
    return (float)(loss * scale);
}

/* ---------------------------------------------------------------- optimiser
 * torch.optim.Adam semantics (configs/instant_ngp/nerf_blender_local01.py:14-18: lr 1e-2,
 * betas (0.9,0.99), eps 1e-15, weight_decay 1e-6 as L2 added to the gradient). */
void xo_adam(float* p, const float* g, float* m, float* v, size_t n, int step, float lr, float b1, float b2,
             float eps, float wd) {
    float bc1 = 1.f - powf(b1, (float)step), bc2 = 1.f - powf(b2, (float)step);
    float step_size = lr / bc1, bc2s = sqrtf(bc2);
    for (size_t i = 0; i < n; ++i) {
        float gi = g[i] + wd * p[i];
        m[i] = b1 * m[i] + (1.f - b1) * gi;
        v[i] = b2 * v[i] + (1.f - b2) * gi * gi;
        float denom = sqrtf(v[i]) / bc2s + eps;
        p[i] = p[i] - step_size * (m[i] / denom);
    }
}
