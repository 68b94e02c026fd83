// K1 (occupancy-grid ray march), K2 (re-pack), K3/K4/K5 (compositor fwd / bwd / inference).
// Hand-written for gfx950 (wave64).  This translation unit is compiled with -ffp-contract=off:
// K1's index decisions hinge on `o + t*d` being mul-then-add exactly like the reference's
// CPU-compiled kernels (the only runnable reference; SURVEY.md section 7 "Hard parts").
//
// Reference: /root/reference/extensions/ngp_raymarch/src/{ray_sampler,compacted_coord,calc_rgb}.cu
// and include/{ray_sampler_header,raymarch_shared}.h (line cites inline).
#include "xr_common.h"
#include "xr_aux.h"
#include <cfloat>
#include <cstdlib>

#define RM_BLOCK 256
#ifndef K1W_MAX_RAYS
#define K1W_MAX_RAYS 32768  // XR_K1_WIDE launches of up to this many rays march with 8 lanes per ray (k1_count_w); 0: always one ray per lane
#endif
#define K1_TL 128         // per-ray t-list capacity (4*n_rays*K1_TL bytes of workspace); longer rays -- 3 of 12.5 K in a
                          // training batch, 9 % at 64 -- fall back to a re-march of their tail in the write pass
                          // (measured: K1 223 -> 172 us on 12.5 K rays)

// ------------------------------------------------------------------ block-wide exclusive scan
// 256 threads = 4 wave64.  Returns the exclusive prefix of v; *total gets the block sum.
__device__ inline uint32_t block_excl_scan(uint32_t v, uint32_t* total, uint32_t* lds4) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) lds4[wid] = inc;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < RM_BLOCK / 64; ++w) {
        uint32_t t = lds4[w];
        if (w < wid) woff += t;
        tot += t;
    }
    __syncthreads();
    *total = tot;
    return woff + inc - v;
}

// ------------------------------------------------------------------ march helpers
struct Ray { float ox, oy, oz, dx, dy, dz, ix, iy, iz; };

__device__ inline float rm_calc_dt(float t, float cone) {           // ray_sampler_header.h:24-25
    return xr_clampf(t * cone, xr_min_step(), xr_max_step());
}
__device__ inline int rm_mip_from_pos(float px, float py, float pz) { // :37-43
    float m = fabsf(px - 0.5f);
    float b = fabsf(py - 0.5f); if (b > m) m = b;
    float c = fabsf(pz - 0.5f); if (c > m) m = c;
    int e; frexpf(m, &e);
    return min(7, max(0, e + 1));
}
__device__ inline int rm_mip_from_dt(float dt, float px, float py, float pz) { // :45-54
    int mip = rm_mip_from_pos(px, py, pz);
    dt *= 256.0f;
    if (dt < 1.f) return mip;
    int e; frexpf(dt, &e);
    return min(7, max(e, mip));
}
__device__ inline bool rm_occupied(float px, float py, float pz, const uint8_t* __restrict__ bf, int mip) { // :298-319
    float s = scalbnf(1.0f, -mip);
    float qx = (px - 0.5f) * s + 0.5f, qy = (py - 0.5f) * s + 0.5f, qz = (pz - 0.5f) * s + 0.5f;
    int ix = (int)(qx * 128.0f), iy = (int)(qy * 128.0f), iz = (int)(qz * 128.0f);
    ix = min(max(ix, 0), 127); iy = min(max(iy, 0), 127); iz = min(max(iz, 0), 127);
    uint32_t idx = xr_morton3d((uint32_t)ix, (uint32_t)iy, (uint32_t)iz);
    return (bf[(idx >> 3) + (uint32_t)mip * (XR_GRID_CELLS / 8)] >> (idx & 7)) & 1;
}
__device__ inline float rm_lt_min(float a, float b) { return a < b ? a : b; }  // reference `min` = a<b?a:b
__device__ inline float rm_advance_target(float t, float px, float py, float pz, const Ray& r, uint32_t res) { // :271-293
    float rf = (float)res;
    float ppx = rf * px, ppy = rf * py, ppz = rf * pz;
    float tx = (floorf(ppx + 0.5f + 0.5f * copysignf(1.0f, r.dx)) - ppx) * r.ix;
    float ty = (floorf(ppy + 0.5f + 0.5f * copysignf(1.0f, r.dy)) - ppy) * r.iy;
    float tz = (floorf(ppz + 0.5f + 0.5f * copysignf(1.0f, r.dz)) - ppz) * r.iz;
    float tt = rm_lt_min(rm_lt_min(tx, ty), tz);
    return t + fmaxf(tt / rf, 0.0f);
}
__device__ inline float rm_advance(float t, float cone, float px, float py, float pz, const Ray& r, uint32_t res) { // :271-296
    const float target = rm_advance_target(t, px, py, pz, r, res);
    do { t += rm_calc_dt(t, cone); } while (t < target);
    return t;
}
__device__ inline float rm_aabb_tmin(float lo, float hi, const Ray& r) {   // raymarch_shared.h:506-563
    float tmin = (lo - r.ox) / r.dx, tmax = (hi - r.ox) / r.dx;
    if (tmin > tmax) { float s = tmin; tmin = tmax; tmax = s; }
    float tymin = (lo - r.oy) / r.dy, tymax = (hi - r.oy) / r.dy;
    if (tymin > tymax) { float s = tymin; tymin = tymax; tymax = s; }
    if (tmin > tymax || tymin > tmax) return FLT_MAX;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    float tzmin = (lo - r.oz) / r.dz, tzmax = (hi - r.oz) / r.dz;
    if (tzmin > tzmax) { float s = tzmin; tzmin = tzmax; tzmax = s; }
    if (tmin > tzmax || tzmin > tmax) return FLT_MAX;
    if (tzmin > tmin) tmin = tzmin;
    return tmin;
}
__device__ inline bool rm_contains(float lo, float hi, float px, float py, float pz) { // :570-575
    return px >= lo && px <= hi && py >= lo && py <= hi && pz >= lo && pz <= hi;
}
__device__ inline Ray rm_load_ray(const float* __restrict__ o, const float* __restrict__ d, uint32_t i) {
    Ray r;
    r.ox = o[3 * i]; r.oy = o[3 * i + 1]; r.oz = o[3 * i + 2];
    r.dx = d[3 * i]; r.dy = d[3 * i + 1]; r.dz = d[3 * i + 2];
    r.ix = 1.0f / r.dx; r.iy = 1.0f / r.dy; r.iz = 1.0f / r.dz;
    return r;
}

// ------------------------------------------------------------------ K1 pass A: count (ray_sampler.cu:26-74)
// One ray per lane: a chain of dependent bitfield-byte loads (~0.6 us each from L2) plus ~150 VALU instructions per
// visited lattice point.  The arithmetic is the bound, not the load: keeping the last 32-bit word of the bitfield in a
// register (32 consecutive Morton indices = one 4x4x2 block of cells, so most steps re-use it and skip the load) made the
// pass 5 % SLOWER (12.5 K rays: 176 -> 184 us, profiles/r02_k1_word_cache_ab.txt) -- the compare / branch per step costs
// more than the loads it saves.  Measured dead end: marching 4 rays per lane in lock step (4 loads in flight per lane)
// is 3.5x SLOWER (172 -> 612 us on 12.5 K rays) -- the per-point arithmetic at 4 cycles per wave64 instruction
// then dominates and a wave lasts as long as the longest of 256 rays instead of 64.
// Round 3, also measured and removed: one WAVE per ray.  The lattice t_0, t_1, ... of a ray is a chain of `t += calc_dt(t)` that
// does not depend on the occupancy (samples and the do-while of advance_to_next_voxel step along the same chain), so 64 lanes can
// evaluate 64 consecutive lattice points at once and replay the reference's control flow with ballots -- bit-exact (it passed the
// K1 tests here and on the GPU).  But every lane has to walk the chain to its own point with the same fp32 additions (63 dependent
// steps per window, ~2000 of the ~3000 issue cycles of a window), 12.5 K waves x ~8 windows make it as long as this kernel
// (173 us either way, tools/microbench_k1.py) while occupying every SIMD: beside the training step on the side stream it slowed
// the encode from 81 to 143 us and the iteration from 0.447 to 0.488 ms (profiles/r03_k1_wave_per_ray_ab.txt).
// A SERIES of launches in one (round 5): blockIdx.y = launch c of a series of gridDim.y launches over n_rays rays each -- the marches of
// a whole refresh window (ngp_grid_sampler.py:194-197,268-281: the bitfield and the batch size do not change between two refreshes and
// K1 reads no weights).  Launch c reads rays c * ray_stride .., draws the jitter of hidden-generator call index (first + c) -- the
// generator moves on by 2^32 per launch, ray_sampler.cu:198 -- and owns workspace blocks [c * gridDim.x, (c + 1) * gridDim.x).
__global__ __launch_bounds__(RM_BLOCK) void k1_count(
    uint32_t n_rays, float lo, float hi, const float* __restrict__ rays_o, const float* __restrict__ rays_d,
    const uint8_t* __restrict__ bitfield, float cone, float near_distance, xr_pcg32 rng, uint32_t rng_chunk, uint32_t rng_ray0,
    uint32_t* __restrict__ cnt, uint32_t* __restrict__ local_off, float* __restrict__ start_t,
    uint32_t* __restrict__ block_tot, float* __restrict__ tlist, uint32_t ray_stride) {
    __shared__ uint32_t lds4[4];
    const uint32_t i = blockIdx.x * RM_BLOCK + threadIdx.x;                     // ray of this launch
    const uint32_t wb = blockIdx.y * gridDim.x + blockIdx.x, wi = wb * RM_BLOCK + threadIdx.x;   // workspace block / slot
    rays_o += 3 * (size_t)blockIdx.y * ray_stride; rays_d += 3 * (size_t)blockIdx.y * ray_stride;
    rng.advance((uint64_t)blockIdx.y << 32);
    uint32_t j = 0; float startt = 0.f;
    if (i < n_rays) {
        // :31.  rng_chunk > 0: ray i draws what it would draw as ray i % rng_chunk of launch i / rng_chunk of a series of
        // launches over rng_chunk rays each (the generator moves on by 2^32 per launch, :198) -- a frame the reference marches
        // in chunk-sized launches (networks/nerf.py:50-69) becomes ONE launch with the same samples
        // rng_ray0: this launch holds rays rng_ray0 .. of that series' frame (a rank's band of image rows)
        if (rng_chunk) { const uint32_t g = rng_ray0 + i; rng.advance(((uint64_t)(g / rng_chunk) << 32) + (uint64_t)((g % rng_chunk) * 8u)); }
        else rng.advance((uint64_t)(i * 8u));
        Ray r = rm_load_ray(rays_o, rays_d, i);
        float tmin = fmaxf(rm_aabb_tmin(lo, hi, r), near_distance);          // :42-46
        startt = tmin;
        startt += rm_calc_dt(startt, cone) * rng.next_float();               // :50
        float t = startt;
        for (;;) {                                                           // :58-72
            float px = r.ox + t * r.dx, py = r.oy + t * r.dy, pz = r.oz + t * r.dz;
            if (!(rm_contains(lo, hi, px, py, pz) && j < XR_NERF_STEPS)) break;
            float dt = rm_calc_dt(t, cone);
            int mip = rm_mip_from_dt(dt, px, py, pz);
            if (rm_occupied(px, py, pz, bitfield, mip)) {
                // a sample is fully determined by its t: remember the first K1_TL of them so that the
                // write pass can expand them sample-parallel instead of re-marching
                if (j < K1_TL) tlist[(size_t)wi * K1_TL + j] = t;
                ++j; t += dt;
            } else t = rm_advance(t, cone, px, py, pz, r, XR_NERF_GRIDSIZE >> mip);
        }
    }
    uint32_t tot;
    uint32_t off = block_excl_scan(j, &tot, lds4);
    if (i < n_rays) { cnt[wi] = j; local_off[wi] = off; start_t[wi] = startt; }
    if (threadIdx.x == 0) block_tot[wb] = tot;
}

// ------------------------------------------------------------------ K1 pass A with K1W lanes per ray (training-sized batches)
// A training batch is ~12.8 K rays: one ray per lane makes 200 waves on 1024 SIMDs, each a serial chain of ~150 VALU instructions per
// visited lattice point -- 172-222 us however idle the chip is.  The lattice t_0, t_1 = t_0 + calc_dt(t_0), ... does not depend on
// the occupancy (a sample steps `t += dt`, advance_to_next_voxel steps `do t += calc_dt(t) while (t < target)`: the same chain), so
// the K1W lanes of a ray evaluate K1W consecutive lattice points at once -- lane i reaches its point with i of those additions, the
// same fp32 operations in the same order as the serial walk -- and then replay the reference's control flow over the window with
// group ballots / shuffles: a tested point is either a sample (next point: the following one) or empty (next point: the first one
// not below its voxel-exit target, possibly in a later window).  Points the walk jumps over were evaluated for nothing; with 8
// lanes a window costs ~7 chain steps + one point evaluation + <= 8 replay steps instead of 8 point evaluations in a row.
// (Round 3 tried 64 lanes per ray: 63 chain steps per window and 12.5 K waves made it no faster than the serial kernel.)
// Same counts, same sample t's (bit for bit), same RNG draw; cnt / start_t / tlist as k1_count, the block scan is k1_block_scan.
#ifndef K1W
#define K1W 8
#endif
__global__ __launch_bounds__(RM_BLOCK) void k1_count_w(
    uint32_t n_rays, float lo, float hi, const float* __restrict__ rays_o, const float* __restrict__ rays_d,
    const uint8_t* __restrict__ bitfield, float cone, float near_distance, xr_pcg32 rng, uint32_t rng_chunk, uint32_t rng_ray0,
    uint32_t* __restrict__ cnt, float* __restrict__ start_t, float* __restrict__ tlist) {
    const uint32_t ray = (blockIdx.x * RM_BLOCK + threadIdx.x) / K1W;
    const int li = threadIdx.x & (K1W - 1), g0 = (threadIdx.x & 63) & ~(K1W - 1);      // lane inside its ray's group, the group's first lane of the wave
    if (ray >= n_rays) return;                                                          // (whole groups leave)
    if (rng_chunk) { const uint32_t g = rng_ray0 + ray; rng.advance(((uint64_t)(g / rng_chunk) << 32) + (uint64_t)((g % rng_chunk) * 8u)); }
    else rng.advance((uint64_t)(ray * 8u));
    const Ray r = rm_load_ray(rays_o, rays_d, ray);
    float startt = fmaxf(rm_aabb_tmin(lo, hi, r), near_distance);
    startt += rm_calc_dt(startt, cone) * rng.next_float();
    float t0 = startt, target = 0.f;
    uint32_t j = 0;
    bool seek = false;                                   // the walk is inside a voxel advance: the next tested point is the first one not below `target`
    for (;;) {
        float t = t0;
#pragma unroll
        for (int s = 0; s < K1W - 1; ++s)
            if (s < li) t += rm_calc_dt(t, cone);
        const float px = r.ox + t * r.dx, py = r.oy + t * r.dy, pz = r.oz + t * r.dz;
        const bool inb = rm_contains(lo, hi, px, py, pz);
        bool occ = false;
        float tgt = 0.f;
        if (inb) {
            const float dt = rm_calc_dt(t, cone);
            const int mip = rm_mip_from_dt(dt, px, py, pz);
            occ = rm_occupied(px, py, pz, bitfield, mip);
            if (!occ) tgt = rm_advance_target(t, px, py, pz, r, XR_NERF_GRIDSIZE >> mip);
        }
        // ---- replay of the reference's walk over the window (every lane of the group takes the same decisions)
        uint32_t cur = 0;
        if (seek) {                                      // do { t += dt } while (t < target): the first lattice point with !(t < target)
            const uint32_t m = (uint32_t)(__ballot(!(t < target)) >> g0) & ((1u << K1W) - 1u);
            cur = m ? (uint32_t)__ffs((int)m) - 1u : (uint32_t)K1W;
        }
        bool done = false;
        while (cur < (uint32_t)K1W) {
            const int c_inb = __shfl((int)inb, (int)cur, K1W);
            if (!(c_inb && j < XR_NERF_STEPS)) { done = true; break; }
            seek = false;
            if (__shfl((int)occ, (int)cur, K1W)) {
                if ((uint32_t)li == cur && j < K1_TL) tlist[(size_t)ray * K1_TL + j] = t;
                ++j; ++cur;
            } else {
                const float c_tgt = __shfl(tgt, (int)cur, K1W);
                const uint32_t m = ((uint32_t)(__ballot(!(t < c_tgt)) >> g0) & ((1u << K1W) - 1u)) & ~((2u << cur) - 1u);   // lanes behind `cur`
                if (m) cur = (uint32_t)__ffs((int)m) - 1u;
                else { seek = true; target = c_tgt; cur = (uint32_t)K1W; }
            }
        }
        if (done) break;
        const float tl = __shfl(t, K1W - 1, K1W);
        t0 = tl + rm_calc_dt(tl, cone);
    }
    if (li == 0) { cnt[ray] = j; start_t[ray] = startt; }
}
// per 256-ray block: exclusive prefix of the counts (ray order) and the block total -- what k1_count produces on the fly
__global__ __launch_bounds__(RM_BLOCK) void k1_block_scan(uint32_t n_rays, const uint32_t* __restrict__ cnt, uint32_t* __restrict__ local_off,
                                                          uint32_t* __restrict__ block_tot) {
    __shared__ uint32_t lds4[4];
    const uint32_t i = blockIdx.x * RM_BLOCK + threadIdx.x;
    uint32_t tot;
    const uint32_t off = block_excl_scan(i < n_rays ? cnt[i] : 0u, &tot, lds4);
    if (i < n_rays) local_off[i] = off;
    if (threadIdx.x == 0) block_tot[blockIdx.x] = tot;
}

// ------------------------------------------------------------------ block-total scan (single block)
// block_base[b] = sum_{b'<b} block_tot[b'];  info[0] = grand total, info[1] = first block whose
// end exceeds `limit` (or nb if none).
__global__ __launch_bounds__(1024) void k_scan_blocks(uint32_t nb, const uint32_t* __restrict__ block_tot,
                                                      uint32_t limit, uint32_t* __restrict__ block_base,
                                                      uint32_t* __restrict__ info) {
    __shared__ uint32_t wsum[16];
    __shared__ uint32_t s_running, s_cross;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (threadIdx.x == 0) { s_running = 0; s_cross = nb; }
    __syncthreads();
    for (uint32_t c0 = 0; c0 < nb; c0 += 1024) {
        uint32_t b = c0 + threadIdx.x;
        uint32_t v = b < nb ? block_tot[b] : 0, inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
        if (lane == 63) wsum[wid] = inc;
        __syncthreads();
        uint32_t woff = 0, ctot = 0;
        for (int w = 0; w < 16; ++w) { uint32_t t = wsum[w]; if (w < wid) woff += t; ctot += t; }
        uint32_t base = s_running + woff + inc - v;
        if (b < nb) {
            block_base[b] = base;
            if ((uint64_t)base + v > (uint64_t)limit) atomicMin(&s_cross, b);
        }
        __syncthreads();
        if (threadIdx.x == 0) s_running += ctot;
        __syncthreads();
    }
    if (threadIdx.x == 0) { info[0] = s_running; info[1] = s_cross; }
}

// ------------------------------------------------------------------ K1 pass B: write (ray_sampler.cu:75-115)
__global__ __launch_bounds__(RM_BLOCK) void k1_write(
    uint32_t n_rays, float lo, float hi, const float* __restrict__ rays_o, const float* __restrict__ rays_d,
    const uint8_t* __restrict__ bitfield, float cone, uint32_t max_samples, const uint32_t* __restrict__ cnt,
    const uint32_t* __restrict__ local_off, const float* __restrict__ start_t,
    const uint32_t* __restrict__ block_base, const uint32_t* __restrict__ info, const uint32_t* __restrict__ block_tot,
    float* __restrict__ coords_out,
    int32_t* __restrict__ rays_index, int32_t* __restrict__ numsteps_out, uint32_t* __restrict__ counter2,
    const float* __restrict__ tlist, float* __restrict__ xyz_planes, uint32_t plane_stride, uint32_t ray_stride, size_t coords_stride) {
    __shared__ uint32_t lds4[4];
    const uint32_t b = blockIdx.x, i = b * RM_BLOCK + threadIdx.x;
    const uint32_t nb = gridDim.x;
    // launch blockIdx.y of a series (see k1_count): its rays, its output buffers, its blocks of the workspace
    {
        const size_t c = blockIdx.y;
        rays_o += 3 * c * ray_stride; rays_d += 3 * c * ray_stride;
        rays_index += c * ray_stride; numsteps_out += 2 * c * ray_stride; counter2 += 2 * c;
        coords_out += 7 * c * coords_stride;
        if (xyz_planes) xyz_planes += 3 * c * (size_t)plane_stride;
        if (block_tot) block_tot += c * nb;
        cnt += c * nb * RM_BLOCK; local_off += c * nb * RM_BLOCK; start_t += c * nb * RM_BLOCK; tlist += c * nb * RM_BLOCK * (size_t)K1_TL;
    }
    uint32_t cross, bbase, grand;
    if (block_tot) {
        // up to RM_BLOCK blocks (65 536 rays: every training batch): each workgroup scans the block totals itself -- the one-workgroup
        // k_scan_blocks launch between the two passes is gone (5 us alone, 20-40 us of the side stream's chain beside the training
        // step, where a one-workgroup kernel waits for a slot: profiles/r04_trace_normal_iteration.txt)
        __shared__ uint32_t s_cross, s_bbase;
        if (threadIdx.x == 0) s_cross = nb;
        const uint32_t v = threadIdx.x < nb ? block_tot[threadIdx.x] : 0u;
        const uint32_t ex = block_excl_scan(v, &grand, lds4);
        if (threadIdx.x < nb && (uint64_t)ex + v > (uint64_t)max_samples) atomicMin(&s_cross, threadIdx.x);
        if (threadIdx.x == b) s_bbase = ex;
        __syncthreads();
        cross = s_cross; bbase = s_bbase;
    } else { cross = info[1]; bbase = block_base[b]; grand = info[0]; }
    uint32_t n = 0, base = 0; bool in = i < n_rays;
    if (in) { n = cnt[i]; base = bbase + local_off[i]; }
    // a ray is "valid" unless its range would overflow the sample buffer (:76-82)
    bool valid = in && !((uint64_t)base + n > (uint64_t)max_samples);
    // serial ray index = number of valid rays before this one (:86).  Blocks before the first
    // overflowing block are all-valid, blocks after it are all-invalid.
    uint32_t vtot;
    uint32_t voff = block_excl_scan(valid ? 1u : 0u, &vtot, lds4);
    uint32_t valid_before = (b <= cross) ? b * RM_BLOCK : 0u;  // b > cross: unused (nothing valid)
    if (in) {
        if (!valid) {
            numsteps_out[2 * i] = 0; numsteps_out[2 * i + 1] = (int32_t)base; rays_index[i] = 0;
        } else {
            numsteps_out[2 * i] = (int32_t)n; numsteps_out[2 * i + 1] = (int32_t)base;
            rays_index[i] = n == 0 ? -1 : (int32_t)(valid_before + voff);
        }
    }
    // counters: samples = grand total (the atomicAdd of :75 runs for every ray, overflowing or
    // not); rays = number of valid rays
    if (threadIdx.x == 0) {
        if (b == 0) counter2[1] = grand;
        if (b == cross || (cross == nb && b == nb - 1)) counter2[0] = valid_before + vtot;
    }
    // ---- sample-parallel expansion.  The block's rays own the contiguous row range
    // [bbase, bbase + block total): all 256 lanes walk it with a stride of 256, find the ray of each
    // row by binary search over the block's offsets (LDS) and expand (t -> 7 floats).  Consecutive
    // lanes write consecutive 28-B rows: the stores of a wave cover one contiguous 1.75 KB span.
    __shared__ uint32_t s_off[RM_BLOCK], s_n[RM_BLOCK];
    __shared__ float s_ray[RM_BLOCK][6];
    s_off[threadIdx.x] = in ? local_off[i] : 0xffffffffu;
    s_n[threadIdx.x] = valid ? n : 0u;
    if (in) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { s_ray[threadIdx.x][k] = rays_o[3 * i + k]; s_ray[threadIdx.x][3 + k] = rays_d[3 * i + k]; }
    }
    __syncthreads();
    const uint32_t last = min((uint32_t)RM_BLOCK, n_rays - b * RM_BLOCK) - 1;
    const uint32_t btot = s_off[last] + cnt[b * RM_BLOCK + last];
    const float diag = hi - lo;
    for (uint32_t e = threadIdx.x; e < btot; e += RM_BLOCK) {
        uint32_t lo_r = 0, hi_r = last;                       // last ray with s_off[r] <= e
        while (lo_r < hi_r) {
            const uint32_t mid = (lo_r + hi_r + 1) >> 1;
            if (s_off[mid] <= e) lo_r = mid; else hi_r = mid - 1;
        }
        const uint32_t j = e - s_off[lo_r];
        if (j >= s_n[lo_r] || j >= K1_TL) continue;           // dropped ray, or the tail of a very long ray
        const float t = tlist[((size_t)b * RM_BLOCK + lo_r) * K1_TL + j];
        const float ox = s_ray[lo_r][0], oy = s_ray[lo_r][1], oz = s_ray[lo_r][2];
        const float dx = s_ray[lo_r][3], dy = s_ray[lo_r][4], dz = s_ray[lo_r][5];
        const float px = ox + t * dx, py = oy + t * dy, pz = oz + t * dz;         // same expression as the march
        const float dt = rm_calc_dt(t, cone);
        float* c = coords_out + 7 * ((size_t)bbase + e);
        c[0] = (px - lo) / diag; c[1] = (py - lo) / diag; c[2] = (pz - lo) / diag;
        c[3] = (dt - xr_min_step()) / (xr_max_warp_step() - xr_min_step());       // warp_dt
        c[4] = (dx + 1.0f) * 0.5f; c[5] = (dy + 1.0f) * 0.5f; c[6] = (dz + 1.0f) * 0.5f;   // warp_direction
        if (xyz_planes) {        // the same positions once more as three planes: the encoder reads them with coalesced loads
            const size_t q = (size_t)bbase + e;
            xyz_planes[q] = c[0]; xyz_planes[plane_stride + q] = c[1]; xyz_planes[2 * (size_t)plane_stride + q] = c[2];
        }
    }
    if (!valid || n <= K1_TL) return;
    // tail of a ray with more than K1_TL samples: resume the march right after sample K1_TL-1
    // (t advances by calc_dt(t) after an emitted sample, ray_sampler.cu:107-108) and write rows >= K1_TL
    Ray r = rm_load_ray(rays_o, rays_d, i);
    const float wdx = (r.dx + 1.0f) * 0.5f, wdy = (r.dy + 1.0f) * 0.5f, wdz = (r.dz + 1.0f) * 0.5f;
    float t = tlist[(size_t)i * K1_TL + (K1_TL - 1)];
    t += rm_calc_dt(t, cone);
    uint32_t j = K1_TL;
    float* __restrict__ out = coords_out + 7 * (size_t)base;
    for (;;) {
        float px = r.ox + t * r.dx, py = r.oy + t * r.dy, pz = r.oz + t * r.dz;
        if (!(rm_contains(lo, hi, px, py, pz) && j < n)) break;
        float dt = rm_calc_dt(t, cone);
        int mip = rm_mip_from_dt(dt, px, py, pz);
        if (rm_occupied(px, py, pz, bitfield, mip)) {
            float* c = out + 7 * (size_t)j;
            c[0] = (px - lo) / diag; c[1] = (py - lo) / diag; c[2] = (pz - lo) / diag;
            c[3] = (dt - xr_min_step()) / (xr_max_warp_step() - xr_min_step());
            c[4] = wdx; c[5] = wdy; c[6] = wdz;
            if (xyz_planes) {
                const size_t q = (size_t)base + j;
                xyz_planes[q] = c[0]; xyz_planes[plane_stride + q] = c[1]; xyz_planes[2 * (size_t)plane_stride + q] = c[2];
            }
            ++j; t += dt;
        } else t = rm_advance(t, cone, px, py, pz, r, XR_NERF_GRIDSIZE >> mip);
    }
}

struct RmWorkspace { uint32_t *cnt, *local_off, *block_tot, *block_base, *info; float *start_t, *tlist; };
// `n_series` launches over n_rays rays each (xr_rays_sampler_series): every launch owns whole 256-ray blocks of the per-ray arrays
static size_t rm_ws_layout(uint32_t n_rays, char* base, RmWorkspace* w, uint32_t n_series = 1) {
    const size_t nb = (size_t)xr_div_up(n_rays, RM_BLOCK) * n_series;
    const size_t slots = n_series > 1 ? nb * RM_BLOCK : (size_t)n_rays;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) / 256 * 256; return base ? base + o : (char*)nullptr; };
    char* p0 = take(4ull * slots); char* p1 = take(4ull * slots); char* p2 = take(4ull * slots);
    char* p3 = take(4 * nb); char* p4 = take(4 * nb); char* p5 = take(16);
    char* p6 = take(4ull * slots * K1_TL);
    if (w) { w->tlist = (float*)p6; w->cnt = (uint32_t*)p0; w->local_off = (uint32_t*)p1; w->start_t = (float*)p2;
             w->block_tot = (uint32_t*)p3; w->block_base = (uint32_t*)p4; w->info = (uint32_t*)p5; }
    return off;
}

extern "C" size_t xr_rays_sampler_workspace_bytes(uint32_t n_rays, uint32_t n_series) {
    return rm_ws_layout(n_rays, nullptr, nullptr, n_series ? n_series : 1);
}

// the launches behind xr_rays_sampler3 / xr_rays_sampler_series
static int rm_launch(const float* rays_o, const float* rays_d, const uint8_t* bitfield, uint32_t n_rays, uint32_t n_series, uint32_t ray_stride,
                     float aabb0, float aabb1, float near_distance, float cone_angle, uint32_t max_samples, xr_pcg32 rng,
                     float* coords_out, size_t coords_stride, int32_t* rays_index, int32_t* rays_numsteps, uint32_t* counter2, float* xyz_planes,
                     uint32_t plane_stride, uint32_t rng_chunk, uint32_t rng_ray0, uint32_t flags, void* workspace, hipStream_t stream) {
    RmWorkspace w; rm_ws_layout(n_rays, (char*)workspace, &w, n_series);
    const uint32_t nb = xr_div_up(n_rays, RM_BLOCK);
    // XR_K1_WIDE (the caller says nothing else is running: the march in place at a grid refresh) and at most K1W_MAX_RAYS rays: K1W lanes
    // per ray -- 8x the waves, each shorter: 174 -> 143 us at 12.5 K rays, 170 -> 121 at 4 K (profiles/r04_k1_lanes_per_ray_ab.txt).
    // Not beside the training step: 8x the waves slow the scatter they run beside by more than the march
    // gains (iteration 0.408 -> 0.424 ms); not for frames (65 K rays: 173 -> 280 us, the chip is full of rays either way).
#ifndef K1W_ALWAYS
#define K1W_ALWAYS 0           // 1: every launch of up to K1W_MAX_RAYS rays takes the K1W-lane kernel (A/B of K1W = 2 / 4 beside the training step)
#endif
    if (n_series == 1 && ((flags & XR_K1_WIDE) != 0u || K1W_ALWAYS != 0) && (uint32_t)K1W_MAX_RAYS != 0u && n_rays <= (uint32_t)K1W_MAX_RAYS) {
        hipLaunchKernelGGL(k1_count_w, dim3(xr_div_up(n_rays * K1W, RM_BLOCK)), dim3(RM_BLOCK), 0, stream, n_rays, aabb0, aabb1, rays_o, rays_d,
                           bitfield, cone_angle, near_distance, rng, rng_chunk, rng_ray0, w.cnt, w.start_t, w.tlist);
        hipLaunchKernelGGL(k1_block_scan, dim3(nb), dim3(RM_BLOCK), 0, stream, n_rays, (const uint32_t*)w.cnt, w.local_off, w.block_tot);
    } else
        hipLaunchKernelGGL(k1_count, dim3(nb, n_series), dim3(RM_BLOCK), 0, stream, n_rays, aabb0, aabb1, rays_o, rays_d, bitfield,
                           cone_angle, near_distance, rng, rng_chunk, rng_ray0, w.cnt, w.local_off, w.start_t, w.block_tot, w.tlist, ray_stride);
#ifndef K1_INLINE_SCAN
#define K1_INLINE_SCAN 1
#endif
    const bool inline_scan = (K1_INLINE_SCAN || n_series > 1) && nb <= (uint32_t)RM_BLOCK;
    if (!inline_scan) hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, stream, nb, w.block_tot, max_samples, w.block_base, w.info);
    hipLaunchKernelGGL(k1_write, dim3(nb, n_series), dim3(RM_BLOCK), 0, stream, n_rays, aabb0, aabb1, rays_o, rays_d, bitfield,
                       cone_angle, max_samples, w.cnt, w.local_off, w.start_t, w.block_base, w.info,
                       inline_scan ? (const uint32_t*)w.block_tot : (const uint32_t*)nullptr, coords_out,
                       rays_index, rays_numsteps, counter2, w.tlist, xyz_planes, plane_stride, ray_stride, coords_stride);
    XR_LAUNCH_CHECK();
    return XR_OK;
}

extern "C" int xr_rays_sampler(const float* rays_o, const float* rays_d, const uint8_t* bitfield, uint32_t n_rays,
                                float aabb0, float aabb1, float near_distance, float cone_angle, uint32_t max_samples,
                                uint64_t rng_state, uint64_t rng_inc, float* coords_out, int32_t* rays_index,
                                int32_t* rays_numsteps, uint32_t* counter2, float* xyz_planes, uint32_t plane_stride,
                                uint32_t rng_chunk, uint32_t rng_ray0, uint32_t flags, void* workspace, size_t workspace_bytes, void* stream_) {
    XR_REQUIRE((flags & ~XR_K1_WIDE) == 0, "unknown flag");
    XR_REQUIRE(rays_o && rays_d && bitfield && coords_out && rays_index && rays_numsteps && counter2, "null pointer");
    XR_REQUIRE(!xyz_planes || plane_stride >= max_samples, "a position plane holds max_samples values");
    XR_REQUIRE(n_rays > 0 && n_rays <= (1u << 28), "n_rays out of range");
    XR_REQUIRE(workspace && workspace_bytes >= xr_rays_sampler_workspace_bytes(n_rays, 1), "workspace too small");
    return rm_launch(rays_o, rays_d, bitfield, n_rays, 1, 0, aabb0, aabb1, near_distance, cone_angle, max_samples, xr_pcg32{rng_state, rng_inc},
                     coords_out, 0, rays_index, rays_numsteps, counter2, xyz_planes, plane_stride, rng_chunk, rng_ray0, flags, workspace,
                     (hipStream_t)stream_);
}

// n_series launches of K1 over n_rays rays each as ONE (contract: include/xrnerf_mi355.h).  Launch c is bit for bit the launch
// xr_rays_sampler3 would make with the hidden generator's call index (first + c) on rays c * ray_stride .. and the c-th output buffers.
extern "C" int xr_rays_sampler_series(const float* rays_o, const float* rays_d, uint32_t ray_stride, const uint8_t* bitfield, uint32_t n_rays,
                                      uint32_t n_series, float aabb0, float aabb1, float near_distance, float cone_angle, uint32_t max_samples,
                                      uint64_t rng_state, uint64_t rng_inc, float* coords_out, size_t coords_stride, int32_t* rays_index,
                                      int32_t* rays_numsteps, uint32_t* counter2, float* xyz_planes, uint32_t plane_stride, void* workspace,
                                      size_t workspace_bytes, void* stream_) {
    XR_REQUIRE(rays_o && rays_d && bitfield && coords_out && rays_index && rays_numsteps && counter2, "null pointer");
    XR_REQUIRE(n_series >= 1 && n_series <= 65535u && n_rays > 0 && n_rays <= (1u << 28) && ray_stride >= n_rays, "bad series");
    XR_REQUIRE(coords_stride >= max_samples && (!xyz_planes || plane_stride >= max_samples), "a launch's buffers hold max_samples rows");
    XR_REQUIRE(workspace && workspace_bytes >= xr_rays_sampler_workspace_bytes(n_rays, n_series), "workspace too small");
    const uint32_t nb = xr_div_up(n_rays, RM_BLOCK);
    if (nb <= (uint32_t)RM_BLOCK || n_series == 1)
        return rm_launch(rays_o, rays_d, bitfield, n_rays, n_series, ray_stride, aabb0, aabb1, near_distance, cone_angle, max_samples,
                         xr_pcg32{rng_state, rng_inc}, coords_out, coords_stride, rays_index, rays_numsteps, counter2, xyz_planes, plane_stride,
                         0, 0, 0, workspace, (hipStream_t)stream_);
    // launches of more than 65 536 rays need the one-workgroup scan between the two passes: one after the other
    xr_pcg32 rng{rng_state, rng_inc};
    for (uint32_t c = 0; c < n_series; ++c) {
        int rc = rm_launch(rays_o + 3 * (size_t)c * ray_stride, rays_d + 3 * (size_t)c * ray_stride, bitfield, n_rays, 1, 0, aabb0, aabb1, near_distance,
                           cone_angle, max_samples, rng, coords_out + 7 * c * coords_stride, 0, rays_index + (size_t)c * ray_stride,
                           rays_numsteps + 2 * (size_t)c * ray_stride, counter2 + 2 * c, xyz_planes ? xyz_planes + 3 * (size_t)c * plane_stride : nullptr,
                           plane_stride, 0, 0, 0, workspace, (hipStream_t)stream_);
        if (rc != XR_OK) return rc;
        rng.advance(1ull << 32);
    }
    return XR_OK;
}

// ------------------------------------------------------------------ K2 re-pack (compacted_coord.cu:22-76)
__global__ __launch_bounds__(RM_BLOCK) void k2_count(uint32_t n_rays, const int32_t* __restrict__ numsteps_in,
                                                      uint32_t* __restrict__ local_off, uint32_t* __restrict__ block_tot) {
    __shared__ uint32_t lds4[4];
    const uint32_t i = blockIdx.x * RM_BLOCK + threadIdx.x;
    uint32_t n = i < n_rays ? (uint32_t)numsteps_in[2 * i] : 0u, tot;
    uint32_t off = block_excl_scan(n, &tot, lds4);
    if (i < n_rays) local_off[i] = off;
    if (threadIdx.x == 0) block_tot[blockIdx.x] = tot;
}
__global__ __launch_bounds__(RM_BLOCK) void k2_copy(uint32_t n_rays, uint32_t max_compacted,
                                                     const float* __restrict__ coords_in, const int32_t* __restrict__ numsteps_in,
                                                     const uint32_t* __restrict__ local_off, const uint32_t* __restrict__ block_base,
                                                     const uint32_t* __restrict__ info, float* __restrict__ coords_out,
                                                     int32_t* __restrict__ numsteps_out, uint32_t* __restrict__ rays_counter,
                                                     uint32_t* __restrict__ numstep_counter) {
    const uint32_t i = blockIdx.x * RM_BLOCK + threadIdx.x;
    const int lane = threadIdx.x & 63;
    uint32_t n = 0, base = 0, cbase = 0, nc = 0;
    if (i < n_rays) {
        n = (uint32_t)numsteps_in[2 * i]; base = (uint32_t)numsteps_in[2 * i + 1];
        cbase = block_base[blockIdx.x] + local_off[i];                                   // :63
        nc = min(max_compacted - min(max_compacted, cbase), n);                          // :64
        numsteps_out[2 * i] = (int32_t)nc; numsteps_out[2 * i + 1] = (int32_t)cbase;
    }
    // rays with nc > 0 (:67-71): one wave-level count, one atomic per wave
    unsigned long long m = __ballot(nc > 0);
    if (lane == 0 && m) atomicAdd(rays_counter, (uint32_t)__popcll(m));
    if (i == 0) *numstep_counter = info[0];
    // cooperative copy: the wave walks its 64 rays, all lanes copy one ray's rows coalesced
    for (int s = 0; s < 64; ++s) {
        uint32_t rn = __shfl(nc, s, 64);
        if (rn == 0) continue;
        uint32_t rb = __shfl(base, s, 64), rc = __shfl(cbase, s, 64);
        const float* __restrict__ src = coords_in + 7 * (size_t)rb;
        float* __restrict__ dst = coords_out + 7 * (size_t)rc;
        for (uint32_t e = lane; e < 7 * rn; e += 64) dst[e] = src[e];
    }
}

extern "C" int xr_compacted_coord(const float* coords_in, const int32_t* numsteps_in, uint32_t n_rays,
                                  uint32_t max_compacted, float* coords_out, int32_t* numsteps_out,
                                  uint32_t* rays_counter, uint32_t* numstep_counter, void* workspace,
                                  size_t workspace_bytes, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    XR_REQUIRE(coords_in && numsteps_in && coords_out && numsteps_out && rays_counter && numstep_counter, "null pointer");
    XR_REQUIRE(n_rays > 0, "n_rays == 0");
    XR_REQUIRE(workspace && workspace_bytes >= xr_rays_sampler_workspace_bytes(n_rays, 1), "workspace too small");
    RmWorkspace w; rm_ws_layout(n_rays, (char*)workspace, &w);
    const uint32_t nb = xr_div_up(n_rays, RM_BLOCK);
    XR_HIP(hipMemsetAsync(rays_counter, 0, 4, stream));
    hipLaunchKernelGGL(k2_count, dim3(nb), dim3(RM_BLOCK), 0, stream, n_rays, numsteps_in, w.local_off, w.block_tot);
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, stream, nb, w.block_tot, 0xffffffffu, w.block_base, w.info);
    hipLaunchKernelGGL(k2_copy, dim3(nb), dim3(RM_BLOCK), 0, stream, n_rays, max_compacted, coords_in, numsteps_in,
                       w.local_off, w.block_base, w.info, coords_out, numsteps_out, rays_counter, numstep_counter);
    XR_LAUNCH_CHECK();
    return XR_OK;
}

__global__ __launch_bounds__(RM_BLOCK) void k2_clip(uint32_t n_rays, uint32_t max_compacted, const int32_t* __restrict__ in,
                                                     const uint32_t* __restrict__ counter2, int32_t* __restrict__ out,
                                                     uint32_t* __restrict__ n_valid, uint32_t chunk_rows, uint32_t n_chunks,
                                                     uint32_t ray_stride) {
    const uint32_t i = blockIdx.x * RM_BLOCK + threadIdx.x;
    // launch blockIdx.y of a series (xr_clip_numsteps_series): its rays, its counter pair, its (1 + n_chunks) valid-row counts
    in += 2 * (size_t)blockIdx.y * ray_stride; out += 2 * (size_t)blockIdx.y * ray_stride;
    counter2 += 2 * blockIdx.y; n_valid += (size_t)(1 + n_chunks) * blockIdx.y;
    if (i == 0) {
        const uint32_t total = min(counter2[1], max_compacted);
        n_valid[0] = total;
        for (uint32_t c = 0; c < n_chunks; ++c)       // rows of chunk c = [c*chunk_rows, (c+1)*chunk_rows) that are valid
            n_valid[1 + c] = min(chunk_rows, total - min(total, c * chunk_rows));
    }
    if (i >= n_rays) return;
    const uint32_t n = (uint32_t)in[2 * i], base = (uint32_t)in[2 * i + 1];
    out[2 * i] = (int32_t)min(max_compacted - min(max_compacted, base), n);
    out[2 * i + 1] = (int32_t)base;
}
// K2's clip for the n_series launches of xr_rays_sampler_series: numsteps arrays with ray_stride rows per launch, counter2 [n_series][2],
// n_valid_dev [n_series][2] = (valid rows, valid rows) like xr_clip_numsteps(chunk_rows = max_compacted, n_chunks = 1)
extern "C" int xr_clip_numsteps(const int32_t* numsteps_in, const uint32_t* counter2, uint32_t n_rays, uint32_t n_series, uint32_t ray_stride,
                                       uint32_t max_compacted, int32_t* numsteps_out, uint32_t* n_valid_dev, void* stream_) {
    XR_REQUIRE(numsteps_in && counter2 && numsteps_out && n_valid_dev && n_rays > 0 && n_series >= 1 && n_series <= 65535u && ray_stride >= n_rays, "bad argument");
    hipLaunchKernelGGL(k2_clip, dim3(xr_div_up(n_rays, RM_BLOCK), n_series), dim3(RM_BLOCK), 0, (hipStream_t)stream_, n_rays, max_compacted,
                       numsteps_in, counter2, numsteps_out, n_valid_dev, max_compacted, 1u, ray_stride);
    XR_LAUNCH_CHECK();
    return XR_OK;
}

// ------------------------------------------------------------------ K3 / K5 compositor forward
// Reference: one thread per ray, front to back (calc_rgb.cu:20-66, :158-205).  A training batch has
// only ~1e4 rays -- far too few threads for a 256-CU chip -- so here CG = 16 consecutive lanes share a
// ray: each takes a contiguous chunk of its samples, composites it locally (T_loc, c_loc), and the
// chunks are stitched with the associativity of the transmittance product:
//   T_before(chunk m) = prod_{j<m} T_loc(j),   C = sum_m T_before(m) * c_loc(m).
// Same arithmetic per sample, 16x the parallelism, 1/16 of the serial length; a chunk of up to CG_RC samples is
// fetched with all its loads in flight together (one exposed memory latency instead of one per sample).
#define CG 16
#define CG_RC 4          // chunk lengths up to CG_RC are loaded into registers in one go (rays of <= 64 samples)
__device__ inline void cg_chunk(uint32_t n, uint32_t sub, uint32_t* k0, uint32_t* k1) {
    const uint32_t chunk = (n + CG - 1) / CG;
    *k0 = min(n, sub * chunk); *k1 = min(n, (sub + 1) * chunk);
}
// exclusive prefix over the CG lanes of a ray: T_before and the colour accumulated before this chunk
__device__ inline void cg_stitch(uint32_t sub, float T_loc, float cr, float cg, float cb, float* Tb, float* Cbr, float* Cbg,
                                 float* Cbb, float* Ttot, float* Cr, float* Cg, float* Cb) {
    float tb = 1.f, ar = 0.f, ag = 0.f, ab = 0.f, tt = 1.f, tr = 0.f, tg = 0.f, tbb = 0.f;
#pragma unroll
    for (int m = 0; m < CG; ++m) {
        const float tm = __shfl(T_loc, m, CG), rm = __shfl(cr, m, CG), gm = __shfl(cg, m, CG), bm = __shfl(cb, m, CG);
        if ((uint32_t)m == sub) { tb = tt; ar = tr; ag = tg; ab = tbb; }
        tr += tt * rm; tg += tt * gm; tbb += tt * bm;
        tt *= tm;
    }
    *Tb = tb; *Cbr = ar; *Cbg = ag; *Cbb = ab; *Ttot = tt; *Cr = tr; *Cg = tg; *Cb = tbb;
}

// a chunk longer than CG_RC: batches of CG_LB samples, each batch's loads in flight together.  The per-sample loop this replaces
// exposed one memory latency per sample, and the launch lasts as long as its longest ray: 16 % of a training batch's rays have
// more than 64 samples (max ~150: 9-10 per lane, twice over in the fused kernel) -- tools/microbench_composite.py, 36.9 us as
// captured against 21.8 us with every ray cut to 64.  Same samples in the same order per lane: results unchanged bit for bit.
#define CG_LB 8
template <class F>
__device__ __forceinline__ void cg_long(const float4* __restrict__ raw, const float* __restrict__ coords, uint32_t base, uint32_t k0,
                                        uint32_t k1, F&& f) {
    for (uint32_t b = k0; b < k1; b += CG_LB) {
        float4 o[CG_LB]; float d[CG_LB];
        const uint32_t mm = min((uint32_t)CG_LB, k1 - b);
#pragma unroll
        for (uint32_t u = 0; u < CG_LB; ++u)
            if (u < mm) { o[u] = raw[base + b + u]; d[u] = coords[7 * (size_t)(base + b + u) + 3]; }
#pragma unroll
        for (uint32_t u = 0; u < CG_LB; ++u) if (u < mm) f(o[u], d[u], b + u);
    }
}

template <bool INFERENCE>
__global__ __launch_bounds__(RM_BLOCK) void k_composite_fwd(
    uint32_t n_rays, const float4* __restrict__ raw, const float* __restrict__ coords,
    const int32_t* __restrict__ numsteps, const int32_t* __restrict__ numsteps_c, const float* __restrict__ bg,
    float bg_r, float bg_g, float bg_b, int rgb_act, int density_act, float* __restrict__ rgb_out,
    float* __restrict__ alpha_out) {
    const uint32_t t = blockIdx.x * RM_BLOCK + threadIdx.x;
    const uint32_t i = t / CG, sub = t % CG;
    const bool in = i < n_rays;
    uint32_t n = 0, base = 0;
    if (in) { n = (uint32_t)numsteps_c[2 * i]; base = (uint32_t)numsteps_c[2 * i + 1]; }
    uint32_t k0, k1;
    cg_chunk(n, sub, &k0, &k1);
    float T = 1.f, cr = 0.f, cg = 0.f, cb = 0.f;
    const uint32_t m = k1 - k0;
    float4 oc[CG_RC]; float dc[CG_RC];
    if (m <= CG_RC) {
#pragma unroll
        for (uint32_t u = 0; u < CG_RC; ++u)
            if (u < m) { oc[u] = raw[base + k0 + u]; dc[u] = coords[7 * (size_t)(base + k0 + u) + 3]; }
    }
    auto step = [&](const float4 o, float dtw) {
        const float dt = xr_unwarp_dt(dtw);
        const float density = xr_act_density(o.w, density_act);
        const float alpha = 1.f - __expf(-density * dt);
        const float w = alpha * T;
        cr += w * xr_act_rgb(o.x, rgb_act); cg += w * xr_act_rgb(o.y, rgb_act); cb += w * xr_act_rgb(o.z, rgb_act);
        T *= (1.f - alpha);
    };
    if (m <= CG_RC) {
#pragma unroll
        for (uint32_t u = 0; u < CG_RC; ++u) if (u < m) step(oc[u], dc[u]);
    } else {
        cg_long(raw, coords, base, k0, k1, [&](const float4 o, float dtw, uint32_t) { step(o, dtw); });
    }
    float Tb, ar, ag, ab, Tt, Cr, Cg, Cb;
    cg_stitch(sub, T, cr, cg, cb, &Tb, &ar, &ag, &ab, &Tt, &Cr, &Cg, &Cb);
    if (!in || sub != 0) return;
    float br, bgc, bb;
    if (INFERENCE) { br = bg_r; bgc = bg_g; bb = bg_b; }
    else { br = bg[3 * i]; bgc = bg[3 * i + 1]; bb = bg[3 * i + 2]; }
    if (n == 0) {                                                          // :28-32 / :167-172
        rgb_out[3 * i] = br; rgb_out[3 * i + 1] = bgc; rgb_out[3 * i + 2] = bb;
        if (INFERENCE) alpha_out[i] = 0.f;
        return;
    }
    const bool add_bg = INFERENCE ? true : (n == (uint32_t)numsteps[2 * i]);   // :61-64 / :200-203
    if (add_bg) { Cr += Tt * br; Cg += Tt * bgc; Cb += Tt * bb; }
    rgb_out[3 * i] = Cr; rgb_out[3 * i + 1] = Cg; rgb_out[3 * i + 2] = Cb;
    if (INFERENCE) alpha_out[i] = 1.f - Tt;
}

extern "C" int xr_calc_rgb_forward(const float* network_output, const float* coords, const int32_t* rays_numsteps,
                                   const int32_t* rays_numsteps_compacted, const float* bg_color, uint32_t n_rays,
                                   int rgb_activation, int density_activation, float* rgb_output, void* stream_) {
    XR_REQUIRE(network_output && coords && rays_numsteps && rays_numsteps_compacted && bg_color && rgb_output, "null pointer");
    XR_REQUIRE(((uintptr_t)network_output & 15) == 0, "network_output must be 16-byte aligned");
    XR_REQUIRE(n_rays > 0, "n_rays == 0");
    hipLaunchKernelGGL(k_composite_fwd<false>, dim3(xr_div_up((uint64_t)n_rays * CG, RM_BLOCK)), dim3(RM_BLOCK), 0, (hipStream_t)stream_,
                       n_rays, (const float4*)network_output, coords, rays_numsteps, rays_numsteps_compacted, bg_color,
                       0.f, 0.f, 0.f, rgb_activation, density_activation, rgb_output, (float*)nullptr);
    XR_LAUNCH_CHECK();
    return XR_OK;
}

extern "C" int xr_calc_rgb_inference(const float* network_output, const float* coords, const int32_t* rays_numsteps,
                                     float bg_r, float bg_g, float bg_b, uint32_t n_rays, int rgb_activation,
                                     int density_activation, float* rgb_output, float* alpha_output, void* stream_) {
    XR_REQUIRE(network_output && coords && rays_numsteps && rgb_output && alpha_output, "null pointer");
    XR_REQUIRE(((uintptr_t)network_output & 15) == 0, "network_output must be 16-byte aligned");
    XR_REQUIRE(n_rays > 0, "n_rays == 0");
    hipLaunchKernelGGL(k_composite_fwd<true>, dim3(xr_div_up((uint64_t)n_rays * CG, RM_BLOCK)), dim3(RM_BLOCK), 0, (hipStream_t)stream_,
                       n_rays, (const float4*)network_output, coords, rays_numsteps, rays_numsteps, (const float*)nullptr,
                       bg_r, bg_g, bg_b, rgb_activation, density_activation, rgb_output, alpha_output);
    XR_LAUNCH_CHECK();
    return XR_OK;
}

// ------------------------------------------------------------------ early-terminated rendering (optional)
// The reference's inference compositor walks every sample (its EPSILON is declared but unused,
// calc_rgb.cu:178).  For rendering, samples behind an opaque surface are pure waste: with T < eps = 1e-4
// they can change the pixel by less than 1e-4.  The frame is therefore evaluated in depth slices
// [s0, s1) of per-ray sample indices; before each slice the still-transparent rays are compacted
// (wave ballot + one atomic per wave), only their samples go through encode + MLP, and a per-ray
// state (T, rgb) carries the front-to-back integration across slices.
__global__ __launch_bounds__(RM_BLOCK) void k_slice_select(uint32_t n_rays, const int32_t* __restrict__ numsteps,
                                                            const float* __restrict__ T, float eps, uint32_t s0, uint32_t s1,
                                                            uint32_t* __restrict__ rows, int32_t* __restrict__ ray_off,
                                                            uint32_t* __restrict__ count) {
    const uint32_t i = blockIdx.x * RM_BLOCK + threadIdx.x;
    const int lane = threadIdx.x & 63;
    uint32_t cnt = 0, base = 0;
    if (i < n_rays) {
        const uint32_t n = (uint32_t)numsteps[2 * i];
        base = (uint32_t)numsteps[2 * i + 1];
        if (n > s0 && T[i] > eps) cnt = min(n, s1) - s0;
    }
    // wave-level exclusive prefix of cnt, one atomic per wave reserves the wave's range
    uint32_t inc = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { uint32_t o = __shfl_up(inc, d, 64); if (lane >= d) inc += o; }
    const uint32_t wave_total = __shfl(inc, 63, 64);
    uint32_t wave_base = 0;
    if (lane == 63 && wave_total) wave_base = atomicAdd(count, wave_total);
    wave_base = __shfl(wave_base, 63, 64);
    if (i >= n_rays) return;
    if (cnt == 0) { ray_off[i] = -1; return; }
    const uint32_t off = wave_base + inc - cnt;
    ray_off[i] = (int32_t)off;
    for (uint32_t j = 0; j < cnt; ++j) rows[off + j] = base + s0 + j;
}
__global__ __launch_bounds__(RM_BLOCK) void k_slice_composite(uint32_t n_rays, const float4* __restrict__ raw_s,
                                                               const float* __restrict__ coords, const int32_t* __restrict__ numsteps,
                                                               const int32_t* __restrict__ ray_off, uint32_t s0, uint32_t s1,
                                                               int rgb_act, int density_act, float* __restrict__ T,
                                                               float* __restrict__ rgb_acc) {
    const uint32_t i = blockIdx.x * RM_BLOCK + threadIdx.x;
    if (i >= n_rays) return;
    const int32_t off = ray_off[i];
    if (off < 0) return;
    const uint32_t n = (uint32_t)numsteps[2 * i], base = (uint32_t)numsteps[2 * i + 1];
    const uint32_t cnt = min(n, s1) - s0;
    float t = T[i], cr = rgb_acc[3 * i], cg = rgb_acc[3 * i + 1], cb = rgb_acc[3 * i + 2];
    for (uint32_t j = 0; j < cnt; ++j) {
        const float4 o = raw_s[(uint32_t)off + j];
        const float dt = xr_unwarp_dt(coords[7 * (size_t)(base + s0 + j) + 3]);
        const float alpha = 1.f - __expf(-xr_act_density(o.w, density_act) * dt);
        const float w = alpha * t;
        cr += w * xr_act_rgb(o.x, rgb_act); cg += w * xr_act_rgb(o.y, rgb_act); cb += w * xr_act_rgb(o.z, rgb_act);
        t *= (1.f - alpha);
    }
    T[i] = t; rgb_acc[3 * i] = cr; rgb_acc[3 * i + 1] = cg; rgb_acc[3 * i + 2] = cb;
}
extern "C" int xr_render_slice_select(const int32_t* rays_numsteps, const float* T, uint32_t n_rays, uint32_t s0, uint32_t s1,
                                      float eps, uint32_t* rows_out, int32_t* ray_offset_out, uint32_t* count_out, void* stream_) {
    XR_REQUIRE(rays_numsteps && T && rows_out && ray_offset_out && count_out && n_rays > 0 && s1 > s0, "bad argument");
    XR_HIP(hipMemsetAsync(count_out, 0, 4, (hipStream_t)stream_));
    hipLaunchKernelGGL(k_slice_select, dim3(xr_div_up(n_rays, RM_BLOCK)), dim3(RM_BLOCK), 0, (hipStream_t)stream_, n_rays,
                       rays_numsteps, T, eps, s0, s1, rows_out, ray_offset_out, count_out);
    XR_LAUNCH_CHECK();
    return XR_OK;
}
extern "C" int xr_render_slice_composite(const float* raw_slice, const float* coords, const int32_t* rays_numsteps,
                                         const int32_t* ray_offset, uint32_t n_rays, uint32_t s0, uint32_t s1, int rgb_activation,
                                         int density_activation, float* T, float* rgb_acc, void* stream_) {
    XR_REQUIRE(raw_slice && coords && rays_numsteps && ray_offset && T && rgb_acc && n_rays > 0, "bad argument");
    XR_REQUIRE(((uintptr_t)raw_slice & 15) == 0, "raw must be 16-byte aligned");
    hipLaunchKernelGGL(k_slice_composite, dim3(xr_div_up(n_rays, RM_BLOCK)), dim3(RM_BLOCK), 0, (hipStream_t)stream_, n_rays,
                       (const float4*)raw_slice, coords, rays_numsteps, ray_offset, s0, s1, rgb_activation, density_activation,
                       T, rgb_acc);
    XR_LAUNCH_CHECK();
    return XR_OK;
}

// ------------------------------------------------------------------ K4 compositor backward (calc_rgb.cu:87-139)
// Same CG-lanes-per-ray split: pass 1 composites each chunk locally to obtain, after stitching, the
// transmittance and colour accumulated BEFORE the chunk; pass 2 re-walks the chunk with those as
// start values and emits the gradients (global T_k = T_before * T_loc,k ; prefix C_k = C_before + T_before * c_loc,k).
__global__ __launch_bounds__(RM_BLOCK) void k_composite_bwd(
    uint32_t n_rays, const float4* __restrict__ raw, const int32_t* __restrict__ numsteps_c,
    const float* __restrict__ coords, const float* __restrict__ grad_rgb, const float* __restrict__ rgb_final,
    const float* __restrict__ density_grid_mean, int rgb_act, int density_act, float4* __restrict__ dout) {
    const uint32_t t = blockIdx.x * RM_BLOCK + threadIdx.x;
    const uint32_t i = t / CG, sub = t % CG;
    const bool in = i < n_rays;
    float loss_scale = 128.f; loss_scale /= (float)n_rays;                        // :92-93
    const float l2 = rgb_act == XR_ACT_EXPONENTIAL ? 1e-4f : 0.0f;                // :103
    const float l1 = density_grid_mean[0] < 0.01f ? 1e-4f : 0.0f;                 // :104
    uint32_t n = 0, base = 0;
    if (in) { n = (uint32_t)numsteps_c[2 * i]; base = (uint32_t)numsteps_c[2 * i + 1]; }
    uint32_t k0, k1;
    cg_chunk(n, sub, &k0, &k1);
    float T = 1.f, cr = 0.f, cg = 0.f, cb = 0.f;
    const uint32_t m = k1 - k0;
    float4 oc[CG_RC]; float dc[CG_RC];
    float gr = 0.f, gg = 0.f, gb = 0.f, fr = 0.f, fg = 0.f, fb = 0.f;      // issued with the chunk loads, used by pass 2
    if (in) {
        gr = grad_rgb[3 * i]; gg = grad_rgb[3 * i + 1]; gb = grad_rgb[3 * i + 2];
        fr = rgb_final[3 * i]; fg = rgb_final[3 * i + 1]; fb = rgb_final[3 * i + 2];
    }
    if (m <= CG_RC) {
#pragma unroll
        for (uint32_t u = 0; u < CG_RC; ++u)
            if (u < m) { oc[u] = raw[base + k0 + u]; dc[u] = coords[7 * (size_t)(base + k0 + u) + 3]; }
    }
    auto pass1 = [&](const float4 o, float dtw) {
        const float dt = xr_unwarp_dt(dtw);
        const float alpha = 1.f - __expf(-xr_act_density(o.w, density_act) * dt);
        const float w = alpha * T;
        cr += w * xr_act_rgb(o.x, rgb_act); cg += w * xr_act_rgb(o.y, rgb_act); cb += w * xr_act_rgb(o.z, rgb_act);
        T *= (1.f - alpha);
    };
    if (m <= CG_RC) {
#pragma unroll
        for (uint32_t u = 0; u < CG_RC; ++u) if (u < m) pass1(oc[u], dc[u]);
    } else {
        cg_long(raw, coords, base, k0, k1, [&](const float4 o, float dtw, uint32_t) { pass1(o, dtw); });
    }
    float Tb, ar, ag, ab, Tt, Cr, Cg, Cb;
    cg_stitch(sub, T, cr, cg, cb, &Tb, &ar, &ag, &ab, &Tt, &Cr, &Cg, &Cb);
    if (!in || k0 >= k1) return;
    T = Tb; cr = ar; cg = ag; cb = ab;
    auto pass2 = [&](const float4 o, float dtw, uint32_t k) {
        const float r = xr_act_rgb(o.x, rgb_act), g = xr_act_rgb(o.y, rgb_act), b = xr_act_rgb(o.z, rgb_act);
        const float dt = xr_unwarp_dt(dtw);
        const float density = xr_act_density(o.w, density_act);
        const float alpha = 1.f - __expf(-density * dt);
        const float w = alpha * T;
        cr += w * r; cg += w * g; cb += w * b;
        T *= (1.f - alpha);
        const float sr = fr - cr, sg = fg - cg, sb = fb - cb;                     // suffix
        float4 d;
        d.x = loss_scale * ((w * gr) * xr_dact_rgb(o.x, rgb_act) + fmaxf(0.0f, l2 * o.x));
        d.y = loss_scale * ((w * gg) * xr_dact_rgb(o.y, rgb_act) + fmaxf(0.0f, l2 * o.y));
        d.z = loss_scale * ((w * gb) * xr_dact_rgb(o.z, rgb_act) + fmaxf(0.0f, l2 * o.z));
        const float dot = gr * (T * r - sr) + gg * (T * g - sg) + gb * (T * b - sb);
        d.w = loss_scale * (xr_dact_density(o.w, density_act) * (dt * dot)) + (o.w < 0.f ? -l1 : 0.0f);
        dout[base + k] = d;
    };
    if (m <= CG_RC) {
#pragma unroll
        for (uint32_t u = 0; u < CG_RC; ++u) if (u < m) pass2(oc[u], dc[u], k0 + u);
    } else {
        cg_long(raw, coords, base, k0, k1, pass2);
    }
}

extern "C" int xr_calc_rgb_backward(const float* network_output, const int32_t* rays_numsteps_compacted,
                                    const float* coords, const float* grad_rgb, const float* rgb_output,
                                    const float* density_grid_mean, uint32_t n_rays, int rgb_activation,
                                    int density_activation, float* dloss_doutput, void* stream_) {
    XR_REQUIRE(network_output && rays_numsteps_compacted && coords && grad_rgb && rgb_output && density_grid_mean &&
               dloss_doutput, "null pointer");
    XR_REQUIRE((((uintptr_t)network_output | (uintptr_t)dloss_doutput) & 15) == 0, "raw/grad buffers must be 16-byte aligned");
    XR_REQUIRE(n_rays > 0, "n_rays == 0");
    hipLaunchKernelGGL(k_composite_bwd, dim3(xr_div_up((uint64_t)n_rays * CG, RM_BLOCK)), dim3(RM_BLOCK), 0, (hipStream_t)stream_,
                       n_rays, (const float4*)network_output, rays_numsteps_compacted, coords, grad_rgb, rgb_output,
                       density_grid_mean, rgb_activation, density_activation, (float4*)dloss_doutput);
    XR_LAUNCH_CHECK();
    return XR_OK;
}

// ------------------------------------------------------------------ K3 + 5*Huber (+ masked MSE) + K4 in one launch
// The training step composites (K3), takes scale * HuberLoss(rgb, target) and the alpha-masked squared error the reference
// logs (networks/hashnerf.py:36-44), and runs K4 on dL/drgb -- three launches of 13 + 7 + 29 us with two 1.5-MB round trips
// (rgb, dL/drgb) in between.  K4's first pass IS the forward composite (same chunking, same stitch), so one kernel does all
// three: pass 1 + stitch -> final colour (+ background rule of K3) -> Huber gradient per ray in registers (lane 0 of the ray
// adds the ray's loss terms to a block sum) -> pass 2.  rgb_out, the loss accumulators and dout are what the three kernels
// produce (rgb and dout bit for bit; the two scalars up to the order of the block sums).
// CT_BLOCK = 1024: every workgroup ends with two fp32 atomics on the SAME cache line (loss, mse), and same-address L2 atomics
// retire one per ~13.5 ns -- measured: a launch with every ray EMPTY took 23.7 us at 786 workgroups of 256 threads, 42.6 us at
// 1572 and 13-15 us at 196 workgroups of 1024 (tools/microbench_composite.py).
// Round 3: what bounded the launch after that was the texture-address path of the ONE compute unit a workgroup sits on -- every lane
// reads its own contiguous chunk (16-B rows 64+ B apart across lanes: nothing coalesces) and a workgroup whose 64 rays are long
// issues ~35 such instructions per wave (tools/microbench_composite.py: 32 us as captured, 17 us with every ray cut to 64 samples,
// although only a quarter of the samples go away).  The workgroup's rays own ONE contiguous row range (K1's bases ascend), so it
// is staged through LDS: coalesced 16-B-per-lane loads of raw / dt rows in, each lane composites its chunk out of LDS exactly as
// before (same values, same order: same bits), dL/draw rows go back into the slots and leave with coalesced stores.  Ranges
// above CT_CAP rows (the first iterations, when every ray is long) take the direct path.
#define CT_BLOCK 1024
#define CT_SEGS 68
#define CT_CAP 6144u                          // rows: 96 KB of raw / dL/draw slots + 24 KB of dt
#define CT_MARK 0x7fc0a5a5u                   // dt slot of a row whose dL/draw slot has been written (a NaN payload K1 never stores)
#define CT_LDS_BYTES (CT_CAP * 20u)
__global__ __launch_bounds__(CT_BLOCK) void k_composite_train(
    uint32_t n_rays, const float4* __restrict__ raw, const float* __restrict__ coords, const int32_t* __restrict__ numsteps,
    const int32_t* __restrict__ numsteps_c, const float* __restrict__ bg, const float* __restrict__ target,
    const float* __restrict__ alpha_mask, const float* __restrict__ density_grid_mean, int rgb_act, int density_act, float delta,
    float scale, float* __restrict__ rgb_out, float* __restrict__ loss_mse, float4* __restrict__ dout,
    uint32_t* __restrict__ live_seg_count, int stage) {
    extern __shared__ __attribute__((aligned(16))) float4 ct_rows[];
    __shared__ float ws[CT_BLOCK / 64], ws2[CT_BLOCK / 64];
    // live rows (dL/d(raw) != 0) per XR_LIVE_SEG-row segment, counted while the rows are written: the workgroup's 64 rays own a
    // contiguous row range, i.e. <= CT_SEGS segments from its first row's on (rays of <= 1024 samples; beyond: global adds)
    __shared__ uint32_t seg_hist[CT_SEGS], seg_first, row_lo, row_hi;
    if (live_seg_count && threadIdx.x < CT_SEGS) seg_hist[threadIdx.x] = 0;
    if (threadIdx.x == 0) { row_lo = 0xffffffffu; row_hi = 0u; }
    const uint32_t t = blockIdx.x * CT_BLOCK + threadIdx.x;
    const uint32_t i = t / CG, sub = t % CG;
    const bool in = i < n_rays;
    float loss_scale = 128.f; loss_scale /= (float)n_rays;                        // calc_rgb.cu:92-93
    const float l2 = rgb_act == XR_ACT_EXPONENTIAL ? 1e-4f : 0.0f;
    const float l1 = density_grid_mean[0] < 0.01f ? 1e-4f : 0.0f;
    uint32_t n = 0, base = 0, n_full = 0;
    float br = 0.f, bgc = 0.f, bb = 0.f, tr = 0.f, tg = 0.f, tb = 0.f, am = 0.f;
    if (in) {
        n = (uint32_t)numsteps_c[2 * i]; base = (uint32_t)numsteps_c[2 * i + 1]; n_full = (uint32_t)numsteps[2 * i];
        br = bg[3 * i]; bgc = bg[3 * i + 1]; bb = bg[3 * i + 2];
        tr = target[3 * i]; tg = target[3 * i + 1]; tb = target[3 * i + 2];
        am = alpha_mask[i];
    }
    // the rows of this workgroup's rays: [row_lo, row_hi)
    __syncthreads();
    if (stage && in && sub == 0 && n > 0) { atomicMin(&row_lo, base); atomicMax(&row_hi, base + n); }
    __syncthreads();
    const uint32_t lo = row_lo, hi = row_hi;
    const bool staged = stage && hi > lo && hi - lo <= CT_CAP;                   // uniform over the workgroup
    float* const ct_dt = reinterpret_cast<float*>(ct_rows + CT_CAP);
    if (staged) {
        for (uint32_t r = threadIdx.x; r < hi - lo; r += CT_BLOCK) {
            ct_rows[r] = raw[lo + r];
            ct_dt[r] = coords[7 * (size_t)(lo + r) + 3];
        }
        __syncthreads();
    }
    uint32_t k0, k1;
    cg_chunk(n, sub, &k0, &k1);
    float T = 1.f, cr = 0.f, cg = 0.f, cb = 0.f;
    const uint32_t m = k1 - k0;
    const uint32_t slot0 = base - lo;                                            // LDS slot of the ray's first row (staged)
    float4 oc[CG_RC]; float dc[CG_RC];
    if (m <= CG_RC) {
#pragma unroll
        for (uint32_t u = 0; u < CG_RC; ++u)
            if (u < m) {
                if (staged) { oc[u] = ct_rows[slot0 + k0 + u]; dc[u] = ct_dt[slot0 + k0 + u]; }
                else { oc[u] = raw[base + k0 + u]; dc[u] = coords[7 * (size_t)(base + k0 + u) + 3]; }
            }
    }
    // a chunk longer than CG_RC: rows out of LDS one by one (no memory latency to batch), else cg_long's batches
    auto long_chunk = [&](auto&& f) {
        if (staged) { for (uint32_t k = k0; k < k1; ++k) f(ct_rows[slot0 + k], ct_dt[slot0 + k], k); }
        else cg_long(raw, coords, base, k0, k1, f);
    };
    auto pass1 = [&](const float4 o, float dtw, uint32_t) {
        const float dt = xr_unwarp_dt(dtw);
        const float alpha = 1.f - __expf(-xr_act_density(o.w, density_act) * dt);
        const float w = alpha * T;
        cr += w * xr_act_rgb(o.x, rgb_act); cg += w * xr_act_rgb(o.y, rgb_act); cb += w * xr_act_rgb(o.z, rgb_act);
        T *= (1.f - alpha);
    };
    if (m <= CG_RC) {
#pragma unroll
        for (uint32_t u = 0; u < CG_RC; ++u) if (u < m) pass1(oc[u], dc[u], 0u);
    } else {
        long_chunk(pass1);
    }
    float Tb, ar, ag, ab, Tt, Cr, Cg, Cb;
    cg_stitch(sub, T, cr, cg, cb, &Tb, &ar, &ag, &ab, &Tt, &Cr, &Cg, &Cb);
    // K3's result (:28-32, :61-64)
    float fr, fg, fb;
    if (n == 0) { fr = br; fg = bgc; fb = bb; }
    else { fr = Cr; fg = Cg; fb = Cb; if (n == n_full) { fr += Tt * br; fg += Tt * bgc; fb += Tt * bb; } }
    // scale * HuberLoss and its gradient (utils/metrics.py:8-16), masked squared error
    float gr, gg, gb, acc = 0.f, mse = 0.f;
    {
        const float d3[3] = {fr - tr, fg - tg, fb - tb};
        float g3[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float a = fabsf(d3[c]);
            if (a > delta) { acc += a - 0.5f * delta; g3[c] = scale * (d3[c] > 0.f ? 1.f : -1.f); }
            else { acc += 0.5f / delta * a * a; g3[c] = scale * (d3[c] / delta); }
            const float mm = d3[c] * am; mse += mm * mm;
        }
        gr = g3[0]; gg = g3[1]; gb = g3[2];
    }
    if (!(in && sub == 0)) { acc = 0.f; mse = 0.f; }
    else { rgb_out[3 * i] = fr; rgb_out[3 * i + 1] = fg; rgb_out[3 * i + 2] = fb; }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { acc += __shfl_xor(acc, d, 64); mse += __shfl_xor(mse, d, 64); }
    if ((threadIdx.x & 63) == 0) { ws[threadIdx.x >> 6] = acc; ws2[threadIdx.x >> 6] = mse; }
    if (threadIdx.x == 0) seg_first = base / XR_LIVE_SEG;          // ray bases ascend with the ray index (K1's prefix)
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int w = 0; w < CT_BLOCK / 64; ++w) { a += ws[w]; b += ws2[w]; }
        if (loss_mse && a != 0.f) atomicAdd(loss_mse, scale * a);
        if (loss_mse && b != 0.f) atomicAdd(loss_mse + 1, b);
    }
    if (in && k0 < k1) {
        T = Tb; cr = ar; cg = ag; cb = ab;
        auto pass2 = [&](const float4 o, float dtw, uint32_t k) {
            const float r = xr_act_rgb(o.x, rgb_act), g = xr_act_rgb(o.y, rgb_act), b = xr_act_rgb(o.z, rgb_act);
            const float dt = xr_unwarp_dt(dtw);
            const float density = xr_act_density(o.w, density_act);
            const float alpha = 1.f - __expf(-density * dt);
            const float w = alpha * T;
            cr += w * r; cg += w * g; cb += w * b;
            T *= (1.f - alpha);
            const float sr = fr - cr, sg = fg - cg, sb = fb - cb;                     // suffix
            float4 d;
            d.x = loss_scale * ((w * gr) * xr_dact_rgb(o.x, rgb_act) + fmaxf(0.0f, l2 * o.x));
            d.y = loss_scale * ((w * gg) * xr_dact_rgb(o.y, rgb_act) + fmaxf(0.0f, l2 * o.y));
            d.z = loss_scale * ((w * gb) * xr_dact_rgb(o.z, rgb_act) + fmaxf(0.0f, l2 * o.z));
            const float dot = gr * (T * r - sr) + gg * (T * g - sg) + gb * (T * b - sb);
            d.w = loss_scale * (xr_dact_density(o.w, density_act) * (dt * dot)) + (o.w < 0.f ? -l1 : 0.0f);
            if (staged) { ct_rows[slot0 + k] = d; ct_dt[slot0 + k] = __uint_as_float(CT_MARK); }
            else dout[base + k] = d;
            if (live_seg_count && (d.x != 0.f || d.y != 0.f || d.z != 0.f || d.w != 0.f)) {
                const uint32_t sg_ = (base + k) / XR_LIVE_SEG, rel = sg_ - seg_first;
                if (rel < CT_SEGS) atomicAdd(&seg_hist[rel], 1u); else atomicAdd(&live_seg_count[sg_], 1u);
            }
        };
        if (m <= CG_RC) {
#pragma unroll
            for (uint32_t u = 0; u < CG_RC; ++u) if (u < m) pass2(oc[u], dc[u], k0 + u);
        } else {
            long_chunk(pass2);
        }
    }
    if (!live_seg_count && !staged) return;
    __syncthreads();
    if (live_seg_count && threadIdx.x < CT_SEGS && seg_hist[threadIdx.x])
        atomicAdd(&live_seg_count[seg_first + threadIdx.x], seg_hist[threadIdx.x]);
    if (staged)
        for (uint32_t r = threadIdx.x; r < hi - lo; r += CT_BLOCK)
            if (__float_as_uint(ct_dt[r]) == CT_MARK) dout[lo + r] = ct_rows[r];
}

// ---- the fused compositor, one WAVE per ray (what xr_composite_train2 runs when the caller takes the loss scalars separately; build with
// -DXR_CT_WAVE=0 for the 16-lane kernel above in that case too -- tools/build_variant.sh)
// What bounds the 16-lane form is neither memory nor the loss atomics alone but the lock step: a wave's time is its LONGEST
// chunk (rays of 143 samples: 9 samples per lane, twice over) times ~390 issue cycles per sample (13 transcendentals), with 59 %
// of the lanes sitting on rays without samples, and a workgroup per compute unit makes that 4 waves per SIMD back to back --
// tools/microbench_composite.py follows 11.5 us + 1.3 us x (longest chunk) closely.  Here every ray gets a wave: chunks of
// ceil(n / 64) <= 3 samples (one for the typical ray: consecutive lanes read consecutive 16-B rows), 12.5 K waves over 1024 SIMDs,
// an empty ray costs its numsteps load.  The 64 chunks are stitched by ordered prefixes (16 lanes, then the 4 groups) under
//   (T_l, c_l) o (T_r, c_r) = (T_l T_r, c_l + T_l c_r),
// i.e. another association of the same products than the 16-lane kernels use (results differ from them like two fp32
// summation orders; each is held to the oracle separately).
// No loss accumulation here: the two scalars are a pure function of rgb_out (xr_train_loss_scalars: one fixed-order sum instead
// of 2 x 784 same-line atomics at 13.5 ns each -- reproducible run to run, and off this launch).
#define CW_RAYS 4                          // rays = waves per workgroup
#define CW_RC 2                            // chunk lengths kept in registers (rays of <= 128 samples)
#define CW_SEGS 8
__global__ __launch_bounds__(64 * CW_RAYS) void k_composite_train_w(
    uint32_t n_rays, const float4* __restrict__ raw, const float* __restrict__ coords, const int32_t* __restrict__ numsteps,
    const int32_t* __restrict__ numsteps_c, const float* __restrict__ bg, const float* __restrict__ target,
    const float* __restrict__ density_grid_mean, int rgb_act, int density_act, float delta, float scale,
    float* __restrict__ rgb_out, float4* __restrict__ dout, uint32_t* __restrict__ live_seg_count) {
    __shared__ uint32_t seg_hist[CW_SEGS], seg_first;
    const uint32_t lane = threadIdx.x & 63, i = blockIdx.x * CW_RAYS + (threadIdx.x >> 6);
    const bool in = i < n_rays;
    if (live_seg_count) {
        if (threadIdx.x < CW_SEGS) seg_hist[threadIdx.x] = 0;
        if (threadIdx.x == 0) seg_first = (uint32_t)numsteps_c[2 * i + 1] / XR_LIVE_SEG;    // (the grid has no workgroup without a ray)
        __syncthreads();
    }
    float loss_scale = 128.f; loss_scale /= (float)n_rays;                        // calc_rgb.cu:92-93
    const float l2 = rgb_act == XR_ACT_EXPONENTIAL ? 1e-4f : 0.0f;
    const float l1 = density_grid_mean[0] < 0.01f ? 1e-4f : 0.0f;
    uint32_t n = 0, base = 0, n_full = 0;
    if (in) { n = (uint32_t)numsteps_c[2 * i]; base = (uint32_t)numsteps_c[2 * i + 1]; n_full = (uint32_t)numsteps[2 * i]; }
    if (n > 0) {                                                                   // wave-uniform
        const uint32_t chunk = (n + 63) / 64;
        const uint32_t k0 = min(n, lane * chunk), k1 = min(n, (lane + 1) * chunk), m = k1 - k0;
        float4 oc[CW_RC]; float dc[CW_RC];
        if (chunk <= CW_RC) {
#pragma unroll
            for (uint32_t u = 0; u < CW_RC; ++u)
                if (u < m) { oc[u] = raw[base + k0 + u]; dc[u] = coords[7 * (size_t)(base + k0 + u) + 3]; }
        }
        const float br = bg[3 * i], bgc = bg[3 * i + 1], bb = bg[3 * i + 2];
        const float tr = target[3 * i], tg = target[3 * i + 1], tb = target[3 * i + 2];
        float T = 1.f, cr = 0.f, cg = 0.f, cb = 0.f;
        auto pass1 = [&](const float4 o, float dtw, uint32_t) {
            const float dt = xr_unwarp_dt(dtw);
            const float alpha = 1.f - __expf(-xr_act_density(o.w, density_act) * dt);
            const float w = alpha * T;
            cr += w * xr_act_rgb(o.x, rgb_act); cg += w * xr_act_rgb(o.y, rgb_act); cb += w * xr_act_rgb(o.z, rgb_act);
            T *= (1.f - alpha);
        };
        if (chunk <= CW_RC) {
#pragma unroll
            for (uint32_t u = 0; u < CW_RC; ++u) if (u < m) pass1(oc[u], dc[u], 0u);
        } else {
            cg_long(raw, coords, base, k0, k1, pass1);
        }
        // stitch: the 16-step ordered prefix of cg_stitch inside each group of 16 lanes, then an ordered prefix over the 4 groups.
        // (Not a log-step scan: behind an opaque sample -- T exactly 0 -- every later prefix and the ray's total must be the SAME
        // fp32 number, so that the suffix colour, and with it the whole dL/draw row, is exactly zero there: those rows are what
        // the live-row list drops.  An ordered prefix stops changing once T is 0; a tree combines different partial sums per lane.)
        float Tb1, ar1, ag1, ab1, Tg, Cgr, Cgg, Cgb;
        cg_stitch(lane & 15, T, cr, cg, cb, &Tb1, &ar1, &ag1, &ab1, &Tg, &Cgr, &Cgg, &Cgb);
        float GT = 1.f, Gr = 0.f, Gg = 0.f, Gb = 0.f, Tb = 1.f, ar = 0.f, ag = 0.f, ab = 0.f;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const float th = __shfl(Tg, 16 * h, 64), rh = __shfl(Cgr, 16 * h, 64), gh = __shfl(Cgg, 16 * h, 64), bh = __shfl(Cgb, 16 * h, 64);
            if ((int)(lane >> 4) == h) { Tb = GT * Tb1; ar = Gr + GT * ar1; ag = Gg + GT * ag1; ab = Gb + GT * ab1; }
            Gr += GT * rh; Gg += GT * gh; Gb += GT * bh;
            GT *= th;
        }
        const float Tt = GT;
        float fr = Gr, fg = Gg, fb = Gb;
        if (n == n_full) { fr += Tt * br; fg += Tt * bgc; fb += Tt * bb; }       // K3 (:61-64)
        if (lane == 0) { rgb_out[3 * i] = fr; rgb_out[3 * i + 1] = fg; rgb_out[3 * i + 2] = fb; }
        // gradient of scale * HuberLoss (utils/metrics.py:8-16)
        float g3[3];
        {
            const float d3[3] = {fr - tr, fg - tg, fb - tb};
#pragma unroll
            for (int c = 0; c < 3; ++c) g3[c] = fabsf(d3[c]) > delta ? scale * (d3[c] > 0.f ? 1.f : -1.f) : scale * (d3[c] / delta);
        }
        const float gr = g3[0], gg = g3[1], gb = g3[2];
        if (k0 < k1) {
            T = Tb; cr = ar; cg = ag; cb = ab;
            uint32_t live_lo = 0, live_hi = 0;                                    // live rows of this chunk in its first / next segment
            const uint32_t seg_lo = (base + k0) / XR_LIVE_SEG;
            auto pass2 = [&](const float4 o, float dtw, uint32_t k) {
                const float r = xr_act_rgb(o.x, rgb_act), g = xr_act_rgb(o.y, rgb_act), b = xr_act_rgb(o.z, rgb_act);
                const float dt = xr_unwarp_dt(dtw);
                const float density = xr_act_density(o.w, density_act);
                const float alpha = 1.f - __expf(-density * dt);
                const float w = alpha * T;
                cr += w * r; cg += w * g; cb += w * b;
                T *= (1.f - alpha);
                const float ur = fr - cr, ug = fg - cg, ub = fb - cb;                 // suffix
                float4 d;
                d.x = loss_scale * ((w * gr) * xr_dact_rgb(o.x, rgb_act) + fmaxf(0.0f, l2 * o.x));
                d.y = loss_scale * ((w * gg) * xr_dact_rgb(o.y, rgb_act) + fmaxf(0.0f, l2 * o.y));
                d.z = loss_scale * ((w * gb) * xr_dact_rgb(o.z, rgb_act) + fmaxf(0.0f, l2 * o.z));
                const float dot = gr * (T * r - ur) + gg * (T * g - ug) + gb * (T * b - ub);
                d.w = loss_scale * (xr_dact_density(o.w, density_act) * (dt * dot)) + (o.w < 0.f ? -l1 : 0.0f);
                dout[base + k] = d;
                if (d.x != 0.f || d.y != 0.f || d.z != 0.f || d.w != 0.f) {
                    if ((base + k) / XR_LIVE_SEG == seg_lo) ++live_lo; else ++live_hi;   // a chunk of <= 16 rows spans <= 2 segments
                }
            };
            if (chunk <= CW_RC) {
#pragma unroll
                for (uint32_t u = 0; u < CW_RC; ++u) if (u < m) pass2(oc[u], dc[u], k0 + u);
            } else {
                cg_long(raw, coords, base, k0, k1, pass2);
            }
            if (live_seg_count) {
                const uint32_t rel = seg_lo - seg_first;
                if (live_lo) { if (rel < CW_SEGS) atomicAdd(&seg_hist[rel], live_lo); else atomicAdd(&live_seg_count[seg_lo], live_lo); }
                if (live_hi) { if (rel + 1 < CW_SEGS) atomicAdd(&seg_hist[rel + 1], live_hi); else atomicAdd(&live_seg_count[seg_lo + 1], live_hi); }
            }
        }
    } else if (in && lane == 0) {                                                  // K3 (:28-32): no samples -> the background
        rgb_out[3 * i] = bg[3 * i]; rgb_out[3 * i + 1] = bg[3 * i + 1]; rgb_out[3 * i + 2] = bg[3 * i + 2];
    }
    if (!live_seg_count) return;
    __syncthreads();
    if (threadIdx.x < CW_SEGS && seg_hist[threadIdx.x]) atomicAdd(&live_seg_count[seg_first + threadIdx.x], seg_hist[threadIdx.x]);
}

// scale * sum HuberLoss(rgb - target) and sum ((rgb - target) * alpha)^2 as ONE workgroup's fixed-order sum: out[0], out[1] are
// WRITTEN (utils/metrics.py:8-16, networks/hashnerf.py:36-44)
#define LS_THREADS XR_AUX_LS_VTHREADS
__global__ __launch_bounds__(LS_THREADS) void k_train_loss_scalars(const float* __restrict__ rgb, const float* __restrict__ target,
                                                                  const float* __restrict__ alpha_mask, uint32_t n_rays, float delta,
                                                                  float scale, float* __restrict__ out) {
    __shared__ float ws[LS_THREADS / 64], ws2[LS_THREADS / 64];
    xr_aux_loss_block<LS_THREADS>(rgb, target, alpha_mask, n_rays, delta, scale, out, ws, ws2);      // (xr_aux.h: the one definition)
}
extern "C" int xr_train_loss_scalars(const float* rgb, const float* target, const float* alpha_mask, uint32_t n_rays, float delta,
                                     float scale, float* loss_mse_out, void* stream_) {
    XR_REQUIRE(rgb && target && alpha_mask && loss_mse_out && n_rays > 0, "bad argument");
    hipLaunchKernelGGL(k_train_loss_scalars, dim3(1), dim3(LS_THREADS), 0, (hipStream_t)stream_, rgb, target, alpha_mask, n_rays, delta,
                       scale, loss_mse_out);
    XR_LAUNCH_CHECK();
    return XR_OK;
}

// loss_mse_out == nullptr: the wave-per-ray kernel; the two loss scalars are left to xr_train_loss_scalars (a function of
// rgb_output).  With loss_mse_out: the 16-lanes-per-ray kernel, which ADDS them (block sums + two atomics per workgroup).
extern "C" int xr_composite_train(const float* network_output, const float* coords, const int32_t* rays_numsteps,
                                   const int32_t* rays_numsteps_compacted, const float* bg_color, const float* target,
                                   const float* alpha_mask, const float* density_grid_mean, uint32_t n_rays, int rgb_activation,
                                   int density_activation, float delta, float scale, float* rgb_output, float* loss_mse_out,
                                   float* dloss_doutput, uint32_t* live_seg_count, void* stream_) {
    XR_REQUIRE(network_output && coords && rays_numsteps && rays_numsteps_compacted && bg_color && target && alpha_mask &&
               density_grid_mean && rgb_output && dloss_doutput, "null pointer");
    XR_REQUIRE((((uintptr_t)network_output | (uintptr_t)dloss_doutput) & 15) == 0, "raw/grad buffers must be 16-byte aligned");
    XR_REQUIRE(n_rays > 0, "n_rays == 0");
#ifndef XR_CT_WAVE
#define XR_CT_WAVE 1
#endif
#ifndef XR_CT_STAGE
#define XR_CT_STAGE 1       // 0: every workgroup of the 16-lane kernel on the direct path (no LDS staging of its row range)
#endif
    if (!loss_mse_out && XR_CT_WAVE) {
        hipLaunchKernelGGL(k_composite_train_w, dim3(xr_div_up(n_rays, CW_RAYS)), dim3(64 * CW_RAYS), 0, (hipStream_t)stream_, n_rays,
                           (const float4*)network_output, coords, rays_numsteps, rays_numsteps_compacted, bg_color, target,
                           density_grid_mean, rgb_activation, density_activation, delta, scale, rgb_output, (float4*)dloss_doutput,
                           live_seg_count);
        XR_LAUNCH_CHECK();
        return XR_OK;
    }
    const int stage = XR_CT_STAGE;
    static thread_local bool attr_set = false;
    if (!attr_set) {
        XR_HIP(hipFuncSetAttribute((const void*)k_composite_train, hipFuncAttributeMaxDynamicSharedMemorySize, (int)CT_LDS_BYTES));
        attr_set = true;
    }
    hipLaunchKernelGGL(k_composite_train, dim3(xr_div_up((uint64_t)n_rays * CG, CT_BLOCK)), dim3(CT_BLOCK), CT_LDS_BYTES, (hipStream_t)stream_,
                       n_rays, (const float4*)network_output, coords, rays_numsteps, rays_numsteps_compacted, bg_color, target,
                       alpha_mask, density_grid_mean, rgb_activation, density_activation, delta, scale, rgb_output, loss_mse_out,
                       (float4*)dloss_doutput, live_seg_count, stage);
    XR_LAUNCH_CHECK();
    return XR_OK;
}
