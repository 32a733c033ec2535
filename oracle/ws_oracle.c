/*
 * ws_oracle.c -- CPU restatement of the qts8n/water-sandbox SPH fluid step.
 *
 * TEST INFRASTRUCTURE ONLY (see ws_oracle.h).  PARITY UNPINNED BY THE REFERENCE:
 * the reference has no tests / golden vectors / CPU physics and cannot be built or
 * run in this image, so this restatement is pinned only by known-answer values
 * derived from the reference's source text (tests/test_oracle_kat.py).
 *
 * Build: see oracle/Makefile (-O2 -ffp-contract=off, no fast-math, OpenMP).
 * Every float operation below is a single IEEE binary32 operation written in the
 * order the WGSL writes it; do not "simplify" expressions in this file.
 */
#include "ws_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------- */
/* host-side generators                                                       */
/* ------------------------------------------------------------------------- */

/* src/fluid_compute.rs:20-27 (consts) and :67-79 (Default). delta_time is the
 * f32 constant expression `1. / 60.` (PARTICLE_LOOKAHEAD_SCALAR, :27,:70). */
void wso_default_props(wso_props *out)
{
    out->delta_time = 1.0f / 60.0f;
    out->collision_damping = 0.95f;
    out->smoothing_radius = 0.25f;
    out->target_density = 10.0f;
    out->pressure_scalar = 22.0f;
    out->near_pressure_scalar = 2.0f;
    out->viscosity_strength = 0.1f;
}

/* Rust's f32::powi lowers to llvm.powi.f32; its generic lowering is compiler-rt's
 * __powisf2: square-and-multiply from the low bit.  (For h = 0.25 every product
 * is exact, so the order does not matter at the reference defaults.) */
static float powi_f32(float a, int b)
{
    const int recip = b < 0;
    float r = 1.0f;
    for (;;) {
        if (b & 1) r *= a;
        b /= 2;
        if (b == 0) break;
        a *= a;
    }
    return recip ? 1.0f / r : r;
}

/* src/fluid_compute.rs:55-63.  PI is std::f32::consts::PI. */
void wso_smoothing_kernel(const wso_props *p, wso_kernel *out)
{
    const float PI = 3.14159265358979323846f;
    const float h = p->smoothing_radius;
    out->pow2 = 15.0f / (2.0f * PI * powi_f32(h, 5));
    out->pow2_der = 15.0f / (PI * powi_f32(h, 5));
    out->pow3 = 15.0f / (PI * powi_f32(h, 6));
    out->pow3_der = 45.0f / (PI * powi_f32(h, 6));
    out->spikey_pow3 = 315.0f / (64.0f * PI * powi_f32(h, 9));
}

/* src/helpers.rs:3-20.  Loop nest i (outer), j, k (inner); each point is
 * Vec3(i*diam, j*diam, k*diam) + offset, componentwise f32. */
void wso_cube_fluid(uint32_t ni, uint32_t nj, uint32_t nk, float particle_rad, float *out_xyz)
{
    const float hx = (float)ni * particle_rad;
    const float hy = (float)nj * particle_rad;
    const float hz = (float)nk * particle_rad;
    const float ox = particle_rad - hx;
    const float oy = particle_rad - hy;
    const float oz = particle_rad - hz;
    const float diam = particle_rad * 2.0f;
    size_t o = 0;
    for (uint32_t i = 0; i < ni; i++) {
        const float x = (float)i * diam;
        for (uint32_t j = 0; j < nj; j++) {
            const float y = (float)j * diam;
            for (uint32_t k = 0; k < nk; k++) {
                const float z = (float)k * diam;
                out_xyz[o++] = x + ox;
                out_xyz[o++] = y + oy;
                out_xyz[o++] = z + oz;
            }
        }
    }
}

/* src/fluid_container.rs:42-50: (position - half_size + padding), (position +
 * half_size - padding), left to right, w = 0. */
void wso_get_ext(const float position[3], const float size[3], float padding, float ext_min[4],
                 float ext_max[4])
{
    for (int c = 0; c < 3; c++) {
        const float half = size[c] / 2.0f;
        ext_min[c] = (position[c] - half) + padding;
        ext_max[c] = (position[c] + half) - padding;
    }
    ext_min[3] = 0.0f;
    ext_max[3] = 0.0f;
}

/* src/gravity.rs:6,29-33 */
void wso_default_gravity(float g[4])
{
    g[0] = 0.0f;
    g[1] = -9.8f;
    g[2] = 0.0f;
    g[3] = 0.0f;
}

/* src/fluid_compute.rs:251-273: dim = 2,4,..,P; block = dim/2,..,1. */
uint32_t wso_bit_sorter_stages(uint32_t data_length, wso_bit_sorter *out, uint32_t cap)
{
    uint64_t input_length = 1;
    while (input_length < data_length) input_length <<= 1; /* checked_next_power_of_two */
    uint32_t count = 0;
    for (uint64_t dim = 2; dim <= input_length; dim <<= 1) {
        for (uint64_t block = dim >> 1; block > 0; block >>= 1) {
            if (out && count < cap) {
                out[count].block = (uint32_t)block;
                out[count].dim = (uint32_t)dim;
            }
            count++;
        }
    }
    return count;
}

/* src/fluid_compute.rs:118-130 */
void wso_make_particles(const float *xyz, uint32_t n, wso_particle *out)
{
    memset(out, 0, (size_t)n * sizeof(wso_particle));
    for (uint32_t i = 0; i < n; i++) {
        for (int c = 0; c < 3; c++) {
            out[i].position[c] = xyz[3 * (size_t)i + c];
            out[i].predicted_position[c] = xyz[3 * (size_t)i + c];
        }
    }
}

/* src/fluid_compute.rs:243-249 */
void wso_identity(uint32_t n, uint32_t *out)
{
    for (uint32_t i = 0; i < n; i++) out[i] = i;
}

static uint64_t splitmix64(uint64_t x)
{
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* SURVEY.md 8(d) distribution (B); not part of the reference. */
void wso_uniform_cloud(uint32_t n, uint64_t seed, const float ext_min[4], const float ext_max[4],
                       float *out_xyz)
{
    for (uint32_t i = 0; i < n; i++) {
        for (uint32_t c = 0; c < 3; c++) {
            const uint64_t r = splitmix64(seed ^ (3ull * i + c)) >> 40;
            const float u = (float)r * 5.9604644775390625e-08f; /* 2^-24, exact */
            const float span = ext_max[c] - ext_min[c];
            out_xyz[3 * (size_t)i + c] = ext_min[c] + u * span;
        }
    }
}

/* ------------------------------------------------------------------------- */
/* cell / hash                                                                */
/* ------------------------------------------------------------------------- */

/* assets/simulation.wgsl:121-123: vec3<i32>(floor(position / h)) */
void wso_get_cell(const float pos[3], float h, int32_t cell[3])
{
    for (int c = 0; c < 3; c++) cell[c] = (int32_t)floorf(pos[c] / h);
}

/* assets/simulation.wgsl:125-128: vec3<u32>(cell) is a bit reinterpretation;
 * u32 products and sums wrap; then % num_particles. */
uint32_t wso_hash_cell(const int32_t cell[3], uint32_t n)
{
    const uint32_t x = (uint32_t)cell[0];
    const uint32_t y = (uint32_t)cell[1];
    const uint32_t z = (uint32_t)cell[2];
    return (x * WSO_P1 + y * WSO_P2 + z * WSO_P3) % n;
}

/* assets/simulation.wgsl:6-34: x slowest, z fastest, each -1,0,+1 */
static const int32_t OFFSET_TABLE[27][3] = {
    {-1, -1, -1}, {-1, -1, 0}, {-1, -1, 1}, {-1, 0, -1}, {-1, 0, 0}, {-1, 0, 1}, {-1, 1, -1},
    {-1, 1, 0},   {-1, 1, 1},  {0, -1, -1}, {0, -1, 0},  {0, -1, 1}, {0, 0, -1}, {0, 0, 0},
    {0, 0, 1},    {0, 1, -1},  {0, 1, 0},   {0, 1, 1},   {1, -1, -1}, {1, -1, 0}, {1, -1, 1},
    {1, 0, -1},   {1, 0, 0},   {1, 0, 1},   {1, 1, -1},  {1, 1, 0},  {1, 1, 1},
};

/* ------------------------------------------------------------------------- */
/* K1 hash_particles, assets/simulation.wgsl:130-141                          */
/* ------------------------------------------------------------------------- */
void wso_hash_particles(wso_state *s)
{
    const uint32_t n = s->num_particles;
    const float h = s->props.smoothing_radius;
#pragma omp parallel for schedule(static)
    for (uint32_t index = 0; index < n; index++) {
        s->cell_offsets[index] = WSO_INF;
        const uint32_t pid = s->particle_indicies[index];
        int32_t cell[3];
        wso_get_cell(s->particles[pid].predicted_position, h, cell);
        s->particle_cell_indicies[pid] = wso_hash_cell(cell, n);
    }
}

/* ------------------------------------------------------------------------- */
/* K2 bitonic_sort, assets/bitonic_sort.wgsl:22-46 (one compare-exchange stage) */
/* ------------------------------------------------------------------------- */
void wso_bitonic_sort_stage(wso_state *s, uint32_t block, uint32_t dim)
{
    const uint32_t n = s->num_particles;
    uint32_t *idx = s->particle_indicies;
    const uint32_t *key = s->particle_cell_indicies;
#pragma omp parallel for schedule(static)
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t j = i ^ block;
        if (j < i) continue;  /* `j < i || i >= num_particles` */
        if (j >= n) continue; /* reference reads out of bounds here; N must be a power of two */
        int32_t sign = 1;
        if ((i & dim) != 0) sign = -1;
        const uint32_t key_i = idx[i];
        const uint32_t key_j = idx[j];
        const uint32_t value_i = key[key_i];
        const uint32_t value_j = key[key_j];
        const int32_t diff = (int32_t)(value_i - value_j) * sign;
        if (diff > 0) {
            idx[i] = key_j;
            idx[j] = key_i;
        }
    }
}

/* src/fluid_compute.rs:256-271,320-330: every stage, in table order, every step,
 * applied to the permutation persisted from the previous step. */
void wso_sort_exact(wso_state *s)
{
    uint64_t p = 1;
    while (p < s->num_particles) p <<= 1;
    for (uint64_t dim = 2; dim <= p; dim <<= 1)
        for (uint64_t block = dim >> 1; block > 0; block >>= 1)
            wso_bitonic_sort_stage(s, (uint32_t)block, (uint32_t)dim);
}

/* Timing mode ("OpenMP over particles and a parallel sort"): a stable LSD radix sort of
 * the persisted permutation by key, 11 bits per pass, each pass a parallel counting sort (per-thread
 * histograms over contiguous chunks, so equal keys keep their order).  Gives the same sorted key
 * sequence and cell offsets as the network; the order inside a bucket (hence float summation order)
 * differs -- it is the order a stable sort of the previous permutation gives. */
#define WSO_RADIX_BITS 11
#define WSO_RADIX (1u << WSO_RADIX_BITS)
void wso_sort_fast(wso_state *s)
{
    const uint32_t n = s->num_particles;
    if (n < 2) return;
    const uint32_t *key = s->particle_cell_indicies;
    uint32_t *a = s->particle_indicies;
    uint32_t *b = (uint32_t *)malloc((size_t)n * sizeof(uint32_t));
    int bits = 0;
    while (((uint64_t)1 << bits) < n) bits++; /* keys are < n */
    const int nthreads = wso_max_threads();
    uint32_t *hist = (uint32_t *)malloc((size_t)nthreads * WSO_RADIX * sizeof(uint32_t));
    for (int shift = 0; shift < bits; shift += WSO_RADIX_BITS) {
#pragma omp parallel num_threads(nthreads)
        {
#ifdef _OPENMP
            const int t = omp_get_thread_num(), T = omp_get_num_threads();
#else
            const int t = 0, T = 1;
#endif
            uint32_t *h = hist + (size_t)t * WSO_RADIX;
            memset(h, 0, WSO_RADIX * sizeof(uint32_t));
            const uint32_t lo = (uint32_t)((uint64_t)n * t / T), hi = (uint32_t)((uint64_t)n * (t + 1) / T);
            for (uint32_t i = lo; i < hi; i++) h[(key[a[i]] >> shift) & (WSO_RADIX - 1u)]++;
#pragma omp barrier
#pragma omp single
            {
                uint32_t run = 0; /* digit-major, thread-minor: stable */
                for (uint32_t d = 0; d < WSO_RADIX; d++)
                    for (int q = 0; q < T; q++) {
                        uint32_t *c = hist + (size_t)q * WSO_RADIX + d;
                        const uint32_t v = *c;
                        *c = run;
                        run += v;
                    }
            }
            for (uint32_t i = lo; i < hi; i++) {
                const uint32_t pid = a[i];
                b[h[(key[pid] >> shift) & (WSO_RADIX - 1u)]++] = pid;
            }
        }
        uint32_t *sw = a;
        a = b;
        b = sw;
    }
    if (a != s->particle_indicies) {
        memcpy(s->particle_indicies, a, (size_t)n * sizeof(uint32_t));
        b = a;
    }
    free(b);
    free(hist);
}

/* ------------------------------------------------------------------------- */
/* K3 calculate_cell_offsets, assets/bitonic_sort.wgsl:48-59                  */
/* ------------------------------------------------------------------------- */
void wso_calculate_cell_offsets(wso_state *s)
{
    const uint32_t n = s->num_particles;
    const uint32_t *idx = s->particle_indicies, *key = s->particle_cell_indicies;
    /* atomicMin(cell_offsets[key], index) over a sorted sequence = the first slot of each key:
     * when the keys are in ascending slot order (always, after either sort) every head slot can be
     * written independently; otherwise fall back to the literal serial minimum. */
    int sorted = 1;
#pragma omp parallel for schedule(static) reduction(&& : sorted)
    for (uint32_t index = 1; index < n; index++) sorted = sorted && key[idx[index - 1]] <= key[idx[index]];
    if (sorted) {
#pragma omp parallel for schedule(static)
        for (uint32_t index = 0; index < n; index++) {
            const uint32_t cell_index = key[idx[index]];
            if (index == 0 || key[idx[index - 1]] != cell_index) s->cell_offsets[cell_index] = index;
        }
        return;
    }
    for (uint32_t index = 0; index < n; index++) {
        const uint32_t cell_index = key[idx[index]];
        if (index < s->cell_offsets[cell_index]) s->cell_offsets[cell_index] = index; /* atomicMin */
    }
}

/* ------------------------------------------------------------------------- */
/* smoothing kernels, assets/simulation.wgsl:93-117                           */
/* ------------------------------------------------------------------------- */
static inline float smoothing_kernel(const wso_state *s, float dst)
{
    const float v = s->props.smoothing_radius - dst;
    return v * v * s->kernel.pow2;
}
static inline float smoothing_kernel_near(const wso_state *s, float dst)
{
    const float v = s->props.smoothing_radius - dst;
    return v * v * v * s->kernel.pow3;
}
static inline float smoothing_kernel_derivative(const wso_state *s, float dst)
{
    return (dst - s->props.smoothing_radius) * s->kernel.pow2_der;
}
static inline float smoothing_kernel_derivative_near(const wso_state *s, float dst)
{
    const float v = dst - s->props.smoothing_radius;
    return v * v * s->kernel.pow3_der;
}
static inline float smoothing_kernel_viscosity(const wso_state *s, float dst)
{
    const float v = s->props.smoothing_radius * s->props.smoothing_radius - dst * dst;
    return v * v * v * s->kernel.spikey_pow3;
}

/* distance(a, b) on vec4: sqrt(dot(a-b, a-b)), summed x,y,z,w left to right. */
static inline float distance4(const float a[4], const float b[4])
{
    const float dx = a[0] - b[0];
    const float dy = a[1] - b[1];
    const float dz = a[2] - b[2];
    const float dw = a[3] - b[3];
    return sqrtf(dx * dx + dy * dy + dz * dz + dw * dw);
}

/* The candidate walk shared by K4 and K5 (simulation.wgsl:162-173 / :219-230):
 * for the 27 offsets in table order, start at cell_offsets[hash] and advance
 * while the slot's particle still carries that hash.  Emits particle ids in visit
 * order (reversed as a whole if s->reverse_order). */
typedef struct {
    uint32_t *v;
    size_t len, cap;
} cand_list;

static void cand_push(cand_list *c, uint32_t x)
{
    if (c->len == c->cap) {
        c->cap = c->cap ? 2 * c->cap : 128;
        c->v = (uint32_t *)realloc(c->v, c->cap * sizeof(uint32_t));
    }
    c->v[c->len++] = x;
}

static void collect_candidates(const wso_state *s, const int32_t cell_index[3], cand_list *out)
{
    const uint32_t n = s->num_particles;
    out->len = 0;
    for (int i = 0; i < 27; i++) {
        const int32_t nc[3] = {cell_index[0] + OFFSET_TABLE[i][0], cell_index[1] + OFFSET_TABLE[i][1],
                               cell_index[2] + OFFSET_TABLE[i][2]};
        const uint32_t hash_index = wso_hash_cell(nc, n);
        uint32_t it = s->cell_offsets[hash_index];
        while (it < n) {
            const uint32_t nidx = s->particle_indicies[it];
            if (s->particle_cell_indicies[nidx] != hash_index) break;
            it++;
            cand_push(out, nidx);
        }
    }
    if (s->reverse_order) {
        for (size_t a = 0, b = out->len; a + 1 < b; a++, b--) {
            const uint32_t t = out->v[a];
            out->v[a] = out->v[b - 1];
            out->v[b - 1] = t;
        }
    }
}

/* ------------------------------------------------------------------------- */
/* K4 update_density, assets/simulation.wgsl:143-195                          */
/* ------------------------------------------------------------------------- */
void wso_update_density(wso_state *s)
{
    const uint32_t n = s->num_particles;
    const float h = s->props.smoothing_radius;
    const float DENSITY_PADDING = 0.00001f; /* simulation.wgsl:4 */
#pragma omp parallel
    {
        cand_list cl = {0, 0, 0};
#pragma omp for schedule(dynamic, 256)
        for (uint32_t index = 0; index < n; index++) {
            const uint32_t pid = s->particle_indicies[index];
            const float *origin = s->particles[pid].predicted_position;
            int32_t cell_index[3];
            wso_get_cell(origin, h, cell_index);

            float density = 0.0f;
            float near_density = 0.0f;
            collect_candidates(s, cell_index, &cl);
            for (size_t c = 0; c < cl.len; c++) {
                const wso_particle *nb = &s->particles[cl.v[c]];
                const float dst = distance4(nb->predicted_position, origin);
                if (dst > h) continue;
                density += smoothing_kernel(s, dst);
                near_density += smoothing_kernel_near(s, dst);
            }
            density = density + DENSITY_PADDING;
            near_density = near_density + DENSITY_PADDING;
            s->particles[pid].density[0] = density;
            s->particles[pid].density[1] = near_density;
            const float pressure = s->props.pressure_scalar * (density - s->props.target_density);
            const float near_pressure = s->props.near_pressure_scalar * near_density;
            s->particles[pid].pressure[0] = pressure;
            s->particles[pid].pressure[1] = near_pressure;
        }
        free(cl.v);
    }
}

/* ------------------------------------------------------------------------- */
/* K5 update_pressure_force, assets/simulation.wgsl:197-269                   */
/* ------------------------------------------------------------------------- */
void wso_update_pressure_force(wso_state *s)
{
    const uint32_t n = s->num_particles;
    const float h = s->props.smoothing_radius;
#pragma omp parallel
    {
        cand_list cl = {0, 0, 0};
#pragma omp for schedule(dynamic, 256)
        for (uint32_t index = 0; index < n; index++) {
            const uint32_t pid = s->particle_indicies[index];
            const wso_particle *self = &s->particles[pid];
            const float *origin = self->predicted_position;
            const float *velocity = self->velocity;
            const float pressure = self->pressure[0];
            const float near_pressure = self->pressure[1];
            int32_t cell_index[3];
            wso_get_cell(origin, h, cell_index);

            float pf[3] = {0.0f, 0.0f, 0.0f};
            float vf[3] = {0.0f, 0.0f, 0.0f};
            collect_candidates(s, cell_index, &cl);
            for (size_t c = 0; c < cl.len; c++) {
                const uint32_t nidx = cl.v[c];
                if (pid == nidx) continue; /* :232, before the distance test */
                const wso_particle *nb = &s->particles[nidx];
                const float dst = distance4(nb->predicted_position, origin);
                if (dst > h) continue;
                float dir[3];
                for (int k = 0; k < 3; k++) dir[k] = nb->predicted_position[k] - origin[k];
                if (dst > 0.0f) {
                    for (int k = 0; k < 3; k++) dir[k] = dir[k] / dst;
                } else {
                    dir[0] = 0.0f;
                    dir[1] = 1.0f;
                    dir[2] = 0.0f;
                }
                const float slope = smoothing_kernel_derivative(s, dst);
                const float shared_pressure = (pressure + nb->pressure[0]) / 2.0f;
                const float slope_near = smoothing_kernel_derivative_near(s, dst);
                const float shared_pressure_near = (near_pressure + nb->pressure[1]) / 2.0f;
                for (int k = 0; k < 3; k++)
                    pf[k] += dir[k] * shared_pressure * slope / nb->density[0];
                for (int k = 0; k < 3; k++)
                    pf[k] += dir[k] * shared_pressure_near * slope_near / nb->density[1];
                const float viscosity = smoothing_kernel_viscosity(s, dst);
                for (int k = 0; k < 3; k++) vf[k] += (nb->velocity[k] - velocity[k]) * viscosity;
            }
            wso_particle *out = &s->particles[pid];
            for (int k = 0; k < 3; k++) {
                const float pressure_contribution = pf[k] / self->density[0];
                const float viscosity_contribution = vf[k] * s->props.viscosity_strength;
                out->acceleration[k] = pressure_contribution + viscosity_contribution;
            }
            out->acceleration[3] = 0.0f;
        }
        free(cl.v);
    }
}

/* ------------------------------------------------------------------------- */
/* K6 integrate, assets/simulation.wgsl:271-310.  Indexed by particle id.     */
/* ------------------------------------------------------------------------- */
void wso_integrate(wso_state *s)
{
    const uint32_t n = s->num_particles;
    const float dt = s->props.delta_time;
    const float damping = s->props.collision_damping;
    const float LOOKAHEAD_FACTOR = (float)(1.0 / 50.0); /* simulation.wgsl:3 */
#pragma omp parallel for schedule(static)
    for (uint32_t index = 0; index < n; index++) {
        wso_particle *p = &s->particles[index];
        for (int k = 0; k < 4; k++) p->velocity[k] += (s->gravity[k] + p->acceleration[k]) * dt;
        for (int k = 0; k < 4; k++) p->position[k] += p->velocity[k] * dt;
        for (int k = 0; k < 3; k++) {
            if (p->position[k] < s->ext_min[k]) {
                p->velocity[k] *= -1.0f * damping;
                p->position[k] = s->ext_min[k];
            } else if (p->position[k] > s->ext_max[k]) {
                p->velocity[k] *= -1.0f * damping;
                p->position[k] = s->ext_max[k];
            }
        }
        for (int k = 0; k < 4; k++)
            p->predicted_position[k] = p->position[k] + p->velocity[k] * LOOKAHEAD_FACTOR;
    }
}

/* Pass order of src/fluid_compute.rs:309-363. */
void wso_step(wso_state *s, int sort_mode)
{
    wso_hash_particles(s);
    if (sort_mode == WSO_SORT_EXACT)
        wso_sort_exact(s);
    else
        wso_sort_fast(s);
    wso_calculate_cell_offsets(s);
    wso_update_density(s);
    wso_update_pressure_force(s);
    wso_integrate(s);
}

int wso_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void wso_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
