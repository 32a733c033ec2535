/*
 * ws_oracle.h -- CPU restatement of the qts8n/water-sandbox SPH fluid step.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity oracle: a plain-C restatement of
 * the reference's six WGSL compute passes and the host-side input generators.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * it.  The product path (water-sandbox_amd/csrc) never links or calls it.
 *
 * PARITY UNPINNED BY THE REFERENCE: the reference ships no tests, no golden
 * vectors and no CPU physics path, and its toolchain
 * (Rust + wgpu) is absent here, so nothing in this file could be checked against
 * the reference running.  Every function cites the reference lines it restates;
 * the known-answer values in tests/test_oracle_kat.py are derived from those
 * source lines, not from executing the reference.
 *
 * Arithmetic semantics fixed here (WGSL leaves them open): IEEE-754 binary32 for
 * every operation, no FMA contraction (-ffp-contract=off), evaluation order as
 * written in the WGSL, correctly rounded sqrt and divide, distance() summed
 * x,y,z,w left to right, i32<->u32 conversions are bit reinterpretations.
 */
#ifndef WS_ORACLE_H
#define WS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WSO_INF 999999999u           /* assets/simulation.wgsl:36 */
#define WSO_P1 15823u                /* assets/simulation.wgsl:38 */
#define WSO_P2 9737333u              /* assets/simulation.wgsl:39 */
#define WSO_P3 440817757u            /* assets/simulation.wgsl:40 */

/* src/fluid_compute.rs:106-115 == assets/simulation.wgsl:69-76 (80 bytes) */
typedef struct {
    float position[4];
    float density[2];            /* x = density, y = near density */
    float pressure[2];           /* x = pressure, y = near pressure */
    float velocity[4];
    float acceleration[4];
    float predicted_position[4];
} wso_particle;

/* src/fluid_compute.rs:41-51 */
typedef struct {
    float delta_time;
    float collision_damping;
    float smoothing_radius;
    float target_density;
    float pressure_scalar;
    float near_pressure_scalar;
    float viscosity_strength;
} wso_props;

/* src/fluid_compute.rs:30-38 */
typedef struct {
    float pow2;
    float pow2_der;
    float pow3;
    float pow3_der;
    float spikey_pow3;
} wso_kernel;

/* src/fluid_compute.rs:88-93 */
typedef struct {
    uint32_t block;
    uint32_t dim;
} wso_bit_sorter;

/* The worker's buffer set, src/fluid_compute.rs:299-308. All pointers are owned
 * by the caller. */
typedef struct {
    uint32_t num_particles;
    wso_props props;
    wso_kernel kernel;
    float ext_min[4];            /* FluidContainerExt, src/fluid_container.rs:19-22 */
    float ext_max[4];
    float gravity[4];            /* src/gravity.rs:11-13 */
    wso_particle *particles;     /* "particles" */
    uint32_t *particle_indicies; /* "particle_indicies" (sic) */
    uint32_t *particle_cell_indicies;
    uint32_t *cell_offsets;
    int reverse_order;           /* 0 = neighbour order as written; 1 = reversed
                                    (used only to calibrate reorder noise) */
} wso_state;

enum { WSO_SORT_EXACT = 0, WSO_SORT_FAST = 1 };

/* ---- host-side generators ------------------------------------------------ */
void wso_default_props(wso_props *out);                           /* fluid_compute.rs:20-27,67-79 */
void wso_smoothing_kernel(const wso_props *p, wso_kernel *out);   /* fluid_compute.rs:55-63 */
void wso_cube_fluid(uint32_t ni, uint32_t nj, uint32_t nk, float particle_rad,
                    float *out_xyz);                              /* helpers.rs:3-20 */
void wso_get_ext(const float position[3], const float size[3], float padding,
                 float ext_min[4], float ext_max[4]);             /* fluid_container.rs:42-50 */
void wso_default_gravity(float g[4]);                             /* gravity.rs:6,29-33 */
uint32_t wso_bit_sorter_stages(uint32_t data_length, wso_bit_sorter *out,
                               uint32_t cap);                     /* fluid_compute.rs:251-273 */
void wso_make_particles(const float *xyz, uint32_t n, wso_particle *out); /* fluid_compute.rs:118-130 */
void wso_identity(uint32_t n, uint32_t *out);                     /* fluid_compute.rs:243-249 */
/* Uniform-cloud generator of SURVEY.md 8(d) (B): counter-based splitmix64. */
void wso_uniform_cloud(uint32_t n, uint64_t seed, const float ext_min[4],
                       const float ext_max[4], float *out_xyz);

/* ---- the cell/hash helpers ------------------------------------------------ */
void wso_get_cell(const float pos[3], float h, int32_t cell[3]);  /* simulation.wgsl:121-123 */
uint32_t wso_hash_cell(const int32_t cell[3], uint32_t n);        /* simulation.wgsl:125-128 */

/* ---- the six passes -------------------------------------------------------- */
void wso_hash_particles(wso_state *s);                            /* K1 simulation.wgsl:130-141 */
void wso_bitonic_sort_stage(wso_state *s, uint32_t block, uint32_t dim); /* K2 bitonic_sort.wgsl:22-46 */
void wso_sort_exact(wso_state *s);                                /* K2 x S, fluid_compute.rs:320-330 */
void wso_sort_fast(wso_state *s);                                 /* any stable sort (timing mode) */
void wso_calculate_cell_offsets(wso_state *s);                    /* K3 bitonic_sort.wgsl:48-59 */
void wso_update_density(wso_state *s);                            /* K4 simulation.wgsl:143-195 */
void wso_update_pressure_force(wso_state *s);                     /* K5 simulation.wgsl:197-269 */
void wso_integrate(wso_state *s);                                 /* K6 simulation.wgsl:271-310 */

/* One step in the pass order of src/fluid_compute.rs:309-363. */
void wso_step(wso_state *s, int sort_mode);

int wso_max_threads(void);
void wso_set_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
