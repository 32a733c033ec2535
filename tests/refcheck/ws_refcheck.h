// ws_refcheck.h -- TEST-ONLY declarations of the reference-order validation mode (WS_FLAG_REFERENCE_ORDER).
// Included by csrc/ws_internal.h only when the sources are built with -DWS_WITH_REFCHECK, i.e. only into
// tests/libwsfluid_refcheck.so; libwsfluid.so contains none of this (tests/test_abi.py checks the binary).
#pragma once

// the reference's own buffer set, by particle id (src/fluid_compute.rs:299-308)
struct WsRef {
    float4 *pos = nullptr, *vel = nullptr, *pred = nullptr, *acc = nullptr;  // w = 0
    float2 *dens = nullptr;    // (density, near density)
    uint32_t *perm = nullptr;  // particle_indicies
    uint32_t *keys = nullptr;  // particle_cell_indicies (by particle id)
    uint32_t *offs = nullptr;  // cell_offsets
};

struct WsDev;
void wsk_ref_step(hipStream_t s, const WsDev &d, WsRef r);
void wsk_ref_load(hipStream_t s, const ws_particle80 *in_dev, WsRef r, uint32_t n, bool reset_index);
void wsk_ref_store(hipStream_t s, const WsDev &d, WsRef r, ws_particle80 *out_dev, uint32_t n);
