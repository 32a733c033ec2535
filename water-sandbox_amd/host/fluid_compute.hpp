// fluid_compute.hpp -- header-only C++17 mirror of the reference's fluid host interface
// (src/fluid_compute.rs, src/fluid_container.rs, src/gravity.rs, src/helpers.rs) over the C ABI
// of include/wsfluid.h.  The reference's host is Rust; no Rust toolchain exists in this image, so
// this is the compiled host side that is built and exercised here (rust/fluid_compute.rs is the
// source-only Bevy shim).  Names and argument meaning follow the reference.
#pragma once

#include <array>
#include <stdexcept>
#include <string>
#include <vector>

#include "wsfluid.h"

namespace water_sandbox {

struct Vec3 {
    float x, y, z;
};

// src/fluid_compute.rs:41-51,:67-79
struct FluidStaticProps {
    float delta_time = 1.0f / 60.0f;
    float collision_damping = 0.95f;
    float smoothing_radius = 0.25f;
    float target_density = 10.0f;
    float pressure_scalar = 22.0f;
    float near_pressure_scalar = 2.0f;
    float viscosity_strength = 0.1f;

    // :55-63
    ws_smoothing_kernel get_smoothing_kernel() const
    {
        ws_params p{};
        p.smoothing_radius = smoothing_radius;
        ws_smoothing_kernel k{};
        ws_get_smoothing_kernel(&p, &k);
        return k;
    }
};

// src/gravity.rs:9-33
struct Gravity {
    std::array<float, 4> value{0.f, -9.8f, 0.f, 0.f};
    void set_zero() { value = {0.f, 0.f, 0.f, 0.f}; }
    void set_default() { value = {0.f, -9.8f, 0.f, 0.f}; }
};

// src/fluid_container.rs:17-51
struct FluidContainerExt {
    std::array<float, 4> ext_min, ext_max;
};

struct FluidContainer {
    std::array<float, 3> position{0.f, 0.f, 0.f};
    std::array<float, 3> size{16.f, 9.f, 9.f};
    FluidContainerExt get_ext(float padding) const
    {
        FluidContainerExt e{};
        ws_get_ext(position.data(), size.data(), padding, e.ext_min.data(), e.ext_max.data());
        return e;
    }
};

// src/helpers.rs:3-20
inline std::vector<Vec3> cube_fluid(unsigned ni, unsigned nj, unsigned nk, float particle_rad)
{
    std::vector<Vec3> pts((size_t)ni * nj * nk);
    ws_cube_fluid(ni, nj, nk, particle_rad, reinterpret_cast<float *>(pts.data()));
    return pts;
}

using FluidParticle = ws_particle80;  // src/fluid_compute.rs:106-115

struct WsError : std::runtime_error {
    ws_status status;
    WsError(ws_status s, const std::string &what) : std::runtime_error(what), status(s) {}
};

// The role of AppComputeWorker<FluidWorker>, src/fluid_compute.rs:239-366.
class FluidWorker {
   public:
    static constexpr float PARTICLE_RADIUS = 0.1f;  // :20

    // FluidWorker::build, :277-366
    static FluidWorker build(const FluidStaticProps &props, const Gravity &gravity, const FluidContainer &container,
                             const std::vector<Vec3> &points, int device = 0)
    {
        check_abi();
        FluidWorker w;
        w.n_ = (uint32_t)points.size();
        const ws_params p = make_params(props, gravity, container);
        ws_device_cfg cfg{};
        cfg.device = device;
        const ws_status st = ws_create(&p, reinterpret_cast<const float *>(points.data()), w.n_, &cfg, &w.h_);
        if (st != WS_OK) throw WsError(st, ws_last_error(nullptr));
        return w;
    }

    // (a library of another ABI version lays ws_transport / ws_device_cfg out differently: refuse it before any struct crosses)
    static void check_abi()
    {
        if (ws_abi_version() != WS_ABI_VERSION)
            throw WsError(WS_ERR_UNSUPPORTED, "libwsfluid.so speaks another ABI version than include/wsfluid.h of this build");
    }

    // The same worker as ONE x-slab of the domain (multi-GPU: one per GPU; the reference is single-GPU).
    // Every rank is handed ALL points -- FluidParticlesInitial, :82-85 -- and keeps what ws_slab_assign gives it.
    // n_ stays the GLOBAL particle count: read_positions / read_vec / reset / write_slice work on global, id-ordered
    // arrays on a slab handle too (collective calls: every rank makes them at the same point of its frame).
    static FluidWorker build_slab(const FluidStaticProps &props, const Gravity &gravity, const FluidContainer &container,
                                  const std::vector<Vec3> &points, unsigned rank, unsigned world, const ws_transport &transport,
                                  int device = 0, unsigned flags = 0)
    {
        check_abi();
        FluidWorker w;
        w.n_ = (uint32_t)points.size();
        const ws_params p = make_params(props, gravity, container);
        std::vector<uint32_t> owner(points.size());
        ws_status st = ws_slab_assign(&p, reinterpret_cast<const float *>(points.data()), w.n_, world, owner.data());
        if (st != WS_OK) throw WsError(st, ws_last_error(nullptr));
        std::vector<Vec3> mine;
        std::vector<uint32_t> ids;
        for (uint32_t i = 0; i < w.n_; i++)
            if (owner[i] == rank) {
                mine.push_back(points[i]);
                ids.push_back(i);
            }
        ws_device_cfg cfg{};
        cfg.device = device;
        cfg.rank = rank;
        cfg.world_size = world;
        cfg.flags = flags;
        st = ws_slab_create(&p, reinterpret_cast<const float *>(mine.data()), ids.data(), (uint32_t)mine.size(), w.n_, &cfg,
                            &transport, &w.h_);
        if (st != WS_OK) throw WsError(st, ws_last_error(nullptr));
        return w;
    }

    FluidWorker(FluidWorker &&o) noexcept : h_(o.h_), n_(o.n_) { o.h_ = nullptr; }
    FluidWorker &operator=(FluidWorker &&o) noexcept
    {
        if (this != &o) {
            ws_destroy(h_);
            h_ = o.h_;
            n_ = o.n_;
            o.h_ = nullptr;
        }
        return *this;
    }
    FluidWorker(const FluidWorker &) = delete;
    ~FluidWorker() { ws_destroy(h_); }

    // AppComputeWorker::run / ready, :396,:474
    void run() { check(ws_step(h_)); }
    void sync() { check(ws_sync(h_)); }
    bool ready()
    {
        int r = 0;
        check(ws_ready(h_, &r));
        return r != 0;
    }
    // worker.read_vec::<FluidParticle>("particles"), :478
    std::vector<FluidParticle> read_vec()
    {
        std::vector<FluidParticle> out(n_);
        check(ws_read_particles(h_, out.data()));
        return out;
    }
    std::vector<Vec3> read_positions()
    {
        std::vector<Vec3> out(n_);
        check(ws_read_positions(h_, reinterpret_cast<float *>(out.data())));
        return out;
    }
    // the same into a buffer the host keeps (page-locked once with pin()), optionally overlapped with the next
    // step: read_positions_begin(buf); run(); read_positions_end();
    void read_positions_into(std::vector<Vec3> &buf)
    {
        buf.resize(n_);
        check(ws_read_positions(h_, reinterpret_cast<float *>(buf.data())));
    }
    void read_positions_begin(std::vector<Vec3> &buf)
    {
        if (buf.size() != n_) throw WsError(WS_ERR_INVALID_ARG, "read_positions_begin: wrong buffer size");
        check(ws_read_positions_begin(h_, reinterpret_cast<float *>(buf.data())));
    }
    void read_positions_end() { check(ws_read_positions_end(h_)); }
    // ... or into the library's own page-locked double buffer (no buffer of the host's to pin): after _end, view() is the
    // frame just read and stays untouched while the next frame's copy is in flight
    void read_positions_begin_owned() { check(ws_read_positions_begin(h_, nullptr)); }
    const Vec3 *read_positions_view()
    {
        const float *p = nullptr;
        check(ws_read_positions_view(h_, &p));
        return reinterpret_cast<const Vec3 *>(p);
    }
    void pin(std::vector<Vec3> &buf) { check(ws_pin_host_buffer(h_, buf.data(), buf.size() * sizeof(Vec3))); }
    void unpin(std::vector<Vec3> &buf) { check(ws_unpin_host_buffer(h_, buf.data())); }
    // velocities.length() per particle: the input of update_particle_color, :489-502
    std::vector<float> read_speeds()
    {
        std::vector<float> out(n_);
        check(ws_read_speeds(h_, out.data()));
        return out;
    }
    // the three worker.write calls of update(), :479-481
    void write(const FluidStaticProps &props, const Gravity &gravity, const FluidContainer &container)
    {
        const ws_params p = make_params(props, gravity, container);
        check(ws_set_params(h_, &p));
    }
    // worker.write_slice("particles", ..), :521
    void write_slice(const std::vector<FluidParticle> &particles)
    {
        if (particles.size() != n_) throw WsError(WS_ERR_INVALID_ARG, "write_slice: wrong particle count");
        check(ws_write_particles(h_, particles.data()));
    }
    // despawn_liquid's reset, :517-524
    void reset(const std::vector<Vec3> &points)
    {
        if (points.size() != n_) throw WsError(WS_ERR_INVALID_ARG, "reset: wrong particle count");
        check(ws_reset(h_, reinterpret_cast<const float *>(points.data())));
    }
    uint32_t num_particles() const { return n_; }
    ws_handle *raw() { return h_; }

    static ws_params make_params(const FluidStaticProps &props, const Gravity &gravity, const FluidContainer &container)
    {
        ws_params p{};
        p.delta_time = props.delta_time;
        p.collision_damping = props.collision_damping;
        p.smoothing_radius = props.smoothing_radius;
        p.target_density = props.target_density;
        p.pressure_scalar = props.pressure_scalar;
        p.near_pressure_scalar = props.near_pressure_scalar;
        p.viscosity_strength = props.viscosity_strength;
        const FluidContainerExt e = container.get_ext(PARTICLE_RADIUS);  // :302
        for (int i = 0; i < 4; i++) {
            p.gravity[i] = gravity.value[i];
            p.ext_min[i] = e.ext_min[i];
            p.ext_max[i] = e.ext_max[i];
        }
        return p;
    }

   private:
    FluidWorker() = default;
    void check(ws_status st)
    {
        if (st != WS_OK) throw WsError(st, ws_last_error(h_));
    }
    ws_handle *h_ = nullptr;
    uint32_t n_ = 0;
};

}  // namespace water_sandbox
