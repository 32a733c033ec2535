// frame_loop.cpp -- drives the host mirror in the reference's per-frame order
// (src/schedule.rs:24-36): Update { despawn_liquid -> (HUD edits) -> update }, PostUpdate { run }.
// Usage: frame_loop [frames] [ni nj nk]            one handle; prints one status line per 10 frames
//        frame_loop --slabs W [frames] [ni nj nk]  the same frame loop on ONE handle and on W x-slabs (one host thread
//                                                  per slab, the library's in-process transport) -- every frame's
//                                                  id-ordered positions must be bit-identical
// Needs an MI355X.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "fluid_compute.hpp"

using namespace water_sandbox;

// One frame of the app, against either kind of worker.  Returns false when update() skipped the frame (!ready()).
// `wait`: a multi-slab run must make the same calls on every rank at every frame, so it waits for the previous step
// instead of skipping the frame on !ready() (a single-GPU app skips, as the reference does, :474-476).
static bool frame(FluidWorker &worker, int f, const std::vector<Vec3> &initial, FluidStaticProps &props, Gravity &gravity,
                  const FluidContainer &container, std::vector<Vec3> &translation, bool wait)
{
    if (wait) worker.sync();
    // Update / DespawnEntities: despawn_liquid (Space pressed on frame 30 in this demo)
    if (f == 30 && worker.ready()) worker.reset(initial);
    // Update / UserInput: the HUD may edit props / gravity (hud.rs:130-165)
    if (f == 20) props.smoothing_radius += 0.1f;  // KeyR: the cell grid is rebuilt
    if (f == 40) props.smoothing_radius -= 0.1f;
    if (f == 45) gravity.set_zero();
    // Update / EntityUpdates: update()
    bool updated = false;
    if (worker.ready()) {
        worker.read_positions_into(translation);
        worker.write(props, gravity, container);
        updated = true;
    }
    // PostUpdate / Pass: AppComputeWorker::run
    worker.run();
    return updated;
}

static int run_single(int frames, unsigned ni, unsigned nj, unsigned nk)
{
    FluidStaticProps props;    // init_resource::<FluidStaticProps>()
    Gravity gravity;           // GravityPlugin
    FluidContainer container;  // GizmoPlugin
    const std::vector<Vec3> initial = cube_fluid(ni, nj, nk, FluidWorker::PARTICLE_RADIUS);  // FluidParticlesInitial
    FluidWorker worker = FluidWorker::build(props, gravity, container, initial);
    std::vector<Vec3> translation(initial);  // the entities' Transform.translation, by FluidParticleLabel
    int skipped = 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (int f = 0; f < frames; f++) {
        if (!frame(worker, f, initial, props, gravity, container, translation, false)) skipped++;
        // the app renders here (~16 ms at 60 Hz); a short pause lets the asynchronous step finish so
        // that the next frame's update() finds ready() true, as it does in the real frame loop
        std::this_thread::sleep_for(std::chrono::milliseconds(3));
        if (f % 10 == 9) std::printf("frame %3d  y[0] = %.6f\n", f, translation[0].y);
    }
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("%d frames, %u particles, %.1f frames/s, %d frames skipped on !ready()\n", frames, worker.num_particles(),
                frames / s, skipped);
    return 0;
}

static int run_slabs(unsigned world, int frames, unsigned ni, unsigned nj, unsigned nk)
{
    const FluidContainer container;
    const std::vector<Vec3> initial = cube_fluid(ni, nj, nk, FluidWorker::PARTICLE_RADIUS);
    // the single handle: every frame's positions
    std::vector<std::vector<Vec3>> want(frames);
    {
        FluidStaticProps props;
        Gravity gravity;
        gravity.value[0] = 6.f;  // tilted: fluid crosses the slab faces
        FluidWorker worker = FluidWorker::build(props, gravity, container, initial);
        std::vector<Vec3> translation(initial);
        for (int f = 0; f < frames; f++) {
            frame(worker, f, initial, props, gravity, container, translation, true);
            want[f] = translation;
        }
    }
    void *hub = nullptr;
    if (ws_local_hub_create(world, &hub) != WS_OK) return 3;
    std::vector<int> bad(world, 0);
    std::vector<std::string> errors(world);
    std::vector<std::thread> threads;
    for (unsigned r = 0; r < world; r++)
        threads.emplace_back([&, r] {
            ws_transport tr{};
            ws_local_transport_create(hub, r, &tr);
            try {
                FluidStaticProps props;
                Gravity gravity;
                gravity.value[0] = 6.f;
                FluidWorker worker = FluidWorker::build_slab(props, gravity, container, initial, r, world, tr);
                std::vector<Vec3> translation(initial);
                for (int f = 0; f < frames; f++) {
                    frame(worker, f, initial, props, gravity, container, translation, true);
                    if (std::memcmp(translation.data(), want[f].data(), translation.size() * sizeof(Vec3)) != 0) bad[r]++;
                }
            } catch (const WsError &e) {
                errors[r] = e.what();
                bad[r] = -1;
            }
            ws_local_transport_destroy(&tr);
        });
    for (auto &t : threads) t.join();
    ws_local_hub_destroy(hub);
    int rc = 0;
    for (unsigned r = 0; r < world; r++) {
        if (bad[r] < 0) std::fprintf(stderr, "slab %u: wsfluid error: %s\n", r, errors[r].c_str());
        else if (bad[r]) std::fprintf(stderr, "slab %u: %d of %d frames differ from the single handle\n", r, bad[r], frames);
        if (bad[r]) rc = 4;
    }
    if (!rc)
        std::printf("%u slabs: %d frames (radius change at 20 and 40, reset at 30, gravity off at 45), %zu particles: every "
                    "frame's id-ordered positions bit-identical to the single handle on every rank\n",
                    world, frames, initial.size());
    return rc;
}

int main(int argc, char **argv)
{
    unsigned slabs = 0;
    int a = 1;
    if (argc > 2 && std::strcmp(argv[1], "--slabs") == 0) {
        slabs = (unsigned)atoi(argv[2]);
        a = 3;
    }
    const int frames = argc > a ? atoi(argv[a]) : 60;
    const bool dims = argc > a + 3;
    const unsigned ni = dims ? atoi(argv[a + 1]) : 64, nj = dims ? atoi(argv[a + 2]) : 32, nk = dims ? atoi(argv[a + 3]) : 32;
    try {
        return slabs ? run_slabs(slabs, frames, ni, nj, nk) : run_single(frames, ni, nj, nk);
    } catch (const WsError &e) {
        std::fprintf(stderr, "wsfluid error %d: %s\n", (int)e.status, e.what());
        return 2;
    }
}
