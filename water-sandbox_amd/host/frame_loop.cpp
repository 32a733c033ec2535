// frame_loop.cpp -- drives the host mirror in the reference's per-frame order
// (src/schedule.rs:24-36): Update { despawn_liquid -> (HUD edits) -> update }, PostUpdate { run }.
// Usage: frame_loop [frames] [ni nj nk].  Needs an MI355X; prints one status line per 10 frames.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "fluid_compute.hpp"

using namespace water_sandbox;

int main(int argc, char **argv)
{
    const int frames = argc > 1 ? atoi(argv[1]) : 60;
    const unsigned ni = argc > 4 ? atoi(argv[2]) : 64, nj = argc > 4 ? atoi(argv[3]) : 32, nk = argc > 4 ? atoi(argv[4]) : 32;
    FluidStaticProps props;    // init_resource::<FluidStaticProps>()
    Gravity gravity;           // GravityPlugin
    FluidContainer container;  // GizmoPlugin
    const std::vector<Vec3> initial = cube_fluid(ni, nj, nk, FluidWorker::PARTICLE_RADIUS);  // FluidParticlesInitial
    try {
        FluidWorker worker = FluidWorker::build(props, gravity, container, initial);
        std::vector<Vec3> translation(initial);  // the entities' Transform.translation, by FluidParticleLabel
        int skipped = 0;
        const auto t0 = std::chrono::steady_clock::now();
        for (int f = 0; f < frames; f++) {
            // Update / DespawnEntities: despawn_liquid (Space pressed on frame 30 in this demo)
            if (f == 30 && worker.ready()) worker.reset(initial);
            // Update / UserInput: the HUD may edit props/gravity (hud.rs:130-165)
            if (f == 45) gravity.set_zero();
            // Update / EntityUpdates: update()
            if (worker.ready()) {
                translation = worker.read_positions();
                worker.write(props, gravity, container);
            } else {
                skipped++;
            }
            // PostUpdate / Pass: AppComputeWorker::run
            worker.run();
            // the app renders here (~16 ms at 60 Hz); a short pause lets the asynchronous step finish so
            // that the next frame's update() finds ready() true, as it does in the real frame loop
            std::this_thread::sleep_for(std::chrono::milliseconds(3));
            if (f % 10 == 9) std::printf("frame %3d  y[0] = %.6f\n", f, translation[0].y);
        }
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("%d frames, %u particles, %.1f frames/s, %d frames skipped on !ready()\n", frames,
                    worker.num_particles(), frames / s, skipped);
    } catch (const WsError &e) {
        std::fprintf(stderr, "wsfluid error %d: %s\n", (int)e.status, e.what());
        return 2;
    }
    return 0;
}
