"""MI355X-native SPH fluid step of qts8n/water-sandbox: HIP kernels + C ABI (csrc/,
libwsfluid.so) and the host-side mirror of the reference's fluid worker interface.

The directory name carries a hyphen; import it as `water_sandbox_amd` through the shim module
at the repository root."""
from . import build, fluid, slab, workloads  # noqa: F401
from .fluid import (  # noqa: F401
    PARTICLE_DTYPE,
    FluidWorker,
    WsError,
    WsParams,
    cube_fluid,
    default_params,
    get_ext,
    get_smoothing_kernel,
    load_library,
    make_params,
)
