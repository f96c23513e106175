"""Loader of the HIP shared library behind ``include/upkie_hip.h``.

The product path has NO CPU fallback: if the library is missing or no HIP
device is visible, calls fail loudly with `UpkieHipError` /
`MissingOptionalDependency`-style messages rather than computing elsewhere.
"""

import ctypes as C
import os
import subprocess

from . import abi
from .exceptions import UpkieRuntimeError

_HERE = os.path.dirname(os.path.abspath(__file__))
# UPKIE_HIP_LIBRARY: another build of the same sources (A/B measurements of a kernel change on one GPU box)
LIB_PATH = os.environ.get("UPKIE_HIP_LIBRARY") or os.path.join(_HERE, "_lib", "libupkie_hip.so")
SOURCES = [
    os.path.join(_HERE, "csrc", "upkie_hip.hip"),
    os.path.join(_HERE, "csrc", "step_instances.hip"),
    os.path.join(_HERE, "csrc", "step_instances.hpp"),
    os.path.join(_HERE, "csrc", "step_kernels.hpp"),
    os.path.join(_HERE, "csrc", "host_setup.hpp"),
    os.path.join(_HERE, "csrc", "dynamics.hpp"),
    os.path.join(_HERE, "csrc", "bullet_like.hpp"),
    os.path.join(_HERE, "csrc", "state_words.hpp"),
    os.path.join(_HERE, "csrc", "mpc.hpp"),
    os.path.join(_HERE, "csrc", "pair.hpp"),
    os.path.join(_HERE, "csrc", "octet.hpp"),
    os.path.join(_HERE, "csrc", "observers.hpp"),
    os.path.join(_HERE, "csrc", "rollout.hpp"),
    os.path.join(_HERE, "csrc", "wave_io.hpp"),
    os.path.join(_HERE, "..", "include", "upkie_hip.h"),
]

INSTANCES_SOURCE = SOURCES[1]

## Every symbol `include/upkie_hip.h` declares.
EXPORTED_SYMBOLS = (
    "upkie_hip_device_count",
    "upkie_hip_struct_bytes",
    "upkie_sim_create",
    "upkie_sim_destroy",
    "upkie_sim_set_config",
    "upkie_sim_last_error",
    "upkie_sim_state_bytes",
    "upkie_sim_pgs_tolerance",
    "upkie_sim_lanes_per_env",
    "upkie_sim_lanes_per_env_of",
    "upkie_sim_set_census",
    "upkie_sim_set_lanes_per_env",
    "upkie_sim_guard_counts",
    "upkie_sim_release_graph_captures",
    "upkie_sim_set_final_observation",
    "upkie_sim_set_contact_manifold",
    "upkie_sim_set_randomization",
    "upkie_sim_set_external_forces",
    "upkie_sim_sample_body_inertials",
    "upkie_sim_sample_pushes",
    "upkie_sim_contact_sweeps",
    "upkie_sim_reset",
    "upkie_sim_step_pendulum",
    "upkie_sim_step_pendulum_agent",
    "upkie_sim_step_pendulum_packed",
    "upkie_sim_step_pendulum_agent_packed",
    "upkie_sim_step_pendulum_agent_records",
    "upkie_sim_step_pendulum_agent_rollout",
    "upkie_sim_step_gyropod",
    "upkie_sim_step_servos",
    "upkie_sim_servo_policy",
    "upkie_sim_step_servos_policy",
    "upkie_sim_step_base_velocity",
    "upkie_sim_step_base_velocity_mpc",
    "upkie_sim_observe",
    "upkie_sim_contact_points",
    "upkie_sim_autoreset_done",
    "upkie_mpc_create",
    "upkie_mpc_destroy",
    "upkie_mpc_last_error",
    "upkie_mpc_workspace_bytes",
    "upkie_mpc_reset",
    "upkie_mpc_step",
    "upkie_mpc_step_env",
    "upkie_sim_attach_observers",
    "upkie_observers_create",
    "upkie_observers_destroy",
    "upkie_observers_last_error",
    "upkie_observers_state_bytes",
    "upkie_observers_reset",
    "upkie_observers_step",
    "upkie_rollout_gae",
    "upkie_linear_policy",
)


class UpkieHipError(UpkieRuntimeError):
    """Error reported by the HIP library (status code + message)."""

    def __init__(self, status: int, message: str):
        super().__init__(f"upkie_hip status {status}: {message}")
        self.status = status


INSTANCE_GROUPS = 16  # UPKIE_INSTANCE_GROUPS of csrc/step_instances.hpp

HIPCC_FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    # SLP-packing scalar fp32 chains into v_pk_* costs ~1700 v_mov and
    # 180 extra registers in the step kernel (tools/isa_stats.sh)
    "-fno-slp-vectorize",
    # the scheduler's AMDGPU-specific register-pressure trackers: another schedule of the same code, 1.4 % faster
    # step kernel A/B (profiles/r03_ab_scheduler_flags.txt; max-ilp, max-memory-clause, no post-RA: slower)
    "-mllvm",
    "-amdgpu-use-amdgpu-trackers=1",
    "-fPIC",
    # the device code objects compressed inside the fat binary (zstd; the HIP runtime unpacks one when its first kernel is
    # launched): the library is 4.2 MB instead of 8.0 (round 5: VERDICT r4 asked for <= 5 MB)
    "--offload-compress",
]


def build(force: bool = False, verbose: bool = False, jobs: int = 0) -> str:
    """Compile the HIP library for gfx950 with hipcc (in-tree): the C-ABI's
    translation unit (`upkie_hip.hip`: host code and the small kernels) and the
    ~100 step-kernel instantiations in `INSTANCE_GROUPS` groups
    (`step_instances.hip -DUPKIE_INSTANCE_GROUP=g`), compiled side by side on
    `jobs` cores (default: all this process may use), then linked."""
    stale = force or not os.path.exists(LIB_PATH)
    if not stale:
        mtime = os.path.getmtime(LIB_PATH)
        stale = any(
            os.path.exists(s) and os.path.getmtime(s) > mtime for s in SOURCES
        )
    if stale:
        os.makedirs(os.path.dirname(LIB_PATH), exist_ok=True)
        tag = f"{LIB_PATH}.{os.getpid()}"
        # (the heaviest units first -- the one-lane kernels, groups 0-7, then the C-ABI's own unit --: the build ends when the
        # total work does, not when a late-started heavy group does)
        units = [(INSTANCES_SOURCE, [f"-DUPKIE_INSTANCE_GROUP={g}"], f"{tag}.g{g}.o") for g in range(8)]
        units += [(SOURCES[0], [], f"{tag}.abi.o")]
        units += [(INSTANCES_SOURCE, [f"-DUPKIE_INSTANCE_GROUP={g}"], f"{tag}.g{g}.o") for g in range(8, INSTANCE_GROUPS)]
        jobs = jobs or (len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
        partial = f"{tag}.partial"  # (a library is never loadable half-written: built beside it, renamed into place)
        objects = [obj for _, _, obj in units]
        try:
            pending, running, failed = list(units), [], None
            while (pending or running) and failed is None:
                while pending and len(running) < jobs:
                    src, defines, obj = pending.pop(0)
                    cmd = ["hipcc"] + HIPCC_FLAGS + defines + ["-c", src, "-o", obj]
                    running.append((subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True), cmd))
                proc, cmd = running.pop(0)
                out, _ = proc.communicate()
                if verbose or proc.returncode != 0:
                    print(" ".join(cmd))
                    print(out)
                if proc.returncode != 0:
                    failed = cmd
            for proc, _ in running:
                proc.kill()
                proc.communicate()
            if failed is not None:
                raise UpkieRuntimeError("hipcc failed to build libupkie_hip.so: " + " ".join(failed))
            result = subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objects + ["-o", partial], capture_output=True, text=True)
            if verbose or result.returncode != 0:
                print(result.stdout)
                print(result.stderr)
            if result.returncode != 0:
                raise UpkieRuntimeError("hipcc failed to link libupkie_hip.so")
            os.replace(partial, LIB_PATH)
        finally:
            for path in objects + [partial]:
                if os.path.exists(path):
                    os.remove(path)
    return LIB_PATH


_lib = None


def _check_struct_sizes(lib) -> None:
    """The structs this package writes (`abi.py`) must be the structs the library reads: a build of another version of
    the header, loaded through UPKIE_HIP_LIBRARY for an A/B run, may have shorter or longer config structs (they grow
    at the tail) and would read past what ctypes hands over. `upkie_hip_struct_bytes` (round 5) tells; a library
    without it is accepted only as such an override and only with a warning."""
    if not hasattr(lib, "upkie_hip_struct_bytes"):
        if os.environ.get("UPKIE_HIP_LIBRARY"):
            import warnings

            warnings.warn(f"{LIB_PATH} predates upkie_hip_struct_bytes: struct layouts not checked against upkie_amd/abi.py")
            return
        raise UpkieRuntimeError(f"{LIB_PATH} does not export upkie_hip_struct_bytes: rebuild it (`__graft_entry__.build()`)")
    lib.upkie_hip_struct_bytes.restype = C.c_int64
    lib.upkie_hip_struct_bytes.argtypes = [C.c_int]
    for which, cls in abi.STRUCT_IDS.items():
        theirs, ours = int(lib.upkie_hip_struct_bytes(which)), C.sizeof(cls)
        if theirs != ours:
            raise UpkieRuntimeError(
                f"{LIB_PATH} was built from another version of include/upkie_hip.h: sizeof({cls.__name__}) is {theirs} B there, "
                f"{ours} B in upkie_amd/abi.py; rebuild the library or point UPKIE_HIP_LIBRARY at a matching build"
            )


def load() -> C.CDLL:
    """Load the library (never falls back to anything else)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise UpkieRuntimeError(
            f"{LIB_PATH} not found: build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback"
        )
    # PyTorch owns the device memory and streams handed to the library, so both
    # must run on ONE HIP runtime instance: load torch's first and let the
    # dynamic loader resolve libupkie_hip.so's libamdhip64 dependency to it.
    import torch  # noqa: F401

    lib = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    _check_struct_sizes(lib)
    lib.upkie_hip_device_count.restype = C.c_int
    lib.upkie_sim_create.restype = C.c_int
    lib.upkie_sim_create.argtypes = [
        C.POINTER(abi.UpkieSimConfig),
        C.POINTER(abi.UpkieModel),
        C.POINTER(vp),
    ]
    lib.upkie_sim_set_config.restype = C.c_int
    lib.upkie_sim_set_config.argtypes = [vp, C.POINTER(abi.UpkieSimConfig)]
    lib.upkie_sim_destroy.restype = C.c_int
    lib.upkie_sim_destroy.argtypes = [vp]
    lib.upkie_sim_last_error.restype = C.c_char_p
    lib.upkie_sim_last_error.argtypes = [vp]
    lib.upkie_sim_pgs_tolerance.restype = C.c_double
    lib.upkie_sim_pgs_tolerance.argtypes = [vp]
    lib.upkie_sim_state_bytes.restype = C.c_int64
    lib.upkie_sim_state_bytes.argtypes = [vp]
    lib.upkie_sim_lanes_per_env.restype = C.c_int
    lib.upkie_sim_lanes_per_env.argtypes = [vp]
    lib.upkie_sim_set_census.restype = C.c_int
    lib.upkie_sim_set_census.argtypes = [vp, vp]
    lib.upkie_sim_set_lanes_per_env.restype = C.c_int
    lib.upkie_sim_set_lanes_per_env.argtypes = [vp, C.c_int]
    lib.upkie_sim_release_graph_captures.restype = C.c_int
    lib.upkie_sim_release_graph_captures.argtypes = [vp]
    lib.upkie_sim_guard_counts.restype = C.c_int
    lib.upkie_sim_guard_counts.argtypes = [vp, C.POINTER(C.c_uint32), C.c_int, vp]
    lib.upkie_sim_set_final_observation.restype = C.c_int
    lib.upkie_sim_set_final_observation.argtypes = [vp, vp]
    # (round-4 entry points: bound when present, so that tools/ab_step.py can still load an OLDER build of the library
    # through UPKIE_HIP_LIBRARY for an A/B on one box; the shipped library exports them all: tests/test_abi.py)
    if hasattr(lib, "upkie_sim_lanes_per_env_of"):
        lib.upkie_sim_lanes_per_env_of.restype = C.c_int
        lib.upkie_sim_lanes_per_env_of.argtypes = [vp, C.c_int]
    if hasattr(lib, "upkie_sim_set_contact_manifold"):
        lib.upkie_sim_set_contact_manifold.restype = C.c_int
        lib.upkie_sim_set_contact_manifold.argtypes = [vp, vp]
    lib.upkie_sim_servo_policy.restype = C.c_int
    lib.upkie_sim_servo_policy.argtypes = [vp, vp, C.POINTER(abi.UpkieServoPolicy), vp, vp]
    lib.upkie_sim_step_servos_policy.restype = C.c_int
    lib.upkie_sim_step_servos_policy.argtypes = [vp, vp, C.POINTER(abi.UpkieServoPolicy), vp, vp, vp, vp, vp, vp]
    lib.upkie_sim_set_randomization.restype = C.c_int
    lib.upkie_sim_set_randomization.argtypes = [vp, vp, vp, C.POINTER(C.c_double)]
    lib.upkie_sim_set_external_forces.restype = C.c_int
    lib.upkie_sim_set_external_forces.argtypes = [vp, vp, C.POINTER(abi.UpkieExternalForces)]
    lib.upkie_sim_sample_body_inertials.restype = C.c_int
    lib.upkie_sim_sample_body_inertials.argtypes = [vp, vp, vp, C.c_double, vp]
    lib.upkie_sim_contact_sweeps.restype = C.c_int
    lib.upkie_sim_contact_sweeps.argtypes = [vp, C.c_int32, vp, vp, vp, vp, vp, vp]
    lib.upkie_sim_sample_pushes.restype = C.c_int
    lib.upkie_sim_sample_pushes.argtypes = [vp, vp, C.c_uint32, C.c_double, vp]
    lib.upkie_sim_reset.restype = C.c_int
    lib.upkie_sim_reset.argtypes = [vp, vp, vp, vp, vp]
    for name in (
        "upkie_sim_step_pendulum",
        "upkie_sim_step_gyropod",
        "upkie_sim_step_servos",
    ):
        fn = getattr(lib, name)
        fn.restype = C.c_int
        fn.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    lib.upkie_sim_step_pendulum_agent.restype = C.c_int
    lib.upkie_sim_step_pendulum_agent.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    lib.upkie_sim_step_pendulum_packed.restype = C.c_int
    lib.upkie_sim_step_pendulum_packed.argtypes = [vp, vp, vp, vp, vp]
    lib.upkie_sim_step_pendulum_agent_packed.restype = C.c_int
    lib.upkie_sim_step_pendulum_agent_packed.argtypes = [vp, vp, vp, vp]
    lib.upkie_sim_step_pendulum_agent_records.restype = C.c_int
    lib.upkie_sim_step_pendulum_agent_records.argtypes = [vp, vp, vp, vp, vp]
    lib.upkie_sim_step_pendulum_agent_rollout.restype = C.c_int
    lib.upkie_sim_step_pendulum_agent_rollout.argtypes = [vp, vp, vp, vp, C.c_int32, vp]
    lib.upkie_sim_step_base_velocity.restype = C.c_int
    lib.upkie_sim_step_base_velocity.argtypes = [vp] * 11
    lib.upkie_sim_step_base_velocity_mpc.restype = C.c_int
    lib.upkie_sim_step_base_velocity_mpc.argtypes = [vp] * 13
    lib.upkie_sim_observe.restype = C.c_int
    lib.upkie_sim_observe.argtypes = [
        vp,
        vp,
        C.POINTER(abi.UpkieSpineObservation),
        C.c_int,
        vp,
    ]
    lib.upkie_sim_autoreset_done.restype = C.c_int
    lib.upkie_sim_autoreset_done.argtypes = [vp, C.c_int, vp, vp, vp, vp]
    lib.upkie_sim_contact_points.restype = C.c_int
    lib.upkie_sim_contact_points.argtypes = [vp, vp, vp, vp]
    lib.upkie_mpc_create.restype = C.c_int
    lib.upkie_mpc_create.argtypes = [C.POINTER(abi.UpkieMpcConfig), C.POINTER(vp)]
    lib.upkie_mpc_destroy.restype = C.c_int
    lib.upkie_mpc_destroy.argtypes = [vp]
    lib.upkie_mpc_last_error.restype = C.c_char_p
    lib.upkie_mpc_last_error.argtypes = [vp]
    lib.upkie_mpc_workspace_bytes.restype = C.c_int64
    lib.upkie_mpc_workspace_bytes.argtypes = [vp]
    lib.upkie_mpc_reset.restype = C.c_int
    lib.upkie_mpc_reset.argtypes = [vp, vp, vp, vp, vp]
    lib.upkie_mpc_step.restype = C.c_int
    lib.upkie_mpc_step.argtypes = [vp, vp, vp, vp, vp, C.c_double, vp, vp, vp]
    lib.upkie_mpc_step_env.restype = C.c_int
    lib.upkie_mpc_step_env.argtypes = [vp, vp, vp, vp, vp, vp, C.c_double, vp, vp]
    lib.upkie_sim_attach_observers.restype = C.c_int
    lib.upkie_sim_attach_observers.argtypes = [vp, C.POINTER(abi.UpkieObserverConfig), vp]
    lib.upkie_observers_create.restype = C.c_int
    lib.upkie_observers_create.argtypes = [C.POINTER(abi.UpkieObserverConfig), C.POINTER(vp)]
    lib.upkie_observers_destroy.restype = C.c_int
    lib.upkie_observers_destroy.argtypes = [vp]
    lib.upkie_observers_last_error.restype = C.c_char_p
    lib.upkie_observers_last_error.argtypes = [vp]
    lib.upkie_observers_state_bytes.restype = C.c_int64
    lib.upkie_observers_state_bytes.argtypes = [vp]
    lib.upkie_observers_reset.restype = C.c_int
    lib.upkie_observers_reset.argtypes = [vp, vp, vp, vp]
    lib.upkie_observers_step.restype = C.c_int
    lib.upkie_observers_step.argtypes = [
        vp,
        vp,
        C.POINTER(abi.UpkieObserverInput),
        C.POINTER(abi.UpkieObserverOutput),
        vp,
    ]
    if hasattr(lib, "upkie_linear_policy"):  # (round 4; older builds loaded for A/B runs lack it)
        lib.upkie_linear_policy.restype = C.c_int
        lib.upkie_linear_policy.argtypes = [C.c_int32, C.c_int32, C.c_int32, vp, vp, vp, C.c_double, vp, vp]
    lib.upkie_rollout_gae.restype = C.c_int
    lib.upkie_rollout_gae.argtypes = [C.c_int32, C.c_int32, vp, vp, vp, vp, vp, C.c_double, C.c_double, vp, vp, vp]
    _lib = lib
    return lib


def check(status: int, handle=None, what: str = "sim") -> None:
    """Raise `UpkieHipError` on a negative status."""
    if status >= 0:
        return
    lib = load()
    if what == "mpc":
        msg = lib.upkie_mpc_last_error(handle)
    elif what == "observers":
        msg = lib.upkie_observers_last_error(handle)
    else:
        msg = lib.upkie_sim_last_error(handle)
    raise UpkieHipError(status, msg.decode() if msg else "")
