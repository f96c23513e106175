"""The C-ABI library loads, exports every symbol include/upkie_hip.h declares
and the ctypes mirrors agree with the header. No compute calls (no GPU)."""

import ctypes as C
import os
import re
import subprocess
import tempfile

import pytest

from upkie_amd import abi, lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "upkie_hip.h")


@pytest.fixture(scope="module")
def library():
    lib.build()
    return lib.load()


def test_exports_every_declared_symbol(library):
    with open(HEADER) as f:
        text = f.read()
    declared = set(re.findall(r"\b(upkie_[a-z_]+)\s*\(", text))
    assert declared == set(lib.EXPORTED_SYMBOLS)
    for name in declared:
        assert getattr(library, name) is not None


def test_struct_layouts_match_header():
    probe = r"""
    #include <stdio.h>
    #include <stddef.h>
    #include "upkie_hip.h"
    int main(void) {
      printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(UpkieModel), sizeof(UpkieSimConfig), sizeof(UpkieMpcConfig), sizeof(UpkieSpineObservation), sizeof(UpkieObserverConfig), sizeof(UpkieObserverInput), sizeof(UpkieObserverOutput));
      printf("%zu %zu %zu %zu %zu\n", offsetof(UpkieModel, joint_pos), offsetof(UpkieModel, wheel_radius), offsetof(UpkieModel, gravity), offsetof(UpkieModel, pgs_iterations), offsetof(UpkieModel, enforce_joint_limits));
      printf("%zu %zu %zu %zu %zu\n", offsetof(UpkieSimConfig, dt), offsetof(UpkieSimConfig, fall_pitch), offsetof(UpkieSimConfig, init_pos), offsetof(UpkieSimConfig, seed), offsetof(UpkieSimConfig, agent_clip));
      printf("%zu %zu\n", offsetof(UpkieMpcConfig, sampling_period), offsetof(UpkieMpcConfig, admm_rho));
      printf("%d %d %d %d\n", UPKIE_STATE_WORDS, UPKIE_S_TORQUE, UPKIE_S_DONE, UPKIE_S_CONTACT);
      printf("%zu %zu %zu %d\n", sizeof(UpkieServoPolicy), offsetof(UpkieServoPolicy, velocity_feedback_clip), offsetof(UpkieServoPolicy, fall_pitch), UPKIE_CENSUS_WORDS);
      printf("%zu %zu %zu %d %d %d\n", offsetof(UpkieObserverConfig, dt), offsetof(UpkieObserverConfig, signed_radius), offsetof(UpkieObserverConfig, rotation_ars_to_world), UPKIE_OBSERVER_STATE_WORDS, UPKIE_O_UPPER_LEG_TORQUE, UPKIE_O_ODOMETRY_VELOCITY);
      return 0;
    }
    """
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "probe.c")
        exe = os.path.join(tmp, "probe")
        with open(src, "w") as f:
            f.write(probe)
        subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), src, "-o", exe], check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split("\n")
    sizes = [int(x) for x in out[0].split()]
    assert sizes == [
        C.sizeof(abi.UpkieModel),
        C.sizeof(abi.UpkieSimConfig),
        C.sizeof(abi.UpkieMpcConfig),
        C.sizeof(abi.UpkieSpineObservation),
        C.sizeof(abi.UpkieObserverConfig),
        C.sizeof(abi.UpkieObserverInput),
        C.sizeof(abi.UpkieObserverOutput),
    ]
    m = abi.UpkieModel
    assert [int(x) for x in out[1].split()] == [m.joint_pos.offset, m.wheel_radius.offset, m.gravity.offset, m.pgs_iterations.offset, m.enforce_joint_limits.offset]
    c = abi.UpkieSimConfig
    assert [int(x) for x in out[2].split()] == [c.dt.offset, c.fall_pitch.offset, c.init_pos.offset, c.seed.offset, c.agent_clip.offset]
    p = abi.UpkieMpcConfig
    assert [int(x) for x in out[3].split()] == [p.sampling_period.offset, p.admm_rho.offset]
    assert [int(x) for x in out[4].split()] == [abi.STATE_WORDS, abi.S_TORQUE, abi.S_DONE, abi.S_CONTACT]
    sp = abi.UpkieServoPolicy
    assert [int(x) for x in out[5].split()] == [C.sizeof(sp), sp.velocity_feedback_clip.offset, sp.fall_pitch.offset, abi.CENSUS_WORDS]
    o = abi.UpkieObserverConfig
    assert [int(x) for x in out[6].split()] == [
        o.dt.offset,
        o.signed_radius.offset,
        o.rotation_ars_to_world.offset,
        abi.OBSERVER_STATE_WORDS,
        abi.O_UPPER_LEG_TORQUE,
        abi.O_ODOMETRY_VELOCITY,
    ]


def test_no_silent_cpu_fallback(library):
    """Without a GPU the product path must fail loudly, not compute elsewhere."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the failure path cannot be exercised")
    assert library.upkie_hip_device_count() == 0
    from upkie_amd.model.default_model import default_model

    handle = C.c_void_p()
    cfg, model = abi.default_sim_config(4), default_model()
    status = library.upkie_sim_create(C.byref(cfg), C.byref(model), C.byref(handle))
    assert status == abi.ERR_NO_DEVICE and not handle
    assert b"no HIP device" in library.upkie_sim_last_error(None)
    from upkie_amd.exceptions import UpkieRuntimeError
    from upkie_amd.sim import BatchedSim

    with pytest.raises(UpkieRuntimeError):
        BatchedSim(cfg, model)


def test_product_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py may touch oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "upkie_amd")):
        for name in files:
            if name.endswith((".py", ".hip", ".hpp", ".h")):
                with open(os.path.join(dirpath, name)) as f:
                    text = f.read()
                assert "import oracle" not in text and "from oracle" not in text, name
                assert "upkie_oracle" not in text, name


def test_the_build_watches_every_source_of_the_library():
    """`lib.build()` rebuilds when a source is newer than the library: every file under csrc/ (and the header) must be
    on its list, or an edited kernel header ships with a stale `.so`."""
    import os

    from upkie_amd import lib

    csrc = os.path.join(os.path.dirname(os.path.abspath(lib.__file__)), "csrc")
    watched = {os.path.realpath(s) for s in lib.SOURCES}
    for name in os.listdir(csrc):
        if name.endswith((".hpp", ".hip", ".h")):
            assert os.path.realpath(os.path.join(csrc, name)) in watched, name
    assert any(s.endswith("upkie_hip.h") for s in watched)


def test_observer_filter_error_is_reported_without_a_gpu(library):
    """low_pass_filter throws FilterError when cutoff <= 2 dt
    (upkie/cpp/utils/low_pass_filter.h:22-30): the C-ABI reports it at create,
    before any device work."""
    handle = C.c_void_p()
    cfg = abi.default_observer_config(8, 1.0 / 200.0)  # the 0.01 s upper-leg filter, FloorContact.cpp:87
    assert library.upkie_observers_create(C.byref(cfg), C.byref(handle)) == abi.ERR_INVALID_ARGUMENT
    assert not handle
    assert b"low_pass_filter" in library.upkie_observers_last_error(None)
    cfg = abi.default_observer_config(8, 1e-3)
    cfg.wheel_cutoff_period = 0.002
    assert library.upkie_observers_create(C.byref(cfg), C.byref(handle)) == abi.ERR_INVALID_ARGUMENT
    cfg = abi.default_observer_config(8, float("nan"))
    assert library.upkie_observers_create(C.byref(cfg), C.byref(handle)) == abi.ERR_INVALID_ARGUMENT
    import torch

    if not torch.cuda.is_available():
        cfg = abi.default_observer_config(8, 1e-3)
        assert library.upkie_observers_create(C.byref(cfg), C.byref(handle)) == abi.ERR_NO_DEVICE
        from upkie_amd.exceptions import UpkieRuntimeError
        from upkie_amd.observers import BatchedObservers

        with pytest.raises(UpkieRuntimeError):
            BatchedObservers(cfg)


def test_observer_config_from_spine_config():
    from upkie_amd.observers import observer_config_from_spine_config

    cfg = observer_config_from_spine_config(3, 1e-3)
    assert cfg.num_envs == 3 and cfg.wheel_cutoff_period == 0.2 and cfg.upper_leg_torque_threshold == 10.0
    assert list(cfg.signed_radius) == [0.05, -0.05]
    assert [cfg.rotation_base_to_imu[i] for i in (0, 4, 8)] == [-1.0, 1.0, -1.0]
    cfg = observer_config_from_spine_config(
        3,
        1e-3,
        {
            "floor_contact": {"upper_leg_torque_threshold": 7.0},
            "wheel_contact": {
                "cutoff_period": 0.1,
                "liftoff_inertia": 0.002,
                "min_touchdown_acceleration": 1.0,
                "min_touchdown_torque": 0.02,
                "touchdown_inertia": 0.005,
            },
            "wheel_odometry": {"signed_radius": {"left_wheel": 0.06, "right_wheel": -0.06}},
            "base_orientation": {"rotation_base_to_imu": [[0, -1, 0], [1, 0, 0], [0, 0, 1]]},
        },
    )
    assert cfg.upper_leg_torque_threshold == 7.0 and cfg.wheel_cutoff_period == 0.1 and cfg.touchdown_inertia == 0.005
    assert list(cfg.signed_radius) == [0.06, -0.06]
    assert list(cfg.rotation_base_to_imu) == [0, -1, 0, 1, 0, 0, 0, 0, 1]


def test_library_reports_the_struct_sizes_the_bindings_were_written_for(library):
    """`upkie_hip_struct_bytes` (what lib.load() checks before any call): every public struct, as the built library
    sees it, is as long as its ctypes mirror; an unknown id answers -1; a mismatching library is refused."""
    assert set(abi.STRUCT_IDS) == set(range(len(abi.STRUCT_IDS)))
    for which, cls in abi.STRUCT_IDS.items():
        assert library.upkie_hip_struct_bytes(which) == C.sizeof(cls), cls.__name__
    assert library.upkie_hip_struct_bytes(len(abi.STRUCT_IDS)) == -1

    class Shorter:  # a library built from an older header: UpkieMpcConfig without its last field
        def upkie_hip_struct_bytes(self, which):
            return C.sizeof(abi.STRUCT_IDS[which]) - (8 if abi.STRUCT_IDS[which] is abi.UpkieMpcConfig else 0)

    fake = Shorter()
    fake.upkie_hip_struct_bytes = type("F", (), {"__call__": lambda self, which: Shorter.upkie_hip_struct_bytes(fake, which), "restype": None, "argtypes": None})()
    with pytest.raises(Exception, match="UpkieMpcConfig"):
        lib._check_struct_sizes(fake)


def test_header_documents_streams_and_graph_captures():
    """The limits of the handle's settings blocks are part of the interface: stated in the header,
    the constant mirrored in abi.py."""
    with open(HEADER) as f:
        text = f.read()
    assert "Streams and hipGraphs" in text and "UPKIE_MAX_GRAPH_CAPTURES" in text
    assert int(re.search(r"#define UPKIE_MAX_GRAPH_CAPTURES (\d+)", text).group(1)) == abi.MAX_GRAPH_CAPTURES
