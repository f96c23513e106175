"""ctypes front-end of the CPU fp64 oracle (TEST INFRASTRUCTURE ONLY).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline
leg may import this module. The product package ``upkie_amd`` never does.
"""

import ctypes as C
import os
import subprocess

import numpy as np

from upkie_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libupkie_oracle.so")

_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")


class ServoCommand(C.Structure):
    _fields_ = [
        ("position", C.c_double),
        ("velocity", C.c_double),
        ("feedforward_torque", C.c_double),
        ("kp_scale", C.c_double),
        ("kd_scale", C.c_double),
        ("maximum_torque", C.c_double),
    ]


class Randomization(C.Structure):
    _fields_ = [
        ("body_inertials", C.c_void_p),
        ("ext_force", C.c_void_p),
        ("ext_point", C.c_double * 3),
        ("ext_slots", C.c_void_p),
        ("observer_config", C.c_void_p),
        ("observer_state", C.c_void_p),
        ("bullet_manifold", C.c_void_p),
    ]


BULLET_MANIFOLD_WORDS = 64


class SpineObservation(C.Structure):
    _fields_ = [(name, C.c_void_p) for name, _ in abi.UpkieSpineObservation._fields_]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (no-op when the .so is up to date)."""
    sources = [
        os.path.join(_HERE, name)
        for name in ("upkie_oracle.c", "upkie_oracle_mpc.c", "upkie_oracle_observers.c", "upkie_oracle.h")
    ] + [os.path.join(_HERE, "..", "include", "upkie_hip.h")]
    stale = force or not os.path.exists(_LIB_PATH)
    if not stale:
        mtime = os.path.getmtime(_LIB_PATH)
        stale = any(
            os.path.exists(s) and os.path.getmtime(s) > mtime for s in sources
        )
    if stale:
        subprocess.run(["make", "-C", _HERE, "-B"], check=True, capture_output=True)
    return _LIB_PATH


_REF_PATH = os.path.join(_HERE, "_ref", "libupkie_ref.so")
REFERENCE_ROOT = "/root/reference"


def build_ref(force: bool = False):
    """Compile oracle/_ref from the reference's own sources when they are
    present (build container); returns the path of the library or None."""
    if os.path.isdir(REFERENCE_ROOT) and (force or not os.path.exists(_REF_PATH)):
        subprocess.run(["make", "-C", _HERE, "ref", "-B"], check=True, capture_output=True)
    return _REF_PATH if os.path.exists(_REF_PATH) else None


_ref = None


def ref_lib():
    """The compiled reference pieces (oracle/_ref), or None when they were
    never built."""
    global _ref
    if _ref is None:
        path = build_ref()
        if path is None:
            return None
        _ref = C.CDLL(path)
        _ref.ref_low_pass_filter.restype = C.c_double
        _ref.ref_low_pass_filter.argtypes = [C.c_double, C.c_double, C.c_double, C.c_double, C.POINTER(C.c_int)]
    return _ref


_NATIVE_PATH = os.path.join(_HERE, "_build", "libupkie_oracle_native.so")
_lib = None


def use_native_build() -> bool:
    """bench.py's cpu_baseline: (re)build the oracle with -O3 -march=native ON
    THIS HOST and make it the library this process loads. Must be called
    before the first use of the oracle. The tuned library never travels: a
    sidecar remembers the CPU model it was built for."""
    global _lib
    if _lib is not None:
        return False
    cpu = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            cpu = next((line.split(":", 1)[1].strip() for line in f if line.startswith("model name")), cpu)
    except OSError:
        pass
    stamp = _NATIVE_PATH + ".host"
    built_for = open(stamp).read() if os.path.exists(stamp) else None
    try:
        sources = [os.path.join(_HERE, n) for n in ("upkie_oracle.c", "upkie_oracle_mpc.c", "upkie_oracle_observers.c", "upkie_oracle.h")] + [
            os.path.join(_HERE, "..", "include", "upkie_hip.h")]
        stale = not os.path.exists(_NATIVE_PATH) or any(os.path.getmtime(src) > os.path.getmtime(_NATIVE_PATH) for src in sources if os.path.exists(src))
        if built_for != cpu or stale:
            subprocess.run(["make", "-C", _HERE, "native", "-B"], check=True, capture_output=True)
            with open(stamp, "w") as f:
                f.write(cpu)
        _load(_NATIVE_PATH)
        return True
    except (subprocess.CalledProcessError, OSError):
        _lib = None
        return False


def usable_cores() -> int:
    """Host cores this process may actually run on: the scheduler affinity
    mask capped by the cgroup CPU quota (os.cpu_count() reports the machine's
    hardware threads, which a container is rarely given in full: 256 OpenMP
    threads on a 16-core quota measure oversubscription, not the CPU -- and
    make every oracle window of the test suite three times slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path, parse in (
        ("/sys/fs/cgroup/cpu.max", lambda t: None if t.split()[0] == "max" else int(t.split()[0]) / int(t.split()[1])),
        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: None if int(t) <= 0 else int(t) / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())),
    ):
        try:
            with open(path) as f:
                quota = parse(f.read().strip())
            if quota:
                n = max(1, min(n, int(quota)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def set_threads(threads: int = 0) -> int:
    """Size of the OpenMP team of the oracle's parallel loops (0: the usable
    cores, unless OMP_NUM_THREADS says otherwise). Returns what was set."""
    if threads <= 0:
        env = os.environ.get("OMP_NUM_THREADS", "")
        threads = int(env) if env.isdigit() and int(env) > 0 else usable_cores()
    try:
        C.CDLL("libgomp.so.1").omp_set_num_threads(int(threads))
    except OSError:
        return 0
    return int(threads)


def lib():
    global _lib
    if _lib is None:
        override = os.environ.get("UPKIE_ORACLE_LIBRARY")  # (tests/test_sanitizers.py: the same sources built with -fsanitize)
        if override:
            _load(override)
            return _lib
        if not os.path.exists(_LIB_PATH):
            build()
        _load(_LIB_PATH)
    return _lib


def _load(path):
    global _lib
    _lib = C.CDLL(path)
    _lib.oracle_joint_torque.restype = C.c_double
    _lib.oracle_joint_torque.argtypes = [
        C.c_double,
        C.c_double,
        C.POINTER(ServoCommand),
        C.c_double,
        C.c_double,
        C.c_double,
    ]
    _lib.oracle_total_mass.restype = C.c_double
    _lib.oracle_energy.restype = C.c_double
    _lib.oracle_substep.restype = C.c_int
    _lib.oracle_substep_ext.restype = C.c_int
    _lib.oracle_substep_bullet_like.restype = C.c_int
    _lib.oracle_mpc_solve_exact.restype = C.c_int
    _lib.oracle_observers_check.restype = C.c_int
    _lib.oracle_pitch_frame_in_parent.restype = C.c_double
    _lib.oracle_rollout_pendulum_agent.restype = C.c_int64
    set_threads()
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def philox(counter, key):
    ctr = (C.c_uint32 * 4)(*counter)
    k = (C.c_uint32 * 2)(*key)
    out = (C.c_uint32 * 4)()
    lib().oracle_philox4x32_10(ctr, k, out)
    return list(out)


def joint_torque(q, qd, cmd: dict, kp=20.0, kd=1.0, friction=0.0) -> float:
    c = ServoCommand(
        cmd["position"],
        cmd["velocity"],
        cmd.get("feedforward_torque", 0.0),
        cmd.get("kp_scale", 1.0),
        cmd.get("kd_scale", 1.0),
        cmd["maximum_torque"],
    )
    return lib().oracle_joint_torque(q, qd, C.byref(c), kp, kd, friction)


class Oracle:
    """Batched fp64 restatement, SoA state ``[STATE_WORDS, B]`` float64."""

    def __init__(self, model: abi.UpkieModel, config: abi.UpkieSimConfig):
        self.model = model
        self.config = config
        self.B = config.num_envs
        self.state = np.zeros((abi.STATE_WORDS, self.B), dtype=np.float64)
        self.body_inertials = None  # [70, B] per-env inertial records of the bodies (randomize_inertias)
        self.link_scale = None
        self.ext_force = None  # [3, B] (legacy: trunk, world frame) or [count, 3, B] with ext_slots
        self.ext_point = np.zeros(3)
        self.ext_slots = None  # abi.UpkieExternalForces
        self.observer_config = None  # abi.UpkieObserverConfig: spine observers inside the step ...
        self.observer_state = None  # ... and their memory [OBSERVER_STATE_WORDS, B]
        self.bullet_manifold = None  # `use_bullet_like_contacts`: [BULLET_MANIFOLD_WORDS, B]
        self._lib = lib()

    # -- randomisation -----------------------------------------------------
    def _rnd(self):
        if self.body_inertials is None and self.ext_force is None and self.observer_config is None and self.bullet_manifold is None:
            return None
        r = Randomization()
        r.body_inertials = _ptr(self.body_inertials)
        r.ext_force = _ptr(self.ext_force)
        r.ext_point[:] = list(self.ext_point)
        r.ext_slots = C.cast(C.pointer(self.ext_slots), C.c_void_p) if (self.ext_slots is not None and self.ext_force is not None) else None
        if self.observer_config is not None:
            r.observer_config = C.cast(C.pointer(self.observer_config), C.c_void_p)
            r.observer_state = _ptr(self.observer_state)
        r.bullet_manifold = _ptr(self.bullet_manifold)
        self._rnd_keepalive = r
        return C.byref(r)

    def use_bullet_like_contacts(self, on: bool = True) -> None:
        """Contacts by the Bullet-like specification of upkie_oracle.c (what
        Bullet 3.25's multibody solver is published to do: persistent 4-point
        manifolds, 50 fixed warm-started sweeps, cone friction, no friction
        CFM) instead of the product's: the yardstick of
        tools/bullet_like_deviation.py. Call before `reset()`."""
        self.bullet_manifold = np.zeros((BULLET_MANIFOLD_WORDS, self.B)) if on else None

    def attach_observers(self, config: abi.UpkieObserverConfig):
        """Spine observers inside the step, one cycle per substep."""
        h = self.config.dt / self.config.nb_substeps
        check = abi.UpkieObserverConfig.from_buffer_copy(config)
        check.dt = h
        if self._lib.oracle_observers_check(C.byref(check)) != 0:
            raise ValueError("observer filters need cutoff period > 2 dt (FilterError in the reference)")
        self.observer_config = config
        self.observer_state = np.zeros((abi.OBSERVER_STATE_WORDS, self.B))

    def sample_body_inertials(self, variation: float):
        """randomize_inertias (pybullet_backend.py:571-601): per-link factors
        ``[MAX_LINKS, B]`` and the fused per-env body records ``[70, B]``."""
        records = np.zeros((abi.NB * abi.INERTIAL_WORDS, self.B))
        link_scale = np.zeros((abi.MAX_LINKS, self.B))
        self._lib.oracle_sample_body_inertials(
            C.byref(self.model), C.byref(self.config), C.c_double(variation), _ptr(records), _ptr(link_scale)
        )
        self.link_scale = link_scale
        return records

    def sample_pushes(self, push_index: int, max_norm: float):
        """Push number `push_index` of every env: world-frame force ``[3, B]``,
        norm ~ U(0, max_norm), random horizontal direction."""
        force = np.zeros((3, self.B))
        self._lib.oracle_sample_pushes(C.byref(self.config), C.c_uint32(int(push_index) & 0xFFFFFFFF), C.c_double(max_norm), _ptr(force))
        return force

    # -- env API -----------------------------------------------------------
    def reset(self, mask=None):
        obs6 = np.zeros((self.B, 6))
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        self._lib.oracle_reset(
            C.byref(self.model),
            C.byref(self.config),
            _ptr(self.state),
            _ptr(m),
            self._rnd(),
            _ptr(obs6),
        )
        return obs6

    def _step(self, fn, act, obs_dim):
        obs = np.zeros((self.B, obs_dim))
        rew = np.zeros(self.B)
        term = np.zeros(self.B, dtype=np.uint8)
        trunc = np.zeros(self.B, dtype=np.uint8)
        act = np.ascontiguousarray(act, dtype=np.float64)
        fn(
            C.byref(self.model),
            C.byref(self.config),
            _ptr(self.state),
            _ptr(act),
            _ptr(obs),
            _ptr(rew),
            _ptr(term),
            _ptr(trunc),
            self._rnd(),
        )
        return obs, rew, term, trunc

    def step_pendulum(self, act):
        return self._step(self._lib.oracle_step_pendulum, act, 4)

    def step_gyropod(self, act):
        return self._step(self._lib.oracle_step_gyropod, act, 6)

    def step_servos(self, act):
        obs, rew, term, trunc = self._step(self._lib.oracle_step_servos, act, 30)
        return obs.reshape(self.B, 6, 5), rew, term, trunc

    def step_pendulum_agent(self, obs):
        obs = np.ascontiguousarray(obs, dtype=np.float64).copy()
        rew = np.zeros(self.B)
        term = np.zeros(self.B, dtype=np.uint8)
        trunc = np.zeros(self.B, dtype=np.uint8)
        self._lib.oracle_step_pendulum_agent(
            C.byref(self.model),
            C.byref(self.config),
            _ptr(self.state),
            _ptr(obs),
            _ptr(rew),
            _ptr(term),
            _ptr(trunc),
            self._rnd(),
        )
        return obs, rew, term, trunc

    def rollout_pendulum_agent(self, obs, steps: int):
        """`steps` consecutive `step_pendulum_agent` in one parallel region
        (bench.py's cpu_baseline); returns (obs, terminations seen)."""
        obs = np.ascontiguousarray(obs, dtype=np.float64).copy()
        falls = self._lib.oracle_rollout_pendulum_agent(
            C.byref(self.model), C.byref(self.config), _ptr(self.state), _ptr(obs), C.c_int32(steps), self._rnd()
        )
        return obs, int(falls)

    def observe(self, update_imu: bool = True) -> dict:
        B = self.B
        out = {
            "pitch": np.zeros(B),
            "angular_velocity": np.zeros((B, 3)),
            "linear_velocity": np.zeros((B, 3)),
            "rotation_base_to_world": np.zeros((B, 9)),
            "floor_contact": np.zeros(B, dtype=np.uint8),
            "imu_orientation": np.zeros((B, 4)),
            "imu_angular_velocity": np.zeros((B, 3)),
            "imu_linear_acceleration": np.zeros((B, 3)),
            "imu_raw_linear_acceleration": np.zeros((B, 3)),
            "servo": np.zeros((B, 6, 5)),
            "wheel_odometry": np.zeros((B, 2)),
        }
        so = SpineObservation()
        for name in out:
            setattr(so, name, _ptr(out[name]))
        self._lib.oracle_observe(
            C.byref(self.model),
            C.byref(self.config),
            _ptr(self.state),
            C.byref(so),
            C.c_int(1 if update_imu else 0),
        )
        return out

    def contact_points(self) -> np.ndarray:
        """[B, 2, 8]: per tire [exists, position in world, force in world, 0]
        (PyBulletBackend.get_contact_points, pybullet_backend.py:660-716)."""
        out = np.zeros((self.B, 2, 8))
        self._lib.oracle_contact_points(C.byref(self.model), C.byref(self.config), _ptr(self.state), self._rnd(), _ptr(out))
        return out

    # -- low level ---------------------------------------------------------
    def substep(self, env: int, tau, h: float) -> int:
        s = np.ascontiguousarray(self.state[:, env])
        tau = np.ascontiguousarray(tau, dtype=np.float64)
        scale = (
            np.ascontiguousarray(self.body_inertials[:, env])
            if self.body_inertials is not None
            else None
        )
        if self.ext_force is not None and self.ext_slots is not None:
            force = np.ascontiguousarray(self.ext_force[:, :, env])
            contact = self._lib.oracle_substep_ext(
                C.byref(self.model), _ptr(s), _ptr(tau), C.c_double(h), _ptr(scale), _ptr(force), C.byref(self.ext_slots)
            )
        else:
            force = np.ascontiguousarray(self.ext_force[:, env]) if self.ext_force is not None else None
            point = np.ascontiguousarray(self.ext_point, dtype=np.float64)
            contact = self._lib.oracle_substep(
                C.byref(self.model), _ptr(s), _ptr(tau), C.c_double(h), _ptr(scale), _ptr(force), _ptr(point)
            )
        self.state[:, env] = s
        return contact


def mass_matrix_and_bias(model, pos, quat, linvel, angvel, q, qd):
    M = np.zeros((12, 12))
    h = np.zeros(12)
    args = [np.ascontiguousarray(a, dtype=np.float64) for a in (pos, quat, linvel, angvel, q, qd)]
    lib().oracle_mass_matrix_and_bias(C.byref(model), *[_ptr(a) for a in args], _ptr(M), _ptr(h))
    return M, h


def total_mass(model) -> float:
    return lib().oracle_total_mass(C.byref(model))


def center_of_mass(model, q=None):
    q = np.zeros(6) if q is None else np.ascontiguousarray(q, dtype=np.float64)
    out = np.zeros(3)
    lib().oracle_center_of_mass(C.byref(model), _ptr(q), _ptr(out))
    return out


def energy(model, pos, quat, linvel, angvel, q, qd) -> float:
    args = [np.ascontiguousarray(a, dtype=np.float64) for a in (pos, quat, linvel, angvel, q, qd)]
    return lib().oracle_energy(C.byref(model), *[_ptr(a) for a in args])


class ObserverOracle:
    """fp64 observer pipeline for B envs (upkie_oracle_observers.c)."""

    def __init__(self, config: abi.UpkieObserverConfig):
        self.config = config
        self.B = config.num_envs
        if lib().oracle_observers_check(C.byref(config)) != 0:
            raise ValueError("observer filters need cutoff period > 2 dt (FilterError in the reference)")
        self.state = np.zeros((abi.OBSERVER_STATE_WORDS, self.B))

    def reset(self, mask=None):
        if mask is None:
            self.state[:] = 0.0
        else:
            self.state[:, np.asarray(mask, dtype=bool)] = 0.0

    def step(self, servo, imu_orientation=None, imu_angular_velocity=None, cross_button=None) -> dict:
        B = self.B
        servo = np.ascontiguousarray(servo, dtype=np.float64).reshape(B, 6, 5)
        q = None if imu_orientation is None else np.ascontiguousarray(imu_orientation, dtype=np.float64).reshape(B, 4)
        w = None if imu_angular_velocity is None else np.ascontiguousarray(imu_angular_velocity, dtype=np.float64).reshape(B, 3)
        cb = None if cross_button is None else np.ascontiguousarray(cross_button, dtype=np.uint8).reshape(B)
        out = dict(
            base_pitch=np.zeros(B),
            base_angular_velocity=np.zeros((B, 3)),
            rotation_base_to_world=np.zeros((B, 9)),
            floor_contact=np.zeros(B, dtype=np.uint8),
            upper_leg_torque=np.zeros(B),
            wheel_contact=np.zeros((B, 2, 4)),
            wheel_odometry=np.zeros((B, 2)),
        )
        lib().oracle_observers_step(
            C.byref(self.config), _ptr(self.state), _ptr(servo), _ptr(q), _ptr(w), _ptr(cb),
            _ptr(out["base_pitch"]), _ptr(out["base_angular_velocity"]), _ptr(out["rotation_base_to_world"]),
            _ptr(out["floor_contact"]), _ptr(out["upper_leg_torque"]), _ptr(out["wheel_contact"]), _ptr(out["wheel_odometry"]),
        )
        if q is None:
            for k in ("base_pitch", "base_angular_velocity", "rotation_base_to_world"):
                out.pop(k)
        return out


def pitch_frame_in_parent(R) -> float:
    R = np.ascontiguousarray(R, dtype=np.float64).reshape(9)
    return float(lib().oracle_pitch_frame_in_parent(_ptr(R)))


def base_orientation_from_imu(q_wxyz, base_to_imu, ars_to_world):
    q = np.ascontiguousarray(q_wxyz, dtype=np.float64).reshape(4)
    a = np.ascontiguousarray(base_to_imu, dtype=np.float64).reshape(9)
    b = np.ascontiguousarray(ars_to_world, dtype=np.float64).reshape(9)
    R = np.zeros(9)
    lib().oracle_base_orientation_from_imu(_ptr(q), _ptr(a), _ptr(b), _ptr(R))
    return R.reshape(3, 3)
