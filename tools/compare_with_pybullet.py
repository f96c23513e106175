"""Side-by-side run of the TRUE reference (the upkie package on PyBullet +
upkie_description, MPCBalancer on qpmpc + proxsuite) and this repository's HIP
path, for a machine where the reference's dependencies are installed (they are
not in the build container: SURVEY.md section 8c). One command per pin that
DESIGN.md section 4 lists as missing, each writing a JSON report:

    python tools/compare_with_pybullet.py --workload pendulum --steps 400 --json a8_pendulum.json     # a8 (C2 inputs)
    python tools/compare_with_pybullet.py --workload servos   --steps 800 --json a8_servos.json       # a8 (C5 inputs)
    python tools/compare_with_pybullet.py --workload mpc      --steps 600 --json a18_mpc.json         # a18

  pendulum  Upkie-PyBullet-Pendulum vs Upkie-HIP-Pendulum under the README agent: per-step observation differences.
  servos    Upkie-PyBullet-Servos vs Upkie-HIP-Servos under examples/pybullet/torque_balancing.py's law (the C5
            workload of SURVEY 8d: wheel friction 0.1, a torso push of --push-force N held 20 steps every 400 steps):
            per-step JOINT TORQUES, positions and velocities of all six servos -- what BASELINE's north_star asks to
            match "within a stated float tolerance".
  mpc       upkie.controllers.MPCBalancer (ProxQP, eps_abs 1e-3) vs upkie_amd.mpc.BatchedMpc on the SAME sequence of
            spine observations and targets (recorded from this repository's BaseVelocity env running its own balancer):
            commanded ground velocity and first input per step.

--contact-model bullet_like (the default) runs this repository's side under the Bullet-like contact specification
(`upkie_sim_set_contact_manifold`: persistent 4-point manifolds, 50 fixed sweeps, cone friction) -- the restatement of
what pybullet.stepSimulation() is published to do, and the first thing to hold against the real thing; `default` runs
the product's fast specification. --time adds the reference's env-steps/s on one core (BASELINE.md's B2).

--against doubles replaces the reference by this repository's fp64 oracle doubles (tests/fake_sim.py: test
infrastructure) so that the tool's own plumbing -- both loops, the push schedule, the report -- can be exercised where
the reference's dependencies are missing (tests/test_examples_gpu.py does so on the GPU box). It pins nothing.

Expect agreement of the wrapper arithmetic and, with upkie_description's URDF loaded on both sides, step-for-step
agreement of the dynamics at the tolerances DESIGN.md section 4 holds the kernels to against the oracle; every number
lands in the report either way.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)

PUSH_PERIOD, PUSH_HOLD = 400, 20
JOINTS = ("left_hip", "left_knee", "left_wheel", "right_hip", "right_knee", "right_wheel")


def quantiles(x):
    x = np.asarray(x, dtype=np.float64)
    return {f"q{q:g}": float(np.quantile(x, q)) for q in (0.5, 0.9, 0.99, 1.0)} if x.size else {}


def make_pair(args, kind):
    """(reference env, our env) of `kind` in ("Pendulum", "Servos", "BaseVelocity")."""
    import upkie_amd.envs as envs
    from upkie_amd.model.joint_properties import JointProperties
    from upkie_amd.model.model import Model

    props = {n: JointProperties(friction=0.1) for n in ("left_wheel", "right_wheel")} if kind == "Servos" else None
    extra = {} if kind != "BaseVelocity" else {"nb_timesteps": args.horizon}
    ours = envs.make(f"Upkie-HIP-{kind}", frequency=200.0, model=Model(), contact_model=args.contact_model,  # Model() picks up upkie_description's URDF
                     **({"joint_properties": props} if props else {}), **extra)
    if args.against == "doubles":
        from tests.fake_sim import OracleMpc, oracle_sim_factory

        more = {"mpc_factory": OracleMpc} if kind == "BaseVelocity" else {}
        ref = envs.make(f"Upkie-HIP-{kind}", frequency=200.0, model=Model(), contact_model=args.contact_model, sim_factory=oracle_sim_factory,
                        **({"joint_properties": props} if props else {}), **extra, **more)
        return ref, ours
    import gymnasium as gym
    import upkie.envs  # the reference package

    upkie.envs.register()
    kw = dict(frequency=200.0, gui=False, regulate_frequency=False, frequency_checks=False)
    if props:
        from upkie.model.joint_properties import JointProperties as RefProps  # noqa: N811

        kw["joint_properties"] = {n: RefProps(friction=0.1) for n in props}
    return gym.make(f"Upkie-PyBullet-{kind}", **kw, **extra), ours


def run_pendulum(args):
    ref, ours = make_pair(args, "Pendulum")
    gain = np.array([10.0, 1.0, 0.0, 0.1])
    obs_r, _ = ref.reset(seed=0)
    obs_o, _ = ours.reset(seed=0)
    err = []
    ended = None
    for step in range(args.steps):
        act = lambda o: np.clip(gain.dot(np.asarray(o, dtype=np.float64)), -0.99, 0.99).reshape((1,)).astype(np.float32)  # noqa: E731
        obs_r, _, term_r, trunc_r, _ = ref.step(act(obs_r))
        obs_o, _, term_o, trunc_o, _ = ours.step(act(obs_o))
        err.append(np.abs(np.asarray(obs_r, dtype=np.float64) - np.asarray(obs_o, dtype=np.float64)))
        if step % 50 == 0:
            print(f"step {step:4d}  reference {np.asarray(obs_r)}  hip {np.asarray(obs_o)}")
        if term_r or term_o:
            ended = {"step": step, "reference": bool(term_r), "hip": bool(term_o)}
            break
    err = np.array(err)
    report = {"steps_compared": len(err), "terminated": ended,
              "observation_error": {name: quantiles(err[:, i]) for i, name in enumerate(("pitch", "position", "pitch_rate", "velocity"))}}
    if args.time and args.against == "reference":
        obs_r, _ = ref.reset(seed=0)
        t0, n = time.perf_counter(), 2000
        for _ in range(n):
            obs_r, _, term, trunc, _ = ref.step(np.clip(gain.dot(obs_r), -0.99, 0.99).reshape((1,)).astype(np.float32))
            if term or trunc:
                obs_r, _ = ref.reset()
        report["reference_env_steps_per_s_one_core"] = n / (time.perf_counter() - t0)
    return report


def torque_balancing_action(env, pitch):
    """examples/pybullet/torque_balancing.py:15-37: legs held at zero, wheel torques +-10 x pitch, kd_scale 0."""
    action = env.get_neutral_action()
    for name in JOINTS:
        if "wheel" in name:
            action[name]["position"] = np.nan
            action[name]["velocity"] = 0.0
            action[name]["kd_scale"] = 0.0
            action[name]["feedforward_torque"] = (10.0 if name.startswith("left") else -10.0) * pitch
        else:
            action[name]["position"] = 0.0
    return action


def push(env, force):
    """A world-frame force on the torso (PyBulletBackend.set_external_forces, pybullet_backend.py:603-658), or None to end it."""
    target = getattr(env, "unwrapped", env)
    vector = np.zeros(3) if force is None else np.asarray(force, dtype=np.float64)
    for ours in (target, getattr(target, "_vec", None)):  # this repository's envs (the single-robot ones wrap a vector env of one)
        if ours is not None and hasattr(ours, "set_external_forces"):
            ours.set_external_forces({"torso": (vector, False)})
            return True
    backend = getattr(target, "backend", None)
    if backend is not None and hasattr(backend, "set_external_forces"):
        from upkie.utils.external_force import ExternalForce

        backend.set_external_forces({"torso": ExternalForce(vector, local=False)})
        return True
    return False


def run_servos(args):
    ref, ours = make_pair(args, "Servos")
    obs_r, info_r = ref.reset(seed=0)
    obs_o, info_o = ours.reset(seed=0)
    pitch = lambda info: float(info["spine_observation"]["base_orientation"]["pitch"])  # noqa: E731
    rng = np.random.default_rng(0)
    err = {k: [] for k in ("torque", "position", "velocity")}
    pitch_err, pushed = [], True
    for step in range(args.steps):
        phase = step % PUSH_PERIOD
        if phase == 0 and args.push_force > 0:
            angle = rng.uniform(0, 2 * np.pi)
            force = args.push_force * np.array([np.cos(angle), np.sin(angle), 0.0])
            pushed = push(ref, force) and push(ours, force) and pushed
        elif phase == PUSH_HOLD and args.push_force > 0:
            push(ref, None)
            push(ours, None)
        obs_r, _, _, _, info_r = ref.step(torque_balancing_action(ref, pitch(info_r)))
        obs_o, _, _, _, info_o = ours.step(torque_balancing_action(ours, pitch(info_o)))
        for key in err:
            err[key].append([abs(float(obs_r[j][key][0] if np.ndim(obs_r[j][key]) else obs_r[j][key]) -
                                 float(obs_o[j][key][0] if np.ndim(obs_o[j][key]) else obs_o[j][key])) for j in JOINTS])
        pitch_err.append(abs(pitch(info_r) - pitch(info_o)))
        if step % 100 == 0:
            print(f"step {step:4d}  pitch reference {pitch(info_r):+.5f}  hip {pitch(info_o):+.5f}  wheel torque error {err['torque'][-1][2]:.2e} N.m")
        if abs(pitch(info_r)) > 1.0 or abs(pitch(info_o)) > 1.0:
            break
    report = {"steps_compared": len(pitch_err), "pushes_applied_on_both_sides": bool(pushed and args.push_force > 0), "pitch_error": quantiles(pitch_err)}
    for key, rows in err.items():
        rows = np.array(rows)
        report[f"joint_{key}_error"] = {j: quantiles(rows[:, i]) for i, j in enumerate(JOINTS)}
        for window, sel in (("first_100_steps", slice(0, 100)), ("all_steps", slice(None))):
            report[f"joint_{key}_error_{window}_worst_joint"] = quantiles(rows[sel].max(axis=1))
    return report


def run_mpc(args):
    """The balancer alone, on identical inputs: our BaseVelocity env runs closed loop and records what its balancer saw and
    commanded; the reference's MPCBalancer (or the fp64 ADMM double) is stepped on the same observations and targets."""
    import torch

    ref_env, ours = make_pair(args, "BaseVelocity")  # (only `ours` is stepped; the reference side is the balancer below)
    if args.against == "doubles":
        balancer = None
        double = ref_env._vec.mpc_balancer
    else:
        from upkie.controllers import MPCBalancer

        balancer = MPCBalancer(nb_timesteps=args.horizon)
    rng = np.random.default_rng(0)
    ours.reset(seed=0)
    vec = ours._vec
    target = 0.0
    v_err, commanded_r = [], 0.0
    for step in range(args.steps):
        if step % PUSH_PERIOD == 0:
            target = float(rng.uniform(-0.5, 0.5))
        x0 = vec._x0.cpu().numpy()[0].astype(np.float64)  # ground position, pitch, ground velocity, pitch rate: what the balancer reads
        contact = bool(vec._contact.cpu().numpy()[0])
        if balancer is not None:
            spine = {"floor_contact": {"contact": contact}, "base_orientation": {"pitch": x0[1], "angular_velocity": np.array([0.0, x0[3], 0.0])},
                     "wheel_odometry": {"position": x0[0], "velocity": x0[2]}}
            commanded_r = float(balancer.step(target, spine, ours.dt))
        else:
            v, _ = double.step(torch.from_numpy(x0[None].astype(np.float32)), torch.tensor([target]), torch.tensor([contact], dtype=torch.uint8), ours.dt)
            commanded_r = float(v[0])
        ours.step(np.array([target, 0.0], dtype=np.float32))
        v_err.append(abs(commanded_r - float(vec.mpc_balancer.commanded_velocity.cpu().numpy()[0])))
        if step % 100 == 0:
            print(f"step {step:4d}  target {target:+.3f}  commanded reference {commanded_r:+.5f}  hip {float(vec.mpc_balancer.commanded_velocity[0]):+.5f}")
    return {"steps_compared": len(v_err), "horizon": args.horizon, "commanded_velocity_error_m_per_s": quantiles(v_err),
            "note": "ProxQP stops at eps_abs 1e-3: up to 4.7e-3 m/s per step lies inside the reference's own tolerance (DESIGN.md section 4)"}


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--workload", choices=("pendulum", "servos", "mpc"), default="pendulum")
    parser.add_argument("--steps", type=int, default=200)
    parser.add_argument("--time", action="store_true")
    parser.add_argument("--contact-model", choices=("bullet_like", "default"), default="bullet_like")
    parser.add_argument("--against", choices=("reference", "doubles"), default="reference")
    parser.add_argument("--push-force", type=float, default=10.0, help="servos: norm of the torso push [N] (0: none)")
    parser.add_argument("--horizon", type=int, default=50, help="mpc: nb_timesteps (the reference's default)")
    parser.add_argument("--json", default=None, help="write the report here")
    args = parser.parse_args()
    if args.against == "reference":
        try:
            import gymnasium  # noqa: F401
            import upkie.envs  # noqa: F401
        except ImportError as exc:
            print(f"reference not importable here ({exc}); nothing to compare (--against doubles exercises the tool itself)")
            return 0
    report = {"workload": args.workload, "against": args.against, "contact_model": args.contact_model}
    report.update({"pendulum": run_pendulum, "servos": run_servos, "mpc": run_mpc}[args.workload](args))
    text = json.dumps(report, indent=1)
    print(text)
    if args.json:
        with open(args.json, "w") as f:
            f.write(text + "\n")
    return 0


if __name__ == "__main__":
    sys.exit(main())
