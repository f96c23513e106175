"""Environment wrappers on the real HIP path (one MI355X)."""

import math

import numpy as np
import pytest
import torch

import upkie_amd.envs as envs
from upkie_amd import abi
from upkie_amd.utils.robot_state import RobotState
from upkie_amd.utils.robot_state_randomization import RobotStateRandomization

from .fake_sim import OracleMpc, oracle_sim_factory
from .helpers import randomized_config

pytestmark = pytest.mark.gpu


def test_readme_agent_loop_on_gpu_matches_cpu_double():
    """README.md:53-68 loop on "Upkie-HIP-Pendulum" vs the same env on the
    oracle-backed double: observations within the closed-loop tolerance."""
    gpu = envs.make("Upkie-HIP-Pendulum", frequency=200.0)
    cpu = envs.make("Upkie-HIP-Pendulum", frequency=200.0, sim_factory=oracle_sim_factory)
    og, _ = gpu.reset(seed=1)
    oc, _ = cpu.reset(seed=1)
    gain = np.array([10.0, 1.0, 0.0, 0.1])
    for _ in range(200):
        og, rg, tg, ug, ig = gpu.step(gain.dot(og).reshape((1,)))
        oc, rc, tc, uc, ic = cpu.step(gain.dot(oc).reshape((1,)))
        assert (rg, tg, ug) == (rc, tc, uc)
    assert abs(og[0] - oc[0]) < 1e-3 and abs(og[1] - oc[1]) < 1e-3
    sg, sc = ig["spine_observation"], ic["spine_observation"]
    assert sg["floor_contact"]["contact"] == sc["floor_contact"]["contact"]
    assert sg["base_orientation"]["pitch"] == pytest.approx(sc["base_orientation"]["pitch"], abs=1e-3)
    np.testing.assert_allclose(sg["imu"]["orientation"], sc["imu"]["orientation"], atol=1e-3)
    for name in abi.JOINT_NAMES:
        assert sg["servo"][name]["position"] == pytest.approx(sc["servo"][name]["position"], abs=2e-3)


def test_pybullet_id_alias_runs_on_hip():
    env = envs.make("Upkie-PyBullet-Pendulum", frequency=200.0, regulate_frequency=False)
    obs, info = env.reset()
    assert obs.shape == (4,) and "spine_observation" in info
    obs, reward, terminated, truncated, info = env.step(np.array([0.1], dtype=np.float32))
    assert reward == 0.0 and not terminated and not truncated


def test_vector_env_4096_throughput_path():
    env = envs.make("Upkie-HIP-Pendulum-Vec", num_envs=4096, frequency=200.0,
                    init_state=RobotState(randomization=RobotStateRandomization(pitch=0.1, x=0.05, omega_y=0.1, linear_velocity=np.array([0.05, 0, 0]))))
    obs, info = env.reset(seed=0)
    assert obs.is_cuda and obs.shape == (4096, 4)
    gains = torch.tensor([10.0, 1.0, 0.0, 0.1], device=obs.device)
    for _ in range(50):
        action = (obs @ gains).clamp(-0.99, 0.99)[:, None]
        obs, reward, terminated, truncated, info = env.step(action)
    assert not terminated.any() and float(obs[:, 0].abs().max()) < 0.3
    spine = info["spine_observation"]  # materialised lazily by one extra launch
    assert spine["base_orientation"]["pitch"].shape == (4096,)
    assert torch.allclose(spine["base_orientation"]["pitch"], obs[:, 0], atol=1e-6)
    assert bool(spine["floor_contact"]["contact"].all())


def test_base_velocity_env_gpu_matches_cpu_double():
    """UpkieBaseVelocity (MPC balancer in the loop, upkie_base_velocity.py:164-202)."""
    gpu = envs.make("Upkie-HIP-BaseVelocity-Vec", num_envs=64, frequency=200.0, nb_timesteps=16, autoreset=False)
    cpu = envs.make("Upkie-HIP-BaseVelocity-Vec", num_envs=64, frequency=200.0, nb_timesteps=16, autoreset=False,
                    sim_factory=oracle_sim_factory, mpc_factory=OracleMpc)
    gpu.reset(seed=5)
    cpu.reset(seed=5)
    act = torch.zeros(64, 2)
    act[:, 0] = torch.linspace(-0.3, 0.3, 64)
    act[:, 1] = 0.2
    for _ in range(100):
        og, _, tg, _, _ = gpu.step(act)
        oc, _, tc, _, _ = cpu.step(act)
    assert torch.equal(tg.cpu(), tc)
    np.testing.assert_allclose(og.cpu().numpy(), oc.numpy(), atol=1e-5)  # dead-reckoned pose
    vg, vc = gpu.mpc_balancer.commanded_velocity.cpu().numpy(), cpu.mpc_balancer.commanded_velocity.numpy()
    assert np.max(np.abs(vg - vc)) < 5e-3
    pg = gpu.sim.state[abi.S_POS].cpu().numpy()
    pc = cpu.sim.state[abi.S_POS].numpy()
    assert np.max(np.abs(pg - pc)) < 2e-3
    assert not bool(tg.any())  # the MPC keeps everyone upright


def test_base_velocity_fused_path_with_autoreset():
    """Fused MPC + Gyropod launches (the GPU path) vs the generic composition
    on the CPU doubles, with falls and NEXT_STEP autoresets in the loop."""
    kw = dict(num_envs=48, frequency=200.0, nb_timesteps=16, fall_pitch=0.12, seed=3,
              init_state=RobotState(randomization=RobotStateRandomization(pitch=0.1)))
    gpu = envs.make("Upkie-HIP-BaseVelocity-Vec", **kw)
    cpu = envs.make("Upkie-HIP-BaseVelocity-Vec", sim_factory=oracle_sim_factory, mpc_factory=OracleMpc, **kw)
    assert hasattr(gpu.sim, "step_base_velocity") and not hasattr(cpu.sim, "step_base_velocity")
    gpu.reset()
    cpu.reset()
    act = torch.zeros(48, 2)
    act[:, 0] = torch.linspace(-1.0, 1.0, 48)  # hard velocity steps tip some robots past 0.12 rad
    act[:, 1] = 0.3
    in_sync = np.ones(48, dtype=bool)
    falls = 0
    for _ in range(150):
        og, _, tg, _, _ = gpu.step(act)
        oc, _, tc, _, _ = cpu.step(act)
        in_sync &= tg.cpu().numpy() == tc.numpy()
        falls += int(tc.sum())
    assert falls > 0 and in_sync.mean() > 0.8
    np.testing.assert_allclose(og.cpu().numpy()[in_sync], oc.numpy()[in_sync], atol=1e-4)
    vg = gpu.mpc_balancer.commanded_velocity.cpu().numpy()[in_sync]
    vc = cpu.mpc_balancer.commanded_velocity.numpy()[in_sync]
    assert np.max(np.abs(vg - vc)) < 2e-2
    eg = gpu.sim.state[abi.S_EPISODE].cpu().numpy()[in_sync]
    ec = cpu.sim.state[abi.S_EPISODE].numpy()[in_sync]
    assert np.array_equal(eg, ec)


def test_spine_observers_in_the_vector_env_on_device():
    """spine_observers=True on the HIP path: estimator blocks appear in
    info["spine_observation"], balancing robots are seen in contact and the
    observers of autoreset envs restart."""
    import math

    B = 512
    env = envs.make(
        "Upkie-HIP-Pendulum-Vec",
        num_envs=B,
        frequency=200.0,  # 5 observer cycles (1 ms) inside every step
        fall_pitch=0.25,
        init_state=RobotState(position_base_in_world=np.array([0.0, 0.0, 0.58]), randomization=RobotStateRandomization(pitch=0.05)),
        spine_observers=True,
    )
    obs, info = env.reset(seed=1)
    first_episode = env.sim.state[abi.S_EPISODE].clone()
    for k in range(300):
        act = (10.0 * obs[:, 0] + obs[:, 1] + 0.1 * obs[:, 3]).clamp(-0.99, 0.99) + 0.4 * math.sin(0.05 * k)
        if k > 160:
            act[:16] = 3.0  # the first 16 envs are driven into a fall
        obs, _, terminated, _, info = env.step(act.reshape(B, 1))
    spine = info["spine_observation"]
    restarted = env.sim.state[abi.S_EPISODE] != first_episode
    assert restarted[:16].all()
    alive = ~restarted
    assert alive.float().mean() > 0.9
    assert spine["floor_contact"]["contact"][alive].float().mean() > 0.95
    assert spine["floor_contact"]["upper_leg_torque"].shape == (B,)
    assert spine["base_orientation"]["linear_velocity"].shape == (B, 3)
    truth = env.sim.observe(update_imu=False)
    assert torch.allclose(spine["base_orientation"]["pitch"], truth["pitch"], atol=1e-5)
    env.close()


def test_hip_spine_serves_one_env_of_a_gpu_batch():
    """An agent attached through the spine's shared memory to env #5 of a GPU
    batch while a device-side policy balances the other envs."""
    import os
    import threading
    import uuid

    from upkie_amd.envs.backends.spine_backend import SpineBackend
    from upkie_amd.spine import HipSpine, State

    B = 64
    name = f"/upkie_gpu_test_{os.getpid()}_{uuid.uuid4().hex[:6]}"
    env = envs.make("Upkie-HIP-Servos-Vec", num_envs=B, frequency=200.0, autoreset_mode="disabled")
    env.reset(seed=0)
    neutral = env.get_neutral_action()
    radius = float(env.sim.model.wheel_radius)

    def batch_policy(servo_obs):
        spine = env.sim.observe(update_imu=False)
        v = (10.0 * spine["pitch"] + spine["wheel_odometry"][:, 0] + 0.1 * spine["wheel_odometry"][:, 1]).clamp(-0.99, 0.99)
        act = neutral.clone()
        for j, sign in ((2, 1.0), (5, -1.0)):
            act[:, j, 1] = sign * v / radius
        act[:, [0, 1, 3, 4], 0] = 0.0  # legs held straight
        return act

    spine = HipSpine(env, shm_name=name, env_index=5, batch_policy=batch_policy)
    thread = threading.Thread(target=spine.run, kwargs=dict(idle_sleep=1e-4), daemon=True)
    thread.start()
    backend = SpineBackend(shm_name=name, retries=1, timeout_ns=5_000_000_000)
    obs = backend.reset(RobotState(position_base_in_world=np.array([0.0, 0.0, 0.58])))
    for _ in range(100):
        v = 10.0 * obs["base_orientation"]["pitch"] + obs["wheel_odometry"]["position"] + 0.1 * obs["wheel_odometry"]["velocity"]
        w = max(-0.99, min(0.99, v)) / radius
        legs = {"position": 0.0, "velocity": 0.0}
        obs = backend.step({"servo": {
            "left_wheel": {"position": float("nan"), "velocity": +w}, "right_wheel": {"position": float("nan"), "velocity": -w},
            "left_hip": legs, "left_knee": legs, "right_hip": legs, "right_knee": legs}})
    assert abs(obs["base_orientation"]["pitch"]) < 0.1 and obs["floor_contact"]["contact"] is True
    assert obs["time"] == pytest.approx(0.5)
    pitch = env.sim.observe(update_imu=False)["pitch"]
    assert pitch.abs().max() < 0.2  # the device-side policy kept the rest of the batch up
    backend.close()
    spine.interrupt()
    thread.join(timeout=10.0)
    assert spine.state_machine.state == State.kOver
    spine.close()
    env.close()


def test_graph_captured_loop_equals_eager():
    """hipGraph capture of policy ops + env.step (upkie_amd.graphs): same
    trajectory as the eager loop, bit for bit."""
    from upkie_amd.graphs import GraphedLoop
    from upkie_amd.sim import BatchedSim

    def make():
        cfg = abi.default_sim_config(512, seed=3)
        cfg.rand_pitch = 0.1
        cfg.autoreset_mode = abi.AUTORESET_NEXT_STEP
        sim = BatchedSim(cfg)
        sim.reset()
        sim.obs4.copy_(sim.obs6[:, [1, 0, 4, 3]])
        return sim

    gain = torch.tensor([10.0, 1.0, 0.0, 0.1], device="cuda:0")
    eager, graphed = make(), make()
    act_e = torch.zeros(512, device="cuda:0")
    act_g = torch.zeros(512, device="cuda:0")

    def body(sim, act):
        torch.matmul(sim.obs4, gain, out=act)
        act.clamp_(-0.99, 0.99)
        sim.step_pendulum(act)

    loop = GraphedLoop(lambda: body(graphed, act_g), unroll=4, warmup=2)
    # the warm-up and the capture itself advanced `graphed`: bring both to the same state
    eager.state.copy_(graphed.state)
    eager.obs4.copy_(graphed.obs4)
    for _ in range(10):
        loop.replay()
    for _ in range(40):
        body(eager, act_e)
    torch.cuda.synchronize()
    assert torch.equal(eager.state, graphed.state) and torch.equal(eager.obs4, graphed.obs4)


def test_two_graphs_of_one_handle_replay_each_with_its_own_settings():
    """include/upkie_hip.h, "Streams and hipGraphs" (round 5; ADVICE r3 #4 / VERDICT r4 #12): every hipGraph capture of a
    handle's steps gets a settings block of its own. Two graphs of ONE handle recorded at different settings (the fused
    agent's clip: 0.99 and 0 -- the second one commands no velocity at all), replayed alternately between eager launches
    at a third setting, each step with the settings it was recorded with: bit-equal to an eager handle whose settings are
    changed before every step. (Until round 4 all captures of a handle shared one block: the graph recorded first would
    have replayed with the second one's clip.) The ninth capture of a handle is refused with a message."""
    from upkie_amd.graphs import GraphedLoop
    from upkie_amd.lib import UpkieHipError
    from upkie_amd.sim import BatchedSim

    def make():
        cfg = abi.default_sim_config(512, seed=5)
        cfg.rand_pitch = 0.1
        cfg.autoreset_mode = abi.AUTORESET_NEXT_STEP
        sim = BatchedSim(cfg)
        sim.reset()
        sim.obs4.copy_(sim.obs6[:, [1, 0, 4, 3]])
        return sim

    def set_clip(sim, clip):
        sim.config.agent_clip = clip
        sim.push_config()

    eager, graphed = make(), make()
    assert graphed.lanes_per_env == 8  # (the eight-lane kernels are the ones that read the settings from device blocks)
    set_clip(graphed, 0.99)
    loop_a = GraphedLoop(lambda: graphed.step_pendulum_agent(), unroll=1, warmup=2)
    set_clip(graphed, 0.0)
    loop_b = GraphedLoop(lambda: graphed.step_pendulum_agent(), unroll=1, warmup=2)
    set_clip(graphed, 0.5)  # what eager launches of this handle use from here on
    eager.state.copy_(graphed.state)
    eager.obs4.copy_(graphed.obs4)
    schedule = ["a", "b", "e", "a", "a", "b", "e", "b", "a"]
    for what in schedule:
        {"a": loop_a.replay, "b": loop_b.replay, "e": graphed.step_pendulum_agent}[what]()
        set_clip(eager, {"a": 0.99, "b": 0.0, "e": 0.5}[what])
        eager.step_pendulum_agent()
    torch.cuda.synchronize()
    assert torch.equal(eager.state, graphed.state) and torch.equal(eager.obs4, graphed.obs4)
    assert float(graphed.state[abi.S_QD + 2].abs().max()) > 0.0
    # captures beyond UPKIE_MAX_GRAPH_CAPTURES are refused, loudly, and eager launches go on working
    loops = [loop_a, loop_b]
    with pytest.raises((UpkieHipError, RuntimeError)):
        for _ in range(abi.MAX_GRAPH_CAPTURES):
            loops.append(GraphedLoop(lambda: graphed.step_pendulum_agent(), unroll=1, warmup=1))
    torch.cuda.synchronize()
    assert len(loops) == abi.MAX_GRAPH_CAPTURES  # two + six more were recorded, the ninth was not
    graphed.step_pendulum_agent()
    torch.cuda.synchronize()
    # round 6: a long-lived env that re-captures hands the blocks back once its old graphs are dead
    del loops, loop_a, loop_b
    graphed.release_graph_captures()
    again = [GraphedLoop(lambda: graphed.step_pendulum_agent(), unroll=1, warmup=1) for _ in range(abi.MAX_GRAPH_CAPTURES)]
    eager.state.copy_(graphed.state)
    eager.obs4.copy_(graphed.obs4)
    set_clip(eager, 0.5)
    for loop in again[:3]:
        loop.replay()
        eager.step_pendulum_agent()
    torch.cuda.synchronize()
    assert torch.equal(eager.state, graphed.state)


@pytest.mark.parametrize("lanes", ["1", "2", "8"])
@pytest.mark.parametrize("env_id,limit", [("Upkie-HIP-Pendulum-Vec", None), ("Upkie-HIP-Gyropod-Vec", 25), ("Upkie-HIP-Servos-Vec", 12)])
def test_same_step_autoreset_is_one_launch_and_matches_the_double(env_id, limit, lanes, monkeypatch):
    """SAME_STEP autoreset through upkie_sim_autoreset_done (the reset branch
    of the step kernel run on the envs whose DONE word is set) against the
    oracle-backed double: same terminal observations, same restart
    observations, same flags, falls and time limits alike."""
    from upkie_amd.utils.robot_state import RobotState
    from upkie_amd.utils.robot_state_randomization import RobotStateRandomization

    monkeypatch.setenv("UPKIE_LANES_PER_ENV", lanes)
    B = 96
    kw = dict(num_envs=B, frequency=200.0, init_state=RobotState(randomization=RobotStateRandomization(pitch=0.12)),
              autoreset_mode="same_step", max_episode_steps=limit, seed=4)
    if "Servos" not in env_id:
        kw["fall_pitch"] = 0.15
    gpu = envs.make(env_id, **kw)
    cpu = envs.make(env_id, sim_factory=oracle_sim_factory, **kw)
    assert hasattr(gpu.sim, "autoreset_done") and gpu._same_step_layout is not None
    og, _ = gpu.reset(seed=4)
    oc, _ = cpu.reset(seed=4)
    act = gpu.get_neutral_action() if "Servos" in env_id else torch.zeros((B,) + tuple(gpu.single_action_space.shape))
    if "Servos" in env_id:
        act[:, :, 3:5] = 0.0  # limp joints: the robots collapse, states move a lot between resets
    ended = 0
    for step in range(60):
        og, rg, tg, ug, ig = gpu.step(act)
        oc, rc, tc, uc, ic = cpu.step(act.cpu())
        assert torch.equal(tg.cpu(), tc) and torch.equal(ug.cpu(), uc), step
        m = ic["_final_obs"]
        assert torch.equal(ig["_final_obs"].cpu(), m)
        ended += int(m.sum())
        # fp32 vs fp64 closed loops drift: compare what the autoreset itself produces tightly, the rest loosely
        np.testing.assert_allclose(og.cpu()[m].numpy(), oc[m].numpy(), atol=3e-3)
        np.testing.assert_allclose(ig["final_obs"].cpu().numpy(), ic["final_obs"].numpy(), atol=0.05 if "Servos" not in env_id else 0.5)
        np.testing.assert_allclose(og.cpu().numpy(), oc.numpy(), atol=0.05 if "Servos" not in env_id else 0.5)
        # resynchronise the double with the device so that flags keep agreeing
        cpu.sim._o.state[:] = gpu.sim.state.cpu().numpy().astype(np.float64)
    assert ended >= B // 4
    np.testing.assert_array_equal(gpu.sim.state[abi.S_EPISODE].cpu().numpy(), cpu.sim._o.state[abi.S_EPISODE])


def test_base_velocity_same_step_autoreset_reports_the_dead_reckoned_pose():
    """UpkieBaseVelocity with autoreset_mode="same_step" on the fused GPU path:
    the dead-reckoned x, y live in the state words (upkie_base_velocity.py:
    197-199); every step goes through reset(mask=done), whose observation must
    carry them for the envs that did not restart and zeros for those that did."""
    from upkie_amd.utils.robot_state import RobotState
    from upkie_amd.utils.robot_state_randomization import RobotStateRandomization

    kw = dict(num_envs=48, frequency=200.0, nb_timesteps=16, fall_pitch=0.12, seed=3, autoreset_mode="same_step",
              init_state=RobotState(randomization=RobotStateRandomization(pitch=0.1)))
    gpu = envs.make("Upkie-HIP-BaseVelocity-Vec", **kw)
    cpu = envs.make("Upkie-HIP-BaseVelocity-Vec", sim_factory=oracle_sim_factory, mpc_factory=OracleMpc, **kw)
    assert hasattr(gpu.sim, "step_base_velocity")
    gpu.reset()
    cpu.reset()
    act = torch.zeros(48, 2)
    act[:, 0] = torch.linspace(-1.0, 1.0, 48)
    act[:, 1] = 0.3
    in_sync = np.ones(48, dtype=bool)
    ended = 0
    for step in range(120):
        og, _, tg, _, ig = gpu.step(act)
        oc, _, tc, _, ic = cpu.step(act)
        in_sync &= tg.cpu().numpy() == tc.numpy()
        ended += int(tc.sum())
        if step == 40:
            moving = og[:, :2].abs().sum(dim=1).cpu().numpy() > 1e-3
            assert moving[in_sync].mean() > 0.8  # the pose is reported, not zeros
    assert ended > 0 and in_sync.mean() > 0.8
    np.testing.assert_allclose(og.cpu().numpy()[in_sync], oc.numpy()[in_sync], atol=1e-4)
    # explicit masked reset in any mode: untouched envs keep their pose
    mask = torch.zeros(48, dtype=torch.uint8)
    mask[::2] = 1
    before = og.clone()
    obs, _ = gpu.reset(mask=mask)
    assert torch.equal(obs[1::2, :2].cpu(), before[1::2, :2].cpu()) and float(obs[::2, :2].abs().max()) == 0.0


@pytest.mark.parametrize("lanes", ["2", "8"])
@pytest.mark.parametrize("mode", ["pendulum", "gyropod", "servos"])
def test_same_step_autoreset_inside_the_step_equals_the_two_calls(mode, lanes, monkeypatch):
    """`upkie_sim_set_final_observation`: the step call completes a SAME_STEP
    autoreset itself -- inside the launch on eight lanes per env (the finished
    env's lanes go through the reset branch once more), by a second launch
    behind the first on the other mappings -- with the bits of an explicit
    `upkie_sim_step_*` + `upkie_sim_autoreset_done` pair: observations, last
    observations, flags, state, falls and time limits alike."""
    from upkie_amd.sim import BatchedSim

    monkeypatch.setenv("UPKIE_LANES_PER_ENV", lanes)
    B = 333  # a half-filled last wavefront
    cfg = randomized_config(B, seed=9)
    cfg.autoreset_mode = abi.AUTORESET_DISABLED
    cfg.fall_pitch = 0.12
    cfg.rand_pitch = 0.1
    cfg.max_episode_steps = 17
    cfg.torque_measurement_noise[1] = 0.02
    layout, shape = {"pendulum": (abi.OBSERVATION_PENDULUM, (B, 4)), "gyropod": (abi.OBSERVATION_GYROPOD, (B, 6)), "servos": (abi.OBSERVATION_SERVOS, (B, 6, 5))}[mode]
    fused, pair = BatchedSim(cfg), BatchedSim(cfg)
    assert fused.lanes_per_env == int(lanes)
    for sim in (fused, pair):
        sim.reset()
    final_fused = torch.full(shape, -7.0, device="cuda:0")
    final_pair = torch.full(shape, -7.0, device="cuda:0")
    fused.set_final_observation(final_fused)
    rng = np.random.default_rng(1)
    restarted = 0
    for step in range(50):
        if mode == "pendulum":
            act = torch.from_numpy(rng.uniform(-0.5, 0.5, B).astype(np.float32))
            a, b = fused.step_pendulum(act), pair.step_pendulum(act)
        elif mode == "gyropod":
            act = torch.from_numpy(rng.uniform(-0.5, 0.5, (B, 2)).astype(np.float32))
            a, b = fused.step_gyropod(act), pair.step_gyropod(act)
        else:
            act = np.zeros((B, 6, 6), dtype=np.float32)
            act[:, :, 0] = rng.uniform(-0.2, 0.2, (B, 6))
            act[:, [2, 5], 0] = np.nan
            act[:, :, 3:5] = 0.3  # soft joints: some robots sag and the time limit takes the rest
            act[:, :, 5] = 16.0
            a, b = fused.step_servos(torch.from_numpy(act)), pair.step_servos(torch.from_numpy(act))
        restarted += int((pair.state[abi.S_DONE] != 0).sum())
        pair.autoreset_done(layout, b[0], final_pair)
        for x, y in zip(a, b):
            assert torch.equal(x, y), (mode, step)
        assert torch.equal(final_fused, final_pair), (mode, step)
        assert torch.equal(fused.state, pair.state), (mode, step)
    assert restarted > B  # every env ran into the time limit at least once, some fell before
    fused.set_final_observation(None)  # back to flag-only steps
    act0 = torch.zeros(shape[:1] if mode == "pendulum" else ((B, 2) if mode == "gyropod" else (B, 6, 6)), device="cuda:0")
    before = final_fused.clone()
    (fused.step_pendulum if mode == "pendulum" else fused.step_gyropod if mode == "gyropod" else fused.step_servos)(act0)
    assert torch.equal(final_fused, before)


def test_fused_servo_policy_step_and_same_step_resets_replay_from_a_graph():
    """The calls added in round 2 are plain kernel launches on the caller's
    stream (the policy reaches the device as a kernel argument, the SAME_STEP
    buffer is a pointer held by the handle): captured in a hipGraph they
    replay the trajectory of the eager loop bit for bit."""
    from upkie_amd.graphs import GraphedLoop
    from upkie_amd.sim import BatchedSim

    def make():
        cfg = randomized_config(640, seed=12)
        cfg.autoreset_mode = abi.AUTORESET_DISABLED
        cfg.rand_pitch = 0.2
        cfg.max_episode_steps = 9
        sim = BatchedSim(cfg)
        sim.reset()
        sim.set_final_observation(torch.zeros(640, 6, 5, device="cuda:0"))
        return sim

    eager, graphed = make(), make()
    policy = abi.velocity_balancing_policy(float(eager.model.wheel_radius), 0.5, float(eager.model.left_sign))
    changed = abi.torque_balancing_policy(gain=8.0, fall_pitch=0.5, left_sign=float(eager.model.left_sign))
    loop = GraphedLoop(lambda: graphed.step_servos_policy(policy), unroll=3, warmup=1)
    eager.state.copy_(graphed.state)
    eager.final_obs.copy_(graphed.final_obs)
    for _ in range(8):
        loop.replay()
    for _ in range(24):
        eager.step_servos_policy(policy)
    assert torch.equal(eager.state, graphed.state) and torch.equal(eager.obs_servos, graphed.obs_servos)
    assert torch.equal(eager.final_obs, graphed.final_obs)
    assert int(eager.state[abi.S_EPISODE].sum()) > 640 * 2  # the time limit restarted every env, inside the step's launch
    # another policy afterwards: uploaded by the next call, eagerly or not
    for sim in (eager, graphed):
        sim.step_servos_policy(changed)
    assert torch.equal(eager.state, graphed.state)


@pytest.mark.parametrize("env_id", ["Upkie-HIP-Pendulum-Vec", "Upkie-HIP-Servos-Vec"])
def test_graphed_env_step_starts_from_the_reset_state(env_id):
    """`GraphedEnvStep`'s warm-up used to advance the env three steps (possibly through autoresets) between
    the user's `reset()` and the first graphed step, and a Servos env's policy read an all-zero observation first. Now
    the state is snapshot before the warm-up and restored after the capture: the graphed loop replays, bit for bit,
    what the eager loop `obs = env.step(policy(obs))` does from `reset()`; env kinds that compose their step refuse."""
    from upkie_amd.exceptions import UpkieRuntimeError
    from upkie_amd.graphs import GraphedEnvStep

    B = 384
    init = lambda: RobotState(randomization=RobotStateRandomization(pitch=0.2))  # noqa: E731
    kw = dict(num_envs=B, frequency=200.0, seed=9, fall_pitch=0.3) if "Pendulum" in env_id else dict(num_envs=B, frequency=200.0, seed=9)
    eager, graphed = envs.make(env_id, init_state=init(), **kw), envs.make(env_id, init_state=init(), **kw)
    if "Pendulum" in env_id:
        gain = torch.tensor([10.0, 1.0, 0.0, 0.1], device="cuda:0")
        policy = lambda o: (o @ gain).clamp(-0.99, 0.99).unsqueeze(1)  # noqa: E731
    else:
        neutral = eager.get_neutral_action()

        def policy(o):
            a = neutral.clone()
            a[:, 2, 1] = (20.0 * o[:, 0, 0]).clamp(-5.0, 5.0)  # any function of the observation: wheel velocity from the hip angle
            a[:, 5, 1] = -a[:, 2, 1]
            return a
    oe, _ = eager.reset(seed=9)
    og, _ = graphed.reset(seed=9)
    assert graphed.observation is og and float(og.abs().sum()) > 0.0  # (Servos: allocated and filled by reset, not by the first step)
    step = GraphedEnvStep(graphed, policy, warmup=3)
    assert torch.equal(eager.sim.state, graphed.sim.state) and torch.equal(oe, og)  # nothing moved
    for _ in range(12):
        oe = eager.step(policy(oe))[0]
        og = step()[0]
    torch.cuda.synchronize()
    assert torch.equal(eager.sim.state, graphed.sim.state) and torch.equal(oe, og)
    bv = envs.make("Upkie-HIP-BaseVelocity-Vec", num_envs=64, frequency=200.0, nb_timesteps=16)
    bv.reset(seed=0)
    with pytest.raises(UpkieRuntimeError, match="GraphedLoop"):
        GraphedEnvStep(bv, lambda o: torch.zeros(64, 2, device="cuda:0"))
    for e in (eager, graphed, bv):
        e.close()
