"""The Bullet-like contact model ON THE DEVICE (`upkie_sim_set_contact_manifold`,
upkie_amd/csrc/bullet_like.hpp) against the oracle's `bullet_like` mode
(oracle/upkie_oracle.c::bullet_like_contacts) over whole env.step() rollouts of
BASELINE's C2 and C5-share workloads, at the fp32 tolerances the default
contact model is held to. (Substep by substep, manifold bookkeeping included:
tests/test_bullet_like_on_host.py, no GPU.)"""

import numpy as np
import pytest
import torch

from upkie_amd import abi
from upkie_amd.model.default_model import default_model
from upkie_amd.model.model import Model
from upkie_amd.sim import BatchedSim

from .helpers import randomized_config, state_errors

pytestmark = pytest.mark.gpu


def manifolds(sim, ref):
    mh = sim.contact_manifold.cpu().numpy().astype(np.float64).reshape(2, 4, 8, -1)
    mo = ref.bullet_manifold.reshape(2, 4, 8, -1)
    return mh, mo


@pytest.mark.parametrize("lanes", ["8", "1"])
def test_c2_rollout_under_the_bullet_like_model_matches_the_oracle(lanes, monkeypatch):
    """Upkie-Pendulum, README agent, 2048 envs x 60 steps, on the eight-lane
    variant (octet.hpp: what a batch of this size runs) and on the one-lane
    kernels (bullet_like.hpp: every case, any batch size): observations within
    the closed-loop tolerances of the default model, the same points cached,
    applied normal impulses within 1e-4 N.s."""
    from oracle import oracle as O

    monkeypatch.setenv("UPKIE_LANES_PER_ENV", lanes)
    B = 2048
    cfg = randomized_config(B, seed=3)
    sim = BatchedSim(cfg)
    sim.use_bullet_like_contacts()
    assert sim.lanes_per_env == int(lanes) and sim.lanes_per_env_of(abi.OBSERVATION_SERVOS) == int(lanes)  # (round 5: Servos steps too)
    ref = O.Oracle(default_model(), cfg)
    ref.use_bullet_like_contacts()
    sim.reset()
    obs_ref = ref.reset()[:, [1, 0, 4, 3]]
    sim.obs4.copy_(sim.obs6[:, [1, 0, 4, 3]])
    worst = np.zeros((B, 4))
    for k in range(60):
        obs_ref, _, term_ref, _ = ref.step_pendulum_agent(obs_ref)
        obs, _, term, _ = sim.step_pendulum_agent()
        worst = np.maximum(worst, np.abs(obs.cpu().numpy() - obs_ref))
        assert np.array_equal(term.cpu().numpy(), term_ref)
    q = {p: np.quantile(worst, p, axis=0) for p in (0.5, 0.99, 1.0)}
    print("bullet-like C2: |obs - oracle| quantiles", {k: v.round(7).tolist() for k, v in q.items()})
    assert q[1.0][0] <= 1e-3 and q[1.0][1] <= 1e-3, q  # SURVEY A.9 closed-loop tolerance, every env
    assert q[0.5][0] <= 2e-6 and q[0.5][1] <= 5e-6 and q[0.99][0] <= 1e-4 and q[0.99][1] <= 1e-4, q
    mh, mo = manifolds(sim, ref)
    assert np.array_equal(mh[:, :, 7] != 0, mo[:, :, 7] != 0)
    assert (mo[:, :, 7].sum(axis=1) == 1).all()  # rolling wheels: one cached point per tire
    live = mo[:, :, 7] != 0
    assert np.abs(mh[:, :, 6] - mo[:, :, 6])[live].max() < 1e-4 and mo[:, :, 6][live].min() > 0.01  # ~0.026 N.s per tire and substep
    assert np.abs(mh[:, :, :3] - mo[:, :, :3]).transpose(0, 1, 3, 2)[live].max() < 2e-4  # the cached points, wheel frame (fp32 sine / cosine of a wheel angle of tens of radians x 5 cm; measured 4e-5)
    err = state_errors(ref.state, sim.state_numpy())
    assert err["pos"] < 1e-3 and err["quat"] < 1e-3 and err["contact"] == 0, err


@pytest.mark.parametrize("lanes", ["8", "1"])
def test_autoreset_clears_the_manifold_and_falls_match_the_oracle(lanes, monkeypatch):
    """NEXT_STEP autoreset with a small fall pitch: envs fall and restart inside
    the 80 steps (two thirds of them); a reset drops the env's contact cache on both sides."""
    from oracle import oracle as O

    monkeypatch.setenv("UPKIE_LANES_PER_ENV", lanes)
    B = 512
    cfg = randomized_config(B, seed=5, autoreset=True)
    cfg.fall_pitch = 0.12
    sim = BatchedSim(cfg)
    sim.use_bullet_like_contacts()
    ref = O.Oracle(default_model(), cfg)
    ref.use_bullet_like_contacts()
    sim.reset()
    ref.reset()
    act = torch.zeros(B, device=sim.device)  # no balancing: every robot falls
    steps = 80
    ends_h, ends_o = np.zeros((steps, B), dtype=np.uint8), np.zeros((steps, B), dtype=np.uint8)
    worst = 0.0
    for k in range(steps):
        oh, _, th, _ = sim.step_pendulum(act)
        oo, _, to, _ = ref.step_pendulum(np.zeros(B))
        ends_h[k], ends_o[k] = th.cpu().numpy(), to
        in_phase = (ends_h[:k + 1] == ends_o[:k + 1]).all(axis=0)  # (an env whose fall lands on the next step is out of phase from there on)
        worst = max(worst, float(np.abs(oh.cpu().numpy() - oo)[in_phase][:, :2].max()))
    from .test_timed_windows_gpu import compare_falls

    report = compare_falls(ends_h, ends_o)
    print("bullet-like autoreset, lanes", lanes, report, "worst |pitch, position| in phase", worst)
    assert report["episodes_ended_oracle"] >= B // 4  # (measured: 337 of 512 envs fell and restarted inside the 80 steps)
    assert report["envs_every_end_within_1_step"] >= 0.99 and abs(report["episodes_ended_device"] - report["episodes_ended_oracle"]) <= 3, report
    assert worst < 1e-3  # robots falling on locked wheels (measured: 1.2e-4 on one lane, 3.7e-4 on eight)
    mh, mo = manifolds(sim, ref)
    in_phase = (ends_h == ends_o).all(axis=0)
    assert np.array_equal((mh[:, :, 7].sum(axis=1) != 0)[:, in_phase], (mo[:, :, 7].sum(axis=1) != 0)[:, in_phase])  # which tires hold a point


@pytest.mark.parametrize("lanes", ["8", "1"])
def test_c5_share_under_the_bullet_like_model_matches_the_oracle(lanes, monkeypatch):
    """Servos env, per-link inertia randomisation, a push on the torso, wheel
    friction, the torque-balancing action: 1024 envs x 10 steps (the RAND
    instantiations of the Bullet-like kernels): the eight-lane Servos kernel
    (round 5) and the one-lane kernels."""
    from oracle import oracle as O

    monkeypatch.setenv("UPKIE_LANES_PER_ENV", lanes)
    B = 1024
    cfg = randomized_config(B, seed=2)
    cfg.joint_friction[2] = cfg.joint_friction[5] = 0.1
    model = Model().struct
    sim = BatchedSim(cfg, model)
    sim.use_bullet_like_contacts()
    assert sim.lanes_per_env_of(abi.OBSERVATION_SERVOS) == int(lanes)
    ref = O.Oracle(model, cfg)
    ref.use_bullet_like_contacts()
    sim.randomize_inertias(0.2)
    ref.body_inertials = ref.sample_body_inertials(0.2)
    rng = np.random.default_rng(5)
    angle, norm = rng.uniform(0, 2 * np.pi, B), rng.uniform(0.0, 20.0, B)
    force = np.stack([norm * np.cos(angle), norm * np.sin(angle), np.zeros(B)])
    ref.ext_force = force
    ref.ext_point = np.array([0.0, 0.0, -0.1])
    sim.set_external_force(torch.from_numpy(force).float(), point=(0.0, 0.0, -0.1))
    obs_o = ref.reset()
    sim.reset()
    act = np.zeros((B, 6, 6))
    act[:, :, 3] = 1.0
    act[:, :, 4] = 1.0
    act[:, :, 5] = 16.0
    act[:, [2, 5], 0] = np.nan
    act[:, [2, 5], 4] = 0.0
    pitch = obs_o[:, 1]
    for _ in range(10):
        act[:, 2, 2] = 10.0 * pitch
        act[:, 5, 2] = -10.0 * pitch
        so, _, _, _ = ref.step_servos(act)
        sh, _, term, _ = sim.step_servos(torch.from_numpy(act).float())
        st = ref.state
        pitch = np.arcsin(np.clip(2.0 * (st[abi.S_QUAT] * st[abi.S_QUAT + 2] - st[abi.S_QUAT + 3] * st[abi.S_QUAT + 1]), -1, 1))
    err = state_errors(ref.state, sim.state_numpy())
    print("bullet-like C5 share, state errors after 10 steps", err)
    assert err["pos"] < 2e-4 and err["quat"] < 2e-4, err
    assert err["linvel"] < 5e-3 and err["angvel"] < 2e-2, err
    sh = sh.cpu().numpy()
    dq = np.abs(sh[:, :, 0] - so[:, :, 0])
    assert dq[:, [0, 1, 3, 4]].max() <= 1e-3 and np.quantile(dq, 0.5) <= 1e-5, (dq.max(axis=0), np.quantile(dq, [0.5, 0.99]))
    assert np.quantile(dq[:, [2, 5]], 0.99) <= 1e-3, np.quantile(dq[:, [2, 5]], [0.5, 0.99, 1.0])


def test_both_bullet_like_kernels_continue_from_each_others_manifold():
    """Two handles, one on the eight-lane kernels and one on the one-lane kernels all the way, step Pendulum and then
    Servos on their manifolds: the eight-lane kernel writes complete records (point in the wheel frame, on the plane,
    applied impulse, live) -- what the one-lane kernel keeps -- so either kernel finds its cached points where it would
    have put them itself (a handle whose Servos batch outgrows the eight-lane mapping, or whose caller forces a mapping,
    continues on the other kernel's manifold)."""
    B = 256
    import os

    a = BatchedSim(randomized_config(B, seed=8))
    a.use_bullet_like_contacts()
    assert a.lanes_per_env == 8
    a.reset()
    os.environ["UPKIE_LANES_PER_ENV"] = "1"
    try:
        c = BatchedSim(randomized_config(B, seed=8))  # the one-lane kernels all the way
    finally:
        os.environ.pop("UPKIE_LANES_PER_ENV")
    c.use_bullet_like_contacts()
    c.reset()
    for sim in (a, c):
        sim.obs4.copy_(sim.obs6[:, [1, 0, 4, 3]])
    for _ in range(10):
        a.step_pendulum_agent()
        c.step_pendulum_agent()
    ma, mc = a.contact_manifold.cpu().numpy().reshape(2, 4, 8, B), c.contact_manifold.cpu().numpy().reshape(2, 4, 8, B)
    assert (ma[:, :, 7].sum(axis=1) == 1).all() and (mc[:, :, 7].sum(axis=1) == 1).all()
    la, lc = ma[:, :, 7] != 0, mc[:, :, 7] != 0
    pa = np.stack([ma[:, :, i][la] for i in range(7)])  # the live record of every tire, whichever slot holds it
    pc = np.stack([mc[:, :, i][lc] for i in range(7)])
    assert np.abs(pa[:3] - pc[:3]).max() < 2e-4 and np.abs(pa[3:5] - pc[3:5]).max() < 2e-4 and np.abs(pa[6] - pc[6]).max() < 2e-4
    # ... and a Servos step (one lane) continues from the manifold the eight-lane steps left
    act = torch.zeros((B, 6, 6), device=a.device)
    act[:, :, 3] = 1.0
    act[:, :, 4] = 1.0
    act[:, :, 5] = 16.0
    act[:, [2, 5], 0] = float("nan")
    oa = a.step_servos(act)[0].clone()
    oc = c.step_servos(act)[0]
    assert (oa[:, :, :2] - oc[:, :, :2]).abs().max() < 5e-3
    m2 = a.contact_manifold.cpu().numpy().reshape(2, 4, 8, B)
    assert (m2[:, :, 7].sum(axis=1) == 1).all()  # the cached point was replaced, not doubled


def test_contact_point_query_solves_the_bullet_like_model():
    """`get_contact_points` (pybullet_backend.py:660-716) on a handle under the
    Bullet-like model: the cached points and the forces their impulses sum to,
    device against oracle, and the weight carried by the two tires."""
    from oracle import oracle as O

    B = 256
    cfg = randomized_config(B, seed=12)
    sim = BatchedSim(cfg)
    sim.use_bullet_like_contacts()
    ref = O.Oracle(default_model(), cfg)
    ref.use_bullet_like_contacts()
    sim.reset()
    obs_ref = ref.reset()[:, [1, 0, 4, 3]]
    sim.obs4.copy_(sim.obs6[:, [1, 0, 4, 3]])
    for _ in range(40):
        obs_ref = ref.step_pendulum_agent(obs_ref)[0]
        sim.step_pendulum_agent()
    ph, po = sim.contact_points().cpu().numpy().astype(np.float64), ref.contact_points()
    assert np.array_equal(ph[:, :, 0], po[:, :, 0]) and (po[:, :, 0] == 1).all()
    assert np.abs(ph[:, :, 1:4] - po[:, :, 1:4]).max() < 2e-5  # the cached points, world frame
    df = np.abs(ph[:, :, 4:7] - po[:, :, 4:7])
    assert np.quantile(df, 0.5) < 2e-2 and np.quantile(df, 0.99) < 1.0, (np.quantile(df, [0.5, 0.99]), df.max())  # forces [N]: impulses / 1 ms
    weight = 9.81 * float(sum(default_model().mass[:7]))
    assert abs(np.median(ph[:, :, 6].sum(axis=1)) - weight) < 0.05 * weight
    # the query left the manifold alone
    before = sim.contact_manifold.clone()
    sim.contact_points()
    assert torch.equal(before, sim.contact_manifold)


def test_switching_the_model_off_restores_the_default_kernels():
    B = 256
    cfg = randomized_config(B, seed=1)
    a, b = BatchedSim(cfg), BatchedSim(cfg)
    b.use_bullet_like_contacts()
    b.use_bullet_like_contacts(False)
    assert b.lanes_per_env == a.lanes_per_env == 8
    a.reset()
    b.reset()
    a.obs4.copy_(a.obs6[:, [1, 0, 4, 3]])
    b.obs4.copy_(b.obs6[:, [1, 0, 4, 3]])
    for _ in range(5):
        oa = a.step_pendulum_agent()[0]
        ob = b.step_pendulum_agent()[0]
    assert torch.equal(oa, ob)


def test_same_step_autoreset_inside_the_bullet_like_launch_matches_the_oracle():
    """Round 5: the eight-lane Bullet-like kernels have their IN_PLACE instantiations -- a SAME_STEP env under
    `contact_model="bullet_like"` ends an episode, keeps its last observation and restarts the env (dropping its contact
    cache) inside ONE launch, as the default model's kernels do. Public vector env on the device against the same env on
    the oracle double, robots left to fall (fall pitch 0.12): a third of the envs fall and restart within the 90 steps."""
    import upkie_amd.envs as envs
    from upkie_amd.utils.robot_state import RobotState
    from upkie_amd.utils.robot_state_randomization import RobotStateRandomization

    from .fake_sim import oracle_sim_factory
    from .test_timed_windows_gpu import compare_falls

    B, steps = 512, 90
    init = lambda: RobotState(randomization=RobotStateRandomization(pitch=0.05, x=0.02, omega_y=0.05))  # noqa: E731
    kw = dict(num_envs=B, frequency=200.0, seed=4, autoreset_mode="same_step", contact_model="bullet_like", fall_pitch=0.12)
    gpu = envs.make("Upkie-HIP-Pendulum-Vec", init_state=init(), **kw)
    cpu = envs.make("Upkie-HIP-Pendulum-Vec", init_state=init(), sim_factory=oracle_sim_factory, **kw)
    assert gpu.sim.lanes_per_env == 8
    gpu.reset(seed=4)
    cpu.reset(seed=4)
    act_g, act_c = torch.zeros(B, 1, device=gpu.device), torch.zeros(B, 1)
    ends_g, ends_c = np.zeros((steps, B), dtype=np.uint8), np.zeros((steps, B), dtype=np.uint8)
    final_err, restart_err = [], []
    for k in range(steps):
        og, _, tg, _, ig = gpu.step(act_g)
        oc, _, tc, _, ic = cpu.step(act_c)
        ends_g[k], ends_c[k] = tg.cpu().numpy(), tc.numpy()
        same = (ends_g[k] != 0) & (ends_c[k] != 0) & (ends_g[:k].sum(axis=0) == ends_c[:k].sum(axis=0))
        if same.any():
            final_err.append(np.abs(ig["final_obs"].cpu().numpy()[same].astype(np.float64) - ic["final_obs"].numpy()[same])[:, :2].max())
            restart_err.append(np.abs(og.cpu().numpy()[same].astype(np.float64) - oc.numpy()[same])[:, :2].max())
            assert torch.equal(ig["_final_obs"].cpu(), tg.cpu())
    report = compare_falls(ends_g, ends_c)
    print("bullet-like SAME_STEP in the launch:", report, "final_obs", max(final_err), "restart", max(restart_err))
    assert report["episodes_ended_oracle"] >= B // 4 and report["envs_every_end_within_1_step"] >= 0.99, report  # (measured: 175 ends in 90 steps, every one on the oracle's step)
    assert abs(report["episodes_ended_device"] - report["episodes_ended_oracle"]) <= 3, report
    assert max(final_err) < 2e-3 and max(restart_err) < 2e-5  # robots tipping over on locked wheels; the restarted env: the same draw
    # the restarted envs' manifolds: one fresh point per tire on both sides
    mh = gpu.sim.contact_manifold.cpu().numpy().reshape(2, 4, 8, B)
    mo = cpu.sim._o.bullet_manifold.reshape(2, 4, 8, B)
    in_phase = (ends_g == ends_c).all(axis=0)
    assert np.array_equal((mh[:, :, 7].sum(axis=1) != 0)[:, in_phase], (mo[:, :, 7].sum(axis=1) != 0)[:, in_phase])
    gpu.close()
    cpu.close()
