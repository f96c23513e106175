"""The non-finite guard of the step kernels (include/upkie_hip.h, "Non-finite
commands and states"; VERDICT r5 weak #4: a NaN velocity or feedforward torque
passed both clips of the torque law, the state went NaN, `fabsf(pitch) >
fall_pitch` is false for NaN -- never terminated, never reset, NaN observations
for ever). Reference: `assert not np.isnan(target_velocity)`
(pybullet_backend.py:519), fall detection upkie_gyropod.py:333-352.

Through the C-ABI, every env kind, every lane mapping:
  * NaN / +-Inf / 1e30 written into the actions of a fixed per cent of the envs at
    random steps for 2000 steps: no NaN ever leaves a step, and the OTHER envs
    are bit-equal to a run without poison;
  * the same poisoned actions on the fp64 checker: same replaced words, same
    counts, same states;
  * a state word / a push force / an inertial record that is not finite: the
    env reports `terminated`, comes back in the initial state and is
    re-initialised by the autoreset like a fallen robot.
"""

import numpy as np
import pytest
import torch

from upkie_amd import abi
from upkie_amd.model.default_model import default_model
from upkie_amd.mpc import BatchedMpc
from upkie_amd.sim import BatchedSim

from .test_oracle_guard import guard_counts as oracle_guard_counts
from .test_oracle_guard import neutral

pytestmark = pytest.mark.gpu

POISON = (float("nan"), float("inf"), float("-inf"), 1e30, -1e30)


def config(B, autoreset=abi.AUTORESET_NEXT_STEP, seed=5):
    cfg = abi.default_sim_config(B, frequency=200.0, seed=seed)
    cfg.rand_pitch = 0.1
    cfg.rand_x = 0.05
    cfg.rand_omega_y = 0.1
    cfg.autoreset_mode = autoreset
    return cfg


def make_sim(B, lanes, monkeypatch, **kw):
    monkeypatch.setenv("UPKIE_LANES_PER_ENV", str(lanes))
    sim = BatchedSim(config(B, **kw), default_model())
    monkeypatch.delenv("UPKIE_LANES_PER_ENV")
    sim.reset()
    return sim


def clean_action(kind, model, B, rng):
    if kind == "pendulum":
        return rng.uniform(-0.3, 0.3, B).astype(np.float32)
    if kind in ("gyropod", "base_velocity"):
        return np.stack([rng.uniform(-0.3, 0.3, B), rng.uniform(-0.5, 0.5, B)], axis=1).astype(np.float32)
    act = neutral(model, B)
    act[:, [2, 5], 1] = rng.uniform(-3.0, 3.0, (B, 2))  # wheel velocity targets
    act[:, [0, 1, 3, 4], 0] = 0.0  # legs held at zero
    return act.astype(np.float32)


def run(kind, lanes, poisoned_envs, steps, monkeypatch, seed=11):
    """`steps` steps of `kind` on B envs; envs in `poisoned_envs` get a POISON
    value written into a random word of their action with probability 0.3 per
    step. Returns per-step observations' finiteness, the final state, the
    terminated counts of the poisoned envs and the guard counters."""
    B = 1024
    sim = make_sim(B, lanes, monkeypatch)
    model = sim.model
    mpc = None
    if kind == "base_velocity":
        mpc = BatchedMpc(abi.default_mpc_config(B, nb_timesteps=16), device="cuda:0")
        mpc.reset()
        x0 = torch.zeros((B, 4), device=sim.device)
        contact = torch.ones(B, dtype=torch.uint8, device=sim.device)
    rng = np.random.default_rng(seed)  # the same clean actions whatever is poisoned
    prng = np.random.default_rng(seed + 1)
    finite = True
    terminated = 0
    for _ in range(steps):
        act = clean_action(kind, model, B, rng)
        if len(poisoned_envs):
            hit = poisoned_envs[prng.random(len(poisoned_envs)) < 0.3]
            flat = act.reshape(B, -1)
            flat[hit, prng.integers(0, flat.shape[1], len(hit))] = prng.choice(POISON, len(hit))
        a = torch.from_numpy(act).to(sim.device)
        if kind == "pendulum":
            obs, _, term, _ = sim.step_pendulum(a)
        elif kind == "gyropod":
            obs, _, term, _ = sim.step_gyropod(a)
        elif kind == "servos":
            obs, _, term, _ = sim.step_servos(a)
        else:
            obs, _, term, _ = sim.step_base_velocity_mpc(mpc, a, x0, contact)
            finite = finite and bool(torch.isfinite(mpc.commanded_velocity).all()) and bool(torch.isfinite(x0).all())
        finite = finite and bool(torch.isfinite(obs).all())
        terminated += int(term[torch.from_numpy(poisoned_envs).to(sim.device)].sum()) if len(poisoned_envs) else 0
    state = sim.state_numpy().copy()
    extra = None if mpc is None else (mpc.workspace.cpu().numpy().copy(), mpc.commanded_velocity.cpu().numpy().copy())
    counts = sim.guard_counts()
    sim.close()
    return finite, state, terminated, counts, extra


@pytest.mark.parametrize("lanes", [8, 2, 1])
@pytest.mark.parametrize("kind", ["pendulum", "gyropod", "servos", "base_velocity"])
def test_poisoned_actions_never_leave_a_step_and_touch_no_other_env(kind, lanes, monkeypatch):
    B, steps = 1024, 2000
    poisoned = np.sort(np.random.default_rng(0).choice(B, B // 100, replace=False))
    finite_p, state_p, _, counts_p, extra_p = run(kind, lanes, poisoned, steps, monkeypatch)
    finite_c, state_c, _, counts_c, extra_c = run(kind, lanes, np.zeros(0, dtype=np.int64), steps, monkeypatch)
    assert finite_c and counts_c == {"commands_replaced": 0, "states_replaced": 0}
    assert finite_p and np.isfinite(state_p).all(), "a non-finite value left a step"
    others = np.setdiff1d(np.arange(B), poisoned)
    assert np.array_equal(state_p[:, others], state_c[:, others]), "a poisoned env changed another env"
    if extra_p is not None:
        assert np.isfinite(extra_p[0]).all() and np.array_equal(extra_p[0][:, others], extra_c[0][:, others])
    # NaN words are replaced and counted (about 0.3 x 10 envs x 2000 steps x 1/5 of the poison values, where the word is one the guard owns)
    assert counts_p["commands_replaced"] > 100, counts_p
    assert counts_p["states_replaced"] == 0, counts_p  # sanitised commands keep every state finite


@pytest.mark.parametrize("lanes", [8, 2, 1])
def test_poisoned_servo_actions_against_the_oracle(lanes, monkeypatch):
    """The replaced words are the checker's: same poisoned [B, 6, 6] actions on
    both sides for 40 steps, states compared like any closed loop, counters equal."""
    from oracle import oracle as O

    B, steps = 512, 40
    sim = make_sim(B, lanes, monkeypatch)
    ref = O.Oracle(sim.model, config(B))
    ref.reset()
    oracle_guard_counts(reset=True)
    rng = np.random.default_rng(2)
    for _ in range(steps):
        act = clean_action("servos", sim.model, B, rng)
        hit = rng.random(B) < 0.05
        flat = act.reshape(B, -1)
        flat[hit, rng.integers(0, 36, int(hit.sum()))] = rng.choice(POISON, int(hit.sum()))
        obs, _, term, _ = sim.step_servos(torch.from_numpy(act).to(sim.device))
        obs_r, _, term_r, _ = ref.step_servos(act.astype(np.float64))
        assert np.array_equal(term.cpu().numpy(), term_r)
        assert torch.isfinite(obs).all() and np.isfinite(obs_r).all()
    counts = sim.guard_counts()
    assert (counts["commands_replaced"], counts["states_replaced"]) == oracle_guard_counts() and counts["commands_replaced"] > 50
    err = np.abs(sim.state_numpy().astype(np.float64) - ref.state)
    assert err[abi.S_POS:abi.S_POS + 7].max() <= 1e-4 and err[abi.S_Q:abi.S_Q + 2].max() <= 1e-3, err.max(axis=1)
    sim.close()


@pytest.mark.parametrize("lanes", [8, 2, 1])
@pytest.mark.parametrize("kind", ["pendulum", "servos"])
@pytest.mark.parametrize("what", ["state", "force", "inertia"])
def test_an_env_that_is_not_finite_terminates_and_is_reinitialised(kind, what, lanes, monkeypatch):
    from oracle import oracle as O

    B = 256
    sim = make_sim(B, lanes, monkeypatch)
    cfg, model = sim.config, sim.model
    bad = np.array([3, 64, 200])
    force = None
    if what == "force":
        force = torch.zeros((3, B), device=sim.device)
        force[1, torch.from_numpy(bad)] = float("nan")
        sim.set_external_force(force)
    elif what == "inertia":
        sim.randomize_inertias(0.1)
        records = sim.body_inertials.clone()
        records[4, torch.from_numpy(bad)] = float("inf")
        sim.set_body_inertials(records)
    else:
        sim.state[abi.S_ANGVEL + 2, torch.from_numpy(bad)] = float("nan")
    episodes = sim.state[abi.S_EPISODE].cpu().numpy().copy()
    rng = np.random.default_rng(0)

    def step():
        a = torch.from_numpy(clean_action(kind, model, B, rng)).to(sim.device)
        return sim.step_pendulum(a) if kind == "pendulum" else sim.step_servos(a)

    obs, _, term, _ = step()
    expect = np.isin(np.arange(B), bad).astype(np.uint8)
    assert torch.isfinite(obs).all() and torch.isfinite(sim.state).all()
    assert np.array_equal(term.cpu().numpy(), expect)
    s = sim.state_numpy()
    assert np.allclose(s[abi.S_POS:abi.S_POS + 3, bad], np.array(cfg.init_pos[:])[:, None]) and (s[abi.S_QD:abi.S_QD + 6, bad] == 0).all()
    assert (s[abi.S_DONE, bad] == 1).all() and s[abi.S_DONE].sum() == len(bad)
    assert sim.guard_counts() == {"commands_replaced": 0, "states_replaced": len(bad)}
    if what != "state":  # the cause is still there: the envs are re-initialised and caught again, the others never notice
        sim.set_external_force(None) if what == "force" else sim.set_body_inertials(None)
    obs, _, term, _ = step()  # NEXT_STEP autoreset
    assert not term.any() and torch.isfinite(obs).all()
    s = sim.state_numpy()
    assert (s[abi.S_EPISODE, bad] == episodes[bad] + 1).all() and s[abi.S_DONE].sum() == 0
    sim.close()


def test_same_step_autoreset_completes_a_guarded_env_inside_the_call(monkeypatch):
    """SAME_STEP (the IN_PLACE eight-lane kernels): the guarded env's last
    observation is the initial state's, the call returns it re-initialised."""
    B = 128
    sim = make_sim(B, 8, monkeypatch, autoreset=abi.AUTORESET_DISABLED)
    final = torch.zeros((B, 4), device=sim.device)
    sim.set_final_observation(final)
    sim.state[abi.S_LINVEL, 7] = float("inf")
    episodes = sim.state[abi.S_EPISODE].clone()
    obs, _, term, _ = sim.step_pendulum(torch.zeros(B, device=sim.device))
    assert term[7] == 1 and term.sum() == 1 and torch.isfinite(obs).all() and torch.isfinite(final).all()
    assert sim.state[abi.S_EPISODE, 7] == episodes[7] + 1 and sim.state[abi.S_DONE].sum() == 0
    assert abs(float(final[7, 0])) < 1e-6  # pitch of the initial state; the restarted env's is a draw
    sim.close()
