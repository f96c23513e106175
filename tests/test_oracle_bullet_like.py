"""The product's contact specification against the Bullet-like one of
oracle/upkie_oracle.c (what Bullet 3.25's btMultiBodyConstraintSolver is
published to do inside pybullet.stepSimulation(), pybullet_backend.py:306;
SURVEY.md Appendix B.1 / B.2: persistent manifolds of up to four points per
tire, 50 fixed warm-started sweeps, cone friction along the sliding direction,
no friction CFM). Bullet itself is not available here: this bounds how much
the knowingly different contact spec can matter on the headline workload
(profiles/r03_bullet_like_deviation.txt has the whole table, from
tools/bullet_like_deviation.py). Test infrastructure only."""

import numpy as np

from oracle import oracle as O
from tests.helpers import randomized_config
from upkie_amd import abi
from upkie_amd.model.model import Model


def test_c2_trajectories_agree_under_both_contact_specifications():
    model = Model().struct
    cfg = randomized_config(64, seed=0)
    ours, bullet = O.Oracle(model, cfg), O.Oracle(model, cfg)
    bullet.use_bullet_like_contacts()
    oa = ours.reset()[:, [1, 0, 4, 3]]
    ob = bullet.reset()[:, [1, 0, 4, 3]]
    assert np.abs(oa - ob)[:, :2].max() < 1e-6 and np.abs(oa - ob).max() < 1e-4  # (the one torque-free substep of the reset)
    worst = np.zeros(4)
    for _ in range(100):
        oa, _, ta, _ = ours.step_pendulum_agent(oa)
        ob, _, tb, _ = bullet.step_pendulum_agent(ob)
        assert not ta.any() and not tb.any()
        worst = np.maximum(worst, np.abs(oa - ob).max(axis=0))
    # pitch [rad], ground position [m], pitch rate [rad/s], ground velocity [m/s]: the first steps differ most (the
    # Bullet-like normal impulses are warm-started and need a few substeps to settle after the landing)
    assert worst[0] < 1e-5 and worst[1] < 2e-5 and worst[2] < 1e-3 and worst[3] < 1e-3, worst
    # and at the end of the run the two are indistinguishable at the tolerance the kernels are held to
    assert np.abs(oa - ob)[:, 0].max() < 2e-6 and np.abs(oa - ob)[:, 1].max() < 1e-5


def test_bullet_like_manifold_tracks_the_deepest_point_and_carries_the_weight():
    model = Model().struct
    cfg = randomized_config(16, seed=1)
    bullet = O.Oracle(model, cfg)
    bullet.use_bullet_like_contacts()
    obs = bullet.reset()[:, [1, 0, 4, 3]]
    for _ in range(300):  # the README agent: the robots roll back and forth, slowly, while they balance
        obs, *_ = bullet.step_pendulum_agent(obs)
    m = bullet.bullet_manifold.reshape(2, 4, 8, 16)
    live = m[:, :, 7, :]
    count = live.sum(axis=1)
    # a wheel that rolls keeps ONE point: the new deepest point lies within the breaking threshold (2 cm) of the cached
    # one in the wheel's frame (0.8 mm of arc per 1 ms substep at this speed) and replaces it, as in getCacheEntry
    assert count.max() <= 4 and count.min() >= 1
    applied = m[:, :, 6, :] * live
    weight = float(sum(model.mass)) * 9.81 * 1e-3  # impulse per 1 ms substep
    total = applied.sum(axis=(0, 1))
    assert np.all(np.abs(total - weight) < 0.05 * weight), (total, weight)  # the floor carries the robot
