"""CPU restatement of generalized advantage estimation (TEST INFRASTRUCTURE
ONLY). The reference repository contains no learner; the recurrence is the
published one (Schulman et al. 2016, "High-dimensional continuous control
using generalized advantage estimation", eqs. 11-16), written as a plain
double-precision loop."""

import numpy as np


def gae(rewards, values, episode_starts, last_values, last_dones, gamma, gae_lambda):
    rewards = np.asarray(rewards, dtype=np.float64)
    values = np.asarray(values, dtype=np.float64)
    starts = np.asarray(episode_starts, dtype=np.float64)
    T, N = rewards.shape
    advantages = np.zeros((T, N))
    last = np.zeros(N)
    for t in reversed(range(T)):
        if t == T - 1:
            non_terminal = 1.0 - np.asarray(last_dones, dtype=np.float64)
            next_values = np.asarray(last_values, dtype=np.float64)
        else:
            non_terminal = 1.0 - starts[t + 1]
            next_values = values[t + 1]
        delta = rewards[t] + gamma * next_values * non_terminal - values[t]
        last = delta + gamma * gae_lambda * non_terminal * last
        advantages[t] = last
    return advantages, advantages + values
