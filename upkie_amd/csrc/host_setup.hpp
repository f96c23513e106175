// host_setup.hpp -- what `upkie_sim_create` / `upkie_sim_set_config` turn the C-ABI's model and configuration
// structures (include/upkie_hip.h, doubles) into: the fp32 blocks the kernels read (DevModel, DevLimits, DevConfig,
// DevLinks). Host code only, no kernel launch: tests/host_harness.hip includes it with step_kernels.hpp to run the
// product's device arithmetic on the CPU without instantiating a single kernel.
#pragma once
#include <cmath>
#include <cstring>
#include <string>

#include "step_kernels.hpp"

namespace upkie {

// The URDF links behind the composite bodies, as randomize_inertias sees them.
struct DevLinks {
  int count;
  int body[UPKIE_MAX_LINKS];
  int randomized[UPKIE_MAX_LINKS];
  float mass[UPKIE_MAX_LINKS];
  float com[UPKIE_MAX_LINKS][3];
  float inertia[UPKIE_MAX_LINKS][6];
};

// Fuse the (scaled) links of every composite body: mass, centre of mass and
// inertia about it, written to records[(10 * body + word) * stride].
UPKIE_HD void fuse_links(const DevLinks& L, const float (&f)[UPKIE_MAX_LINKS], float* records, size_t stride) {
  for (int b = 0; b < UPKIE_NB; ++b) {
    float m = 0.f, mx = 0.f, my = 0.f, mz = 0.f;
    for (int l = 0; l < L.count; ++l) {
      if (L.body[l] != b) continue;
      const float ml = f[l] * L.mass[l];
      m += ml;
      mx = fmaf(ml, L.com[l][0], mx); my = fmaf(ml, L.com[l][1], my); mz = fmaf(ml, L.com[l][2], mz);
    }
    const float cx = mx / m, cy = my / m, cz = mz / m;
    float I[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int l = 0; l < L.count; ++l) {
      if (L.body[l] != b) continue;
      const float ml = f[l] * L.mass[l];
      const float dx = L.com[l][0] - cx, dy = L.com[l][1] - cy, dz = L.com[l][2] - cz;
      I[0] += f[l] * L.inertia[l][0] + ml * (dy * dy + dz * dz);
      I[1] += f[l] * L.inertia[l][1] + ml * (dx * dx + dz * dz);
      I[2] += f[l] * L.inertia[l][2] + ml * (dx * dx + dy * dy);
      I[3] += f[l] * L.inertia[l][3] - ml * dx * dy;
      I[4] += f[l] * L.inertia[l][4] - ml * dx * dz;
      I[5] += f[l] * L.inertia[l][5] - ml * dy * dz;
    }
    float* r = records + (size_t)(UPKIE_INERTIAL_WORDS * b) * stride;
    r[0] = m;
    r[(size_t)1 * stride] = cx; r[(size_t)2 * stride] = cy; r[(size_t)3 * stride] = cz;
#pragma unroll
    for (int d = 0; d < 6; ++d) r[(size_t)(4 + d) * stride] = I[d];
  }
}

}  // namespace upkie

using namespace upkie;

// smallest relative change of an impulse the fp32 Gauss-Seidel sweeps can resolve (see convert_model)
constexpr float kSweepToleranceFloor = 1e-5f;
static bool convert_model(const UpkieModel* m, DevModel* d, std::string* why) {
  std::memset(d, 0, sizeof(*d));
  for (int i = 0; i < UPKIE_NB; ++i) {
    d->mass[i] = (float)m->mass[i];
    if (!(m->mass[i] > 0.0)) {
      *why = "body masses must be positive";
      return false;
    }
    for (int k = 0; k < 3; ++k) d->com[i][k] = (float)m->com[i][k];
    for (int k = 0; k < 6; ++k) d->inertia[i][k] = (float)m->inertia[i][k];
  }
  for (int j = 0; j < UPKIE_NJ; ++j) {
    const double* a = m->joint_axis[j];
    if (std::fabs(a[0]) > 1e-6 || std::fabs(a[2]) > 1e-6 || std::fabs(std::fabs(a[1]) - 1.0) > 1e-6) {
      *why = "joint axes must be lateral (+-y of the base frame at zero configuration)";
      return false;
    }
    d->joint_sign[j] = a[1] > 0 ? 1.f : -1.f;
    for (int k = 0; k < 3; ++k) d->joint_pos[j][k] = (float)m->joint_pos[j][k];
    d->joint_lower[j] = (float)m->joint_lower[j];
    d->joint_upper[j] = (float)m->joint_upper[j];
    d->joint_effort[j] = (float)m->joint_effort[j];
    d->joint_velocity[j] = (float)m->joint_velocity[j];
    d->joint_damping[j] = (float)m->joint_damping[j];
  }
  d->wheel_radius = (float)m->wheel_radius;
  bool axisym = true;
  for (int w = 0; w < 2; ++w) {
    if (std::fabs(m->wheel_center[w][0]) > 1e-9 || std::fabs(m->wheel_center[w][2]) > 1e-9) {
      *why = "tire centres must lie on the wheel axis";
      return false;
    }
    for (int k = 0; k < 3; ++k) d->wheel_center[w][k] = (float)m->wheel_center[w][k];
    int b = 3 * w + 3;
    const double* I = m->inertia[b];
    const double* c = m->com[b];
    if (std::fabs(c[0]) > 1e-9 || std::fabs(c[2]) > 1e-9 || std::fabs(I[0] - I[2]) > 1e-12 || std::fabs(I[3]) > 1e-12 ||
        std::fabs(I[4]) > 1e-12 || std::fabs(I[5]) > 1e-12)
      axisym = false;
  }
  // ... and stays so under randomize_inertias only if every link fused into a wheel is
  for (int l = 0; l < m->num_links && l < UPKIE_MAX_LINKS; ++l) {
    if (m->link_body[l] != 3 && m->link_body[l] != 6) continue;
    const double* I = m->link_inertia[l];
    const double* c = m->link_com[l];
    if (std::fabs(c[0]) > 1e-9 || std::fabs(c[2]) > 1e-9 || std::fabs(I[0] - I[2]) > 1e-12 || std::fabs(I[3]) > 1e-12 ||
        std::fabs(I[4]) > 1e-12 || std::fabs(I[5]) > 1e-12)
      axisym = false;
  }
  d->wheel_axisymmetric = axisym ? 1 : 0;
  d->wheel_base = (float)m->wheel_base;
  d->left_sign = (float)m->left_sign;
  for (int k = 0; k < 3; ++k) d->imu_pos[k] = (float)m->imu_pos[k];
  for (int k = 0; k < 9; ++k) d->rot_base_to_imu[k] = (float)m->rot_base_to_imu[k];
  d->gravity = (float)m->gravity;
  d->contact_stiffness = (float)m->contact_stiffness;
  d->contact_damping = (float)m->contact_damping;
  d->friction_mu = (float)m->friction_mu;
  d->friction_cfm = (float)m->friction_cfm;
  d->contact_breaking_threshold = (float)m->contact_breaking_threshold;
  d->base_linear_damping = (float)m->base_linear_damping;
  d->base_angular_damping = (float)m->base_angular_damping;
  d->max_joint_velocity = (float)m->max_joint_velocity;
  d->pgs_iterations = m->pgs_iterations;
  // The sweeps run in fp32: a six-term residual cannot be told from zero below ~1e-5 of the largest impulse. With the
  // model's default 1e-6 (the fp64 oracle reaches it within 26 sweeps on the C5 workload) 2.5 % of the infeasible
  // substeps ran into the iteration cap, 50 sweeps that changed nothing, and a launch lasts as long as its slowest
  // wavefront: 4e-6 leaves 0.03 % of them at the cap, 1e-5 none (profiles/r02_sweep_tolerance.txt).
  d->pgs_tolerance = fmaxf((float)m->pgs_tolerance, kSweepToleranceFloor);
  d->enforce_joint_limits = m->enforce_joint_limits ? 1 : 0;
  for (int leg = 0; leg < 2; ++leg) {
    float* t = d->leg_table[leg];
    for (int k = 0; k < 3; ++k) {
      const int b = 1 + 3 * leg + k, j = 3 * leg + k;
      t[LT_MASS + k] = d->mass[b];
      t[LT_SIGN + k] = d->joint_sign[j];
      for (int a = 0; a < 3; ++a) {
        t[LT_COM + 3 * k + a] = d->com[b][a];
        t[LT_POS + 3 * k + a] = d->joint_pos[j][a];
      }
      for (int a = 0; a < 6; ++a) t[LT_INERTIA + 6 * k + a] = d->inertia[b][a];
      t[LT_DAMPING + k] = d->joint_damping[j];
      t[LT_EFFORT + k] = d->joint_effort[j];
      t[LT_VELOCITY + k] = d->joint_velocity[j];
      t[LT_WHEEL_CENTER + k] = d->wheel_center[leg][k];
      t[LT_LOWER + k] = d->joint_lower[j];
      t[LT_UPPER + k] = d->joint_upper[j];
      t[LT_BOUNDED + k] = (d->joint_lower[j] > -1e30f && d->joint_upper[j] < 1e30f) ? 1.f : 0.f;
    }
  }
  for (int leg = 0; leg < 2; ++leg) {
    for (int l = 0; l < 4; ++l) {
      float* t = d->oct_table[4 * leg + l];
      const bool trunk = l == 0, real = !(trunk && leg == 1);
      const int k = trunk ? 0 : l - 1, b = trunk ? 0 : 1 + 3 * leg + k, j = 3 * leg + k;
      t[OT_MASS] = real ? d->mass[b] : 0.f;
      for (int a = 0; a < 3; ++a) {
        t[OT_COM + a] = d->com[b][a];
        t[OT_POS + a] = trunk ? 0.f : d->joint_pos[j][a];
        t[OT_WHEEL_CENTER + a] = d->wheel_center[leg][a];
        t[OT_E + a] = (!trunk && k == a) ? 1.f : 0.f;
      }
      for (int a = 0; a < 6; ++a) t[OT_INERTIA + a] = real ? d->inertia[b][a] : 0.f;
      t[OT_SIGN] = trunk ? 0.f : d->joint_sign[j];
      t[OT_DAMPING] = trunk ? 0.f : d->joint_damping[j];
      t[OT_LOWER] = d->joint_lower[j];
      t[OT_UPPER] = d->joint_upper[j];
      t[OT_BOUNDED] = (!trunk && d->joint_lower[j] > -1e30f && d->joint_upper[j] < 1e30f) ? 1.f : 0.f;
      t[OT_EFFORT] = d->joint_effort[j];
      t[OT_VELOCITY] = d->joint_velocity[j];
      t[OT_WJ] = trunk ? 0.f : 1.f;
      t[OT_W0] = trunk ? 1.f : 0.f;
      t[OT_W0_ONCE] = (trunk && leg == 0) ? 1.f : 0.f;
      t[OT_KEEP_PSI] = (l == 3 && d->wheel_axisymmetric) ? 0.f : 1.f;
      t[OT_KL] = (trunk && leg == 0) ? d->base_linear_damping : 0.f;
      t[OT_KA] = (trunk && leg == 0) ? d->base_angular_damping : 0.f;
    }
  }
  return true;
}

static void model_limits(const DevModel& m, DevLimits* l) {
  l->enforce = m.enforce_joint_limits;
  for (int j = 0; j < UPKIE_NJ; ++j) {
    l->lower[j] = m.joint_lower[j];
    l->upper[j] = m.joint_upper[j];
    l->bounded[j] = (m.joint_lower[j] > -1e30f && m.joint_upper[j] < 1e30f) ? 1 : 0;
  }
}

static bool convert_config(const UpkieSimConfig* c, DevConfig* d, std::string* why) {
  std::memset(d, 0, sizeof(*d));
  if (c->num_envs <= 0 || c->nb_substeps <= 0 || !(c->dt > 0.0)) {
    *why = "num_envs, nb_substeps and dt must be positive";
    return false;
  }
  // low_pass_filter asserts alpha < 0.5 (filters.py:78-79)
  if (c->dt / 1.0 >= 0.5) {
    *why = "dt too large for the leg low-pass filter (alpha >= 0.5)";
    return false;
  }
  for (int j = 0; j < UPKIE_NJ; ++j) {
    // noise is only applied above 1e-10, pybullet_backend.py:463,547
    d->control_noise[j] = c->torque_control_noise[j] > 1e-10 ? (float)c->torque_control_noise[j] : 0.f;
    d->measurement_noise[j] = c->torque_measurement_noise[j] > 1e-10 ? (float)c->torque_measurement_noise[j] : 0.f;
    if (d->control_noise[j] > 0.f) d->any_control_noise = 1;
    if (d->measurement_noise[j] > 0.f) d->any_measurement_noise = 1;
    d->joint_friction[j] = (float)c->joint_friction[j];
    d->init_joint[j] = (float)c->init_joint[j];
  }
  d->num_envs = c->num_envs;
  d->nb_substeps = c->nb_substeps;
  d->dt = (float)c->dt;
  d->h = (float)(c->dt / c->nb_substeps);
  d->kp = (float)c->torque_control_kp;
  d->kd = (float)c->torque_control_kd;
  d->fall_pitch = (float)c->fall_pitch;
  d->max_ground_velocity = (float)c->max_ground_velocity;
  d->max_yaw_velocity = (float)c->max_yaw_velocity;
  d->leg_gain_scale = (float)c->leg_gain_scale;
  d->max_gain_scale = (float)c->max_gain_scale;
  for (int k = 0; k < 3; ++k) {
    d->init_pos[k] = (float)c->init_pos[k];
    d->init_linvel[k] = (float)c->init_linvel[k];
    d->init_angvel[k] = (float)c->init_angvel[k];
    d->rand_linvel[k] = (float)c->rand_linvel[k];
  }
  double qn = 0;
  for (int k = 0; k < 4; ++k) qn += c->init_quat[k] * c->init_quat[k];
  if (std::fabs(qn - 1.0) > 1e-5) {  // rotations.py:50-51
    *why = "init_quat is not normalized";
    return false;
  }
  for (int k = 0; k < 4; ++k) d->init_quat[k] = (float)c->init_quat[k];
  d->rand_roll = (float)c->rand_roll;
  d->rand_pitch = (float)c->rand_pitch;
  d->rand_x = (float)c->rand_x;
  d->rand_z = (float)c->rand_z;
  d->rand_omega_x = (float)c->rand_omega_x;
  d->rand_omega_y = (float)c->rand_omega_y;
  d->seed_lo = (unsigned)(c->seed & 0xffffffffu);
  d->seed_hi = (unsigned)(c->seed >> 32);
  d->env_lo = (unsigned)((uint64_t)c->env_id_offset & 0xffffffffu);
  d->env_hi = (unsigned)((uint64_t)c->env_id_offset >> 32);
  d->autoreset_mode = c->autoreset_mode;
  d->max_episode_steps = c->max_episode_steps > 0 ? c->max_episode_steps : 0;
  for (int k = 0; k < 4; ++k) d->agent_gains[k] = (float)c->agent_gains[k];
  d->agent_clip = (float)c->agent_clip;
  return true;
}

// the links randomize_inertias scales (one per body when the model names none)
static bool convert_links(const UpkieModel* model, DevLinks* out, std::string* why) {
  DevLinks& L = *out;
  L = DevLinks{};
  const int n = model->num_links;
  if (n < 0 || n > UPKIE_MAX_LINKS) {
    *why = "num_links out of range";
    return false;
  }
  L.count = n > 0 ? n : UPKIE_NB;
  for (int l = 0; l < L.count; ++l) {
    const int b = n > 0 ? model->link_body[l] : l;
    if (b < 0 || b >= UPKIE_NB) {
      *why = "link_body out of range";
      return false;
    }
    L.body[l] = b;
    L.randomized[l] = n > 0 ? (model->link_randomized[l] != 0) : 1;
    L.mass[l] = (float)(n > 0 ? model->link_mass[l] : model->mass[b]);
    for (int k = 0; k < 3; ++k) L.com[l][k] = (float)(n > 0 ? model->link_com[l][k] : model->com[b][k]);
    for (int k = 0; k < 6; ++k) L.inertia[l][k] = (float)(n > 0 ? model->link_inertia[l][k] : model->inertia[b][k]);
  }
  return true;
}
