// Test-only: runs the product's device functions on the HOST so that a single
// physics substep can be compared with the oracle without a GPU.
// (step_kernels.hpp + host_setup.hpp, not upkie_hip.hip: no launch, so no kernel is instantiated and this builds in seconds)
#include <string>

#include "../upkie_amd/csrc/host_setup.hpp"

extern "C" int harness_substep(const UpkieModel* model, float* st, const float* tau, float h, const float* records,
                               const float* ext_forces, const UpkieExternalForces* ext_slots) {
  DevModel M;
  std::string why;
  if (!convert_model(model, &M, &why)) return -1;
  Phys s;
  s.pos = v3(st[UPKIE_S_POS], st[UPKIE_S_POS + 1], st[UPKIE_S_POS + 2]);
  s.qw = st[UPKIE_S_QUAT]; s.qx = st[UPKIE_S_QUAT + 1]; s.qy = st[UPKIE_S_QUAT + 2]; s.qz = st[UPKIE_S_QUAT + 3];
  s.linvel = v3(st[UPKIE_S_LINVEL], st[UPKIE_S_LINVEL + 1], st[UPKIE_S_LINVEL + 2]);
  s.angvel = v3(st[UPKIE_S_ANGVEL], st[UPKIE_S_ANGVEL + 1], st[UPKIE_S_ANGVEL + 2]);
  for (int j = 0; j < 6; ++j) { s.q[j] = st[UPKIE_S_Q + j]; s.qd[j] = st[UPKIE_S_QD + j]; }
  float t[6];
  for (int j = 0; j < 6; ++j) t[j] = tau[j];
  ExtSlots x{};
  if (ext_forces && ext_slots) {
    x.count = ext_slots->count;
    for (int i = 0; i < x.count; ++i) {
      x.body[i] = ext_slots->body[i];
      x.local[i] = ext_slots->local[i];
      for (int k = 0; k < 3; ++k) x.point[i][k] = (float)ext_slots->point[i][k];
    }
  }
  const ExtForces ext{ext_forces && ext_slots ? ext_forces : nullptr, 1, &x};  // [count][3], one env
  DevLimits Lm;
  model_limits(M, &Lm);
  BodyInertials bi;
  if (records) load_body_inertials(records, 1, bi);  // [70], one env
  bool c = physics_substep(M, Lm, s, t, h, records ? &bi : nullptr, ext);
  st[UPKIE_S_POS] = s.pos.x; st[UPKIE_S_POS + 1] = s.pos.y; st[UPKIE_S_POS + 2] = s.pos.z;
  st[UPKIE_S_QUAT] = s.qw; st[UPKIE_S_QUAT + 1] = s.qx; st[UPKIE_S_QUAT + 2] = s.qy; st[UPKIE_S_QUAT + 3] = s.qz;
  st[UPKIE_S_LINVEL] = s.linvel.x; st[UPKIE_S_LINVEL + 1] = s.linvel.y; st[UPKIE_S_LINVEL + 2] = s.linvel.z;
  st[UPKIE_S_ANGVEL] = s.angvel.x; st[UPKIE_S_ANGVEL + 1] = s.angvel.y; st[UPKIE_S_ANGVEL + 2] = s.angvel.z;
  for (int j = 0; j < 6; ++j) { st[UPKIE_S_Q + j] = s.q[j]; st[UPKIE_S_QD + j] = s.qd[j]; }
  return c ? 1 : 0;
}

// One substep under the Bullet-like contact specification (bullet_like.hpp) on the host: manifold [64] in / out.
extern "C" int harness_substep_bullet_like(const UpkieModel* model, float* st, const float* tau, float h, float* manifold) {
  DevModel M;
  std::string why;
  if (!convert_model(model, &M, &why)) return -1;
  Phys s;
  s.pos = v3(st[UPKIE_S_POS], st[UPKIE_S_POS + 1], st[UPKIE_S_POS + 2]);
  s.qw = st[UPKIE_S_QUAT]; s.qx = st[UPKIE_S_QUAT + 1]; s.qy = st[UPKIE_S_QUAT + 2]; s.qz = st[UPKIE_S_QUAT + 3];
  s.linvel = v3(st[UPKIE_S_LINVEL], st[UPKIE_S_LINVEL + 1], st[UPKIE_S_LINVEL + 2]);
  s.angvel = v3(st[UPKIE_S_ANGVEL], st[UPKIE_S_ANGVEL + 1], st[UPKIE_S_ANGVEL + 2]);
  for (int j = 0; j < 6; ++j) { s.q[j] = st[UPKIE_S_Q + j]; s.qd[j] = st[UPKIE_S_QD + j]; }
  float t[6];
  for (int j = 0; j < 6; ++j) t[j] = tau[j];
  ExtSlots x{};
  const ExtForces ext{nullptr, 1, &x};
  DevLimits Lm;
  model_limits(M, &Lm);
  float mf[BL_MANIFOLD_WORDS];
  for (int w = 0; w < BL_MANIFOLD_WORDS; ++w) mf[w] = manifold[w];
  const bool c = physics_substep<false, true>(M, Lm, s, t, h, nullptr, ext, nullptr, &mf);
  for (int w = 0; w < BL_MANIFOLD_WORDS; ++w) manifold[w] = mf[w];
  st[UPKIE_S_POS] = s.pos.x; st[UPKIE_S_POS + 1] = s.pos.y; st[UPKIE_S_POS + 2] = s.pos.z;
  st[UPKIE_S_QUAT] = s.qw; st[UPKIE_S_QUAT + 1] = s.qx; st[UPKIE_S_QUAT + 2] = s.qy; st[UPKIE_S_QUAT + 3] = s.qz;
  st[UPKIE_S_LINVEL] = s.linvel.x; st[UPKIE_S_LINVEL + 1] = s.linvel.y; st[UPKIE_S_LINVEL + 2] = s.linvel.z;
  st[UPKIE_S_ANGVEL] = s.angvel.x; st[UPKIE_S_ANGVEL + 1] = s.angvel.y; st[UPKIE_S_ANGVEL + 2] = s.angvel.z;
  for (int j = 0; j < 6; ++j) { st[UPKIE_S_Q + j] = s.q[j]; st[UPKIE_S_QD + j] = s.qd[j]; }
  return c ? 1 : 0;
}

// contact_pgs6() on the host (fp32, the arithmetic of the kernels): A [21] packed lower, rhs [6], lam [6] in / out
extern "C" int harness_contact_pgs6(const UpkieModel* model, const float* A, const float* rhs, float* lam, int both_tires) {
  DevModel M;
  std::string why;
  if (!convert_model(model, &M, &why)) return -1;
  float a[21], r[6], l[6];
  for (int k = 0; k < 21; ++k) a[k] = A[k];
  for (int k = 0; k < 6; ++k) { r[k] = rhs[k]; l[k] = lam[k]; }
  (void)both_tires;  // (the sweeps are one loop for every system: contact_pgs6)
  const int sweeps = contact_pgs6(M, a, r, l);
  for (int k = 0; k < 6; ++k) lam[k] = l[k];
  return sweeps;
}

// contact_solve6() on the host: the same system through the active-set solve first (returns -1 / -2: the attempt that
// was accepted), then the sweeps (returns their count)
extern "C" int harness_contact_solve6(const UpkieModel* model, const float* A, const float* rhs, float* lam) {
  DevModel M;
  std::string why;
  if (!convert_model(model, &M, &why)) return -100;
  float a[21], r[6], l[6];
  for (int k = 0; k < 21; ++k) a[k] = A[k];
  for (int k = 0; k < 6; ++k) { r[k] = rhs[k]; l[k] = lam[k]; }
  const int code = contact_solve6(M, a, r, l);
  for (int k = 0; k < 6; ++k) lam[k] = l[k];
  return code;
}

// fuse_links() on the host: link factors [UPKIE_MAX_LINKS] -> records [70]
extern "C" int harness_fuse_links(const UpkieModel* model, const float* factors, float* records) {
  DevLinks L;
  std::string why;
  if (!convert_links(model, &L, &why)) return -1;
  float f[UPKIE_MAX_LINKS];
  for (int l = 0; l < UPKIE_MAX_LINKS; ++l) f[l] = factors[l];
  fuse_links(L, f, records, 1);
  return L.count;
}

// ---------------------------------------------------------------------------
// The eight-lanes-per-env substep (octet.hpp) on the host: the eight lanes of ONE
// env run as eight threads in lockstep; every lane exchange goes through a
// shared slot array between two barriers (oct_host_get).
#include <atomic>
#include <thread>

#if !defined(__HIP_DEVICE_COMPILE__)

namespace {
struct SpinBarrier {
  std::atomic<int> count{0};
  std::atomic<int> generation{0};
  int parties = 8;
  void wait() {
    const int gen = generation.load(std::memory_order_acquire);
    if (count.fetch_add(1, std::memory_order_acq_rel) + 1 == parties) {
      count.store(0, std::memory_order_relaxed);
      generation.fetch_add(1, std::memory_order_release);
    } else {
      while (generation.load(std::memory_order_acquire) == gen) std::this_thread::yield();
    }
  }
};
void spin_barrier_wait(void* b) { static_cast<SpinBarrier*>(b)->wait(); }
}  // namespace

// st: one env's state words (in / out), tau: six commanded torques, records: [70] or null,
// trunk_wrench: [6] (base frame, about the base origin) or null. Runs `substeps` substeps;
// status[i] receives OCT_CONTACT / OCT_NO_CONTACT of substep i.
static int run_octet(const UpkieModel* model, float* st, const float* tau, float h, const float* records, const float* trunk_wrench, int substeps,
                     int* status, int limits_in_registers, float* bullet_applied /* [2]: Bullet-like contacts on these applied impulses, or null */,
                     int* sweeps_out = nullptr /* [substeps]: sweeps the contact solve of each substep ran, or null */) {
  DevModel M;
  std::string why;
  if (!convert_model(model, &M, &why)) return -1;
  DevLimits Lm;
  model_limits(M, &Lm);
  DevConfig C;
  std::memset(&C, 0, sizeof(C));
  C.h = h;  // load_oct_lane derives the launch's contact constants from it
  SpinBarrier barrier;
  float slots[8];
  OctPhys result[8];
  int lane_status[8][64];
  int lane_sweeps[64];
  std::thread threads[8];
  if (substeps > 64) substeps = 64;
  for (int t = 0; t < 8; ++t) {
    threads[t] = std::thread([&, t]() {
      OctHostLane me{t, slots, spin_barrier_wait, &barrier};
      g_oct_lane = &me;
      const int l = t & 3, leg = t >> 2;
      const OctLane L = load_oct_lane(M, Lm, C, l, leg, records, 1);
      const int joint = 3 * leg + (l > 0 ? l - 1 : 0);
      OctPhys s;
      s.pos = v3(st[UPKIE_S_POS], st[UPKIE_S_POS + 1], st[UPKIE_S_POS + 2]);
      s.qw = st[UPKIE_S_QUAT]; s.qx = st[UPKIE_S_QUAT + 1]; s.qy = st[UPKIE_S_QUAT + 2]; s.qz = st[UPKIE_S_QUAT + 3];
      s.linvel = v3(st[UPKIE_S_LINVEL], st[UPKIE_S_LINVEL + 1], st[UPKIE_S_LINVEL + 2]);
      s.angvel = v3(st[UPKIE_S_ANGVEL], st[UPKIE_S_ANGVEL + 1], st[UPKIE_S_ANGVEL + 2]);
      s.q = l > 0 ? st[UPKIE_S_Q + joint] : 0.f;
      s.qd = l > 0 ? st[UPKIE_S_QD + joint] : 0.f;
      const float own_tau = l > 0 ? tau[joint] : 0.f;
      if (bullet_applied) s.bl_applied = bullet_applied[leg];
      LimitWorkspace workspace;  // (the kernel keeps one per env in LDS; here every lane's thread has its own)
      for (int i = 0; i < substeps; ++i) {
        OctRare rare{0, 0};
        const int r = bullet_applied        ? physics_substep_octet<false, false, true>(M, Lm, L, s, own_tau, h, trunk_wrench, &workspace, &rare)
                      : limits_in_registers ? physics_substep_octet<true>(M, Lm, L, s, own_tau, h, trunk_wrench, &workspace, &rare)
                                            : physics_substep_octet<false>(M, Lm, L, s, own_tau, h, trunk_wrench, &workspace, &rare);
        lane_status[t][i] = r;
        if (t == 1) lane_sweeps[i] = rare.sweeps;
      }
      result[t] = s;
      g_oct_lane = nullptr;
    });
  }
  for (int t = 0; t < 8; ++t) threads[t].join();
  // every lane holds the same base; each joint lane its own joint
  int consistent = 1;
  for (int t = 1; t < 8; ++t) {
    const OctPhys &a = result[0], &b = result[t];
    if (a.pos.x != b.pos.x || a.pos.y != b.pos.y || a.pos.z != b.pos.z || a.qw != b.qw || a.qx != b.qx || a.qy != b.qy || a.qz != b.qz ||
        a.linvel.x != b.linvel.x || a.linvel.y != b.linvel.y || a.linvel.z != b.linvel.z || a.angvel.x != b.angvel.x ||
        a.angvel.y != b.angvel.y || a.angvel.z != b.angvel.z)
      consistent = 0;
    for (int i = 0; i < substeps; ++i)
      if (lane_status[t][i] != lane_status[0][i]) consistent = 0;
  }
  const OctPhys& s = result[0];
  st[UPKIE_S_POS] = s.pos.x; st[UPKIE_S_POS + 1] = s.pos.y; st[UPKIE_S_POS + 2] = s.pos.z;
  st[UPKIE_S_QUAT] = s.qw; st[UPKIE_S_QUAT + 1] = s.qx; st[UPKIE_S_QUAT + 2] = s.qy; st[UPKIE_S_QUAT + 3] = s.qz;
  st[UPKIE_S_LINVEL] = s.linvel.x; st[UPKIE_S_LINVEL + 1] = s.linvel.y; st[UPKIE_S_LINVEL + 2] = s.linvel.z;
  st[UPKIE_S_ANGVEL] = s.angvel.x; st[UPKIE_S_ANGVEL + 1] = s.angvel.y; st[UPKIE_S_ANGVEL + 2] = s.angvel.z;
  for (int t = 0; t < 8; ++t) {
    const int l = t & 3, leg = t >> 2;
    if (l == 0) continue;
    st[UPKIE_S_Q + 3 * leg + l - 1] = result[t].q;
    st[UPKIE_S_QD + 3 * leg + l - 1] = result[t].qd;
  }
  for (int i = 0; i < substeps; ++i) status[i] = lane_status[0][i];
  if (sweeps_out)
    for (int i = 0; i < substeps; ++i) sweeps_out[i] = lane_sweeps[i];
  if (bullet_applied) {
    bullet_applied[0] = result[1].bl_applied;  // (a lane of the left quad, of the right quad)
    bullet_applied[1] = result[5].bl_applied;
  }
  return consistent;
}

extern "C" int harness_substep_octet(const UpkieModel* model, float* st, const float* tau, float h, const float* records,
                                     const float* trunk_wrench, int substeps, int* status, int limits_in_registers) {
  return run_octet(model, st, tau, h, records, trunk_wrench, substeps, status, limits_in_registers, nullptr);
}

// The same, reporting what the contact solve of each substep did: codes[substeps] = sweeps run (> 0), the active set that was
// accepted (-1 / -2, oct_active_set), 0 (admissible direct solution, both tires unloading, or no contact)
extern "C" int harness_substep_octet_codes(const UpkieModel* model, float* st, const float* tau, float h, int substeps, int* status, int* codes) {
  return run_octet(model, st, tau, h, nullptr, nullptr, substeps, status, 0, nullptr, codes);
}

// The eight-lane substep under the Bullet-like contact specification: applied[2] = the tires' applied normal impulses (in / out).
extern "C" int harness_substep_octet_bullet_like(const UpkieModel* model, float* st, const float* tau, float h, int substeps, int* status, float* applied) {
  return run_octet(model, st, tau, h, nullptr, nullptr, substeps, status, 0, applied);
}
// Test hook: the eight-lane Bullet-like solves that follow record the system they sweep and every sweep's impulses into `probe`
// (BulletLikeProbe: system [50], change [64], lam [64][6], sweeps); null: off.
extern "C" void harness_bullet_like_probe(void* probe) { g_bullet_like_probe = static_cast<BulletLikeProbe*>(probe); }
extern "C" int harness_bullet_like_probe_bytes(void) { return (int)sizeof(BulletLikeProbe); }
#endif  // host pass only

// The balancer's host setup (csrc/mpc.hpp: condensing, Minv, its fp32 layout, the fp16 two-term layout of round 6, Minv Kx and
// Minv kv), for tests/test_mpc_host_setup.py. `np` = 16 x tiles; buffers sized by the caller: minv_perm [np * np], kx [np * 4],
// kv [np], minv_h [64 * tiles * ceil(tiles / 2) * 16] (uint16), gx [np * 4], gv [np].
extern "C" int harness_mpc_host_setup(const UpkieMpcConfig* config, int np, float* minv_perm, float* kx, float* kv, uint16_t* minv_h,
                                      float* gx, float* gv, float* scale) {
  std::vector<float> m, x, v, g4, g1;
  std::vector<uint16_t> h;
  std::string why;
  if (!upkie::mpc_host_setup(*config, np, &m, &x, &v, &why, &h, &g4, &g1, scale)) return -1;
  std::copy(m.begin(), m.end(), minv_perm);
  std::copy(x.begin(), x.end(), kx);
  std::copy(v.begin(), v.end(), kv);
  std::copy(h.begin(), h.end(), minv_h);
  std::copy(g4.begin(), g4.end(), gx);
  std::copy(g1.begin(), g1.end(), gv);
  return (int)h.size();
}
